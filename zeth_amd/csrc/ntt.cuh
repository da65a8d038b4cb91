// Radix-2 evaluation-domain transforms over Fr for gfx950: FFT / iFFT / cosetFFT / icosetFFT with the
// semantics of libfqfft::basic_radix2_domain<Fr> (absent submodule; Zeth call sites
// libzeth/core/evaluator_from_lagrange.tcc:43, libzeth/mpc/groth16/powersoftau_utils.tcc:125-134, and inside
// libsnark's r1cs_to_qap_witness_map -- SURVEY.md 3.1 step 5a):
//     FFT:       out[k] = sum_j a[j] w^(jk),  w = root_of_unity^(2^(s - log m))
//     iFFT:      the same with w^-1, then * m^-1
//     cosetFFT:  a[j] *= g^j first;   icosetFFT: iFFT then a[j] *= g^-j      (g = Fr::multiplicative_generator)
// natural order in, natural order out, exact field arithmetic, so outputs are bit-identical to any correct FFT.
//
// Structure: a two-pass "four-step" decomposition m = N1 * N2.  Each pass keeps its sub-transforms resident in
// LDS (up to 4096 elements = 128 KB per workgroup) and runs all of their radix-2 stages there:
//   pass A  tiles of TC adjacent columns: N1-point transforms down the columns (stride-N2 loads of TC*32-byte
//           runs), then the inter-pass twiddle w^(j2 k1), written back in place
//   pass B  tiles of TR adjacent rows: N2-point transforms along contiguous rows, fused output scaling
//           (m^-1 and/or g^-k), written transposed (TR*32-byte runs) to the output buffer
// Coset pre-scaling g^j is fused into pass A's load through two small tables g^(j1 N2), g^(j2).
#pragma once
#include <hip/hip_runtime.h>

#include "field.cuh"

namespace zg {

static constexpr int NTT_LDS_ELEMS = 4096;  // 128 KB of 32-byte elements
static constexpr int NTT_THREADS = 1024;

template <class Fr>
struct NttTables {
  unsigned log_m;
  const Fr* tw;      // w^i, i < m/2 (forward) -- Montgomery form
  const Fr* tw_inv;  // w^-i
  // two-level coset tables, N1 + N2 entries each
  const Fr* g_hi;    // g^(j1 N2), j1 < N1      (input scaling of cosetFFT)
  const Fr* g_lo;    // g^(j2), j2 < N2
  const Fr* gi_lo;   // g^(-k1) * m^-1, k1 < N1  (output scaling of icosetFFT)
  const Fr* gi_hi;   // g^(-N1 k2), k2 < N2
  Fr m_inv;
  unsigned log_n1, log_n2;
};

template <class Fr>
ZG_HD Fr lds_get(const uint4* lds, uint32_t pos) {
  union {
    uint4 q[2];
    Fr f;
  } u;
  u.q[0] = lds[2 * pos];
  u.q[1] = lds[2 * pos + 1];
  return u.f;
}
template <class Fr>
ZG_HD void lds_put(uint4* lds, uint32_t pos, const Fr& f) {
  union {
    uint4 q[2];
    Fr f;
  } u;
  u.f = f;
  lds[2 * pos] = u.q[0];
  lds[2 * pos + 1] = u.q[1];
}
template <class Fr>
ZG_HD Fr glob_get(const Fr* p) {
  union {
    uint4 q[2];
    Fr f;
  } u;
  const uint4* s = reinterpret_cast<const uint4*>(p);
  u.q[0] = s[0];
  u.q[1] = s[1];
  return u.f;
}
template <class Fr>
ZG_HD void glob_put(Fr* p, const Fr& f) {
  union {
    uint4 q[2];
    Fr f;
  } u;
  u.f = f;
  uint4* d = reinterpret_cast<uint4*>(p);
  d[0] = u.q[0];
  d[1] = u.q[1];
}

// w^e for e < m from the half table: w^(e) = -w^(e - m/2) for e >= m/2
template <class Fr>
ZG_HD Fr tw_lookup(const Fr* tw, uint32_t e, uint32_t half_m) {
  Fr w = glob_get(tw + (e & (half_m - 1)));
  return (e & half_m) ? Fr::neg(w) : w;
}

// `cols` interleaved transforms of size 2^logS held in LDS as lds[pos * cols + c], input in bit-reversed
// position order, output in natural order.  tw_step = (m/2) / (S/2): stride of this size's twiddles in the table.
template <class Fr>
__device__ __forceinline__ void lds_fft(uint4* lds, unsigned logS, uint32_t cols, const Fr* __restrict__ tw,
                                        uint32_t tw_step) {
  const uint32_t S = 1u << logS;
  const uint32_t work = (S >> 1) * cols;
  for (unsigned s = 0; s < logS; s++) {
    const uint32_t h = 1u << s;
    for (uint32_t idx = threadIdx.x; idx < work; idx += blockDim.x) {
      const uint32_t b = idx / cols, c = idx - b * cols;
      const uint32_t j = b & (h - 1);
      const uint32_t i0 = ((b >> s) << (s + 1)) + j;
      const uint32_t p0 = i0 * cols + c, p1 = (i0 + h) * cols + c;
      Fr x0 = lds_get<Fr>(lds, p0);
      Fr x1 = lds_get<Fr>(lds, p1);
      if (s != 0) {
        Fr w = glob_get(tw + (size_t)j * ((S >> (s + 1)) * tw_step));
        x1 = Fr::mul(x1, w);
      }
      lds_put<Fr>(lds, p0, Fr::add(x0, x1));
      lds_put<Fr>(lds, p1, Fr::sub(x0, x1));
    }
    __syncthreads();
  }
}

ZG_HD uint32_t bitrev32(uint32_t x, unsigned bits) { return bits ? (__builtin_bitreverse32(x) >> (32 - bits)) : 0; }

// pass A: grid = N2 / TC blocks
template <class Fr, bool COSET_IN>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass_a(Fr* __restrict__ a, NttTables<Fr> T, int inverse, uint32_t TC) {
  extern __shared__ uint4 lds[];
  const uint32_t N1 = 1u << T.log_n1, N2 = 1u << T.log_n2;
  const uint32_t half_m = (1u << T.log_m) >> 1;
  const Fr* tw = inverse ? T.tw_inv : T.tw;
  const uint32_t c0 = blockIdx.x * TC;
  const uint32_t total = N1 * TC;
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t j1 = idx / TC, c = idx - j1 * TC;
    Fr x = glob_get(a + (size_t)j1 * N2 + c0 + c);
    if (COSET_IN) x = Fr::mul(Fr::mul(x, glob_get(T.g_lo + c0 + c)), glob_get(T.g_hi + j1));
    lds_put<Fr>(lds, bitrev32(j1, T.log_n1) * TC + c, x);
  }
  __syncthreads();
  lds_fft<Fr>(lds, T.log_n1, TC, tw, N2);
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t k1 = idx / TC, c = idx - k1 * TC;
    Fr x = lds_get<Fr>(lds, idx);
    const uint32_t e = (c0 + c) * k1;  // < m
    if (e != 0 && N2 > 1) x = Fr::mul(x, tw_lookup(tw, e, half_m ? half_m : 1));
    glob_put(a + (size_t)k1 * N2 + c0 + c, x);
  }
}

// pass B: grid = N1 / TR blocks.  OUT_MODE 0: none, 1: * m^-1, 2: * g^-k m^-1 (two-level tables)
template <class Fr, int OUT_MODE>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass_b(const Fr* __restrict__ a, Fr* __restrict__ out, NttTables<Fr> T,
                                                             int inverse, uint32_t TR) {
  extern __shared__ uint4 lds[];
  const uint32_t N1 = 1u << T.log_n1, N2 = 1u << T.log_n2;
  const Fr* tw = inverse ? T.tw_inv : T.tw;
  const uint32_t r0 = blockIdx.x * TR;
  const uint32_t total = N2 * TR;
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t r = idx / N2, j2 = idx - r * N2;
    Fr x = glob_get(a + (size_t)(r0 + r) * N2 + j2);
    lds_put<Fr>(lds, bitrev32(j2, T.log_n2) * TR + r, x);
  }
  __syncthreads();
  lds_fft<Fr>(lds, T.log_n2, TR, tw, N1);
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t k2 = idx / TR, r = idx - k2 * TR;
    Fr x = lds_get<Fr>(lds, idx);
    if (OUT_MODE == 1) x = Fr::mul(x, T.m_inv);
    if (OUT_MODE == 2) x = Fr::mul(Fr::mul(x, glob_get(T.gi_lo + r0 + r)), glob_get(T.gi_hi + k2));
    glob_put(out + (size_t)k2 * N1 + r0 + r, x);
  }
}

// ---- table construction -------------------------------------------------------------------------
// out[i] = base^(i * step) * scale, i < count
template <class Fr>
__global__ void __launch_bounds__(256) k_pow_table(Fr* __restrict__ out, Fr base, uint64_t step, Fr scale, uint32_t count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint64_t e = (uint64_t)i * step;
  Fr r = scale;
  Fr b = base;
  while (e) {
    if (e & 1) r = Fr::mul(r, b);
    b = Fr::sqr(b);
    e >>= 1;
  }
  out[i] = r;
}

// elementwise helpers used by the witness map and the C ABI
template <class Fr>
__global__ void __launch_bounds__(256) k_fr_convert(Fr* __restrict__ a, size_t n, int to_mont) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x = glob_get(a + i);
  glob_put(a + i, to_mont ? Fr::to_mont(x) : Fr::from_mont(x));
}

}  // namespace zg
