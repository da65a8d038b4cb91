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
// most threads of a workgroup whose rounds take RMAX stages: a work item of 2^RMAX elements per thread, and registers
// for them (radix 8: ~150 VGPRs, so at most 512 threads)
constexpr int ntt_max_threads(int rmax) { return rmax >= 3 ? 512 : 1024; }

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

// `cols` = 2^log_cols interleaved transforms of size 2^logS held in LDS as lds[pos * cols + c], input in bit-reversed
// position order, output in natural order (decimation in time).  tw_step = (m/2) / (S/2): stride of this size's twiddles
// in the table.  The stages run R at a time in registers (R = RMAX = 3 where the transform has lanes to spare): a work
// item owns the 2^R elements base + t 2^s of one column, does the radix-2 stages s .. s+R-1 on them (twelve butterflies
// and seven twiddles for R = 3) and writes them back -- one LDS round trip and one barrier per R stages.  In the first
// round the twiddles of exponent zero are known at compile time and their products are skipped (5 products instead of
// 12).  A last round of fewer stages finishes sizes whose log is not a multiple of R.
template <class Fr, int R, bool FIRST>
__device__ __forceinline__ void lds_fft_round(uint4* lds, unsigned logS, unsigned s, unsigned log_cols, const Fr* __restrict__ tw,
                                              uint32_t tw_step) {
  constexpr uint32_t E = 1u << R;
  const uint32_t S = 1u << logS;
  const uint32_t items = (S >> R) << log_cols;
  const uint32_t cmask = (1u << log_cols) - 1u;
  for (uint32_t idx = threadIdx.x; idx < items; idx += blockDim.x) {
    const uint32_t g = idx >> log_cols, c = idx & cmask;
    const uint32_t j0 = g & ((1u << s) - 1u);
    const uint32_t base = ((g >> s) << (s + R)) + j0;
    Fr x[E];
#pragma unroll
    for (uint32_t t = 0; t < E; t++) x[t] = lds_get<Fr>(lds, ((base + (t << s)) << log_cols) + c);
#pragma unroll
    for (int q = 0; q < R; q++) {
      // stage s + q: pairs (t, t + 2^q); twiddle exponent j = j0 + (t mod 2^q) 2^s, scaled by S / 2^(s+q+1)
      const uint32_t unit = (S >> (s + q + 1)) * tw_step;
#pragma unroll
      for (uint32_t t = 0; t < E; t++) {
        if (t & (1u << q)) continue;
        const uint32_t lowt = t & ((1u << q) - 1u);
        Fr b = x[t + (1u << q)];
        if (!(FIRST && lowt == 0)) {  // FIRST: s == 0, so j0 == 0 and the exponent is lowt 2^s: zero when lowt is
          const Fr w = glob_get(tw + (size_t)(j0 + (lowt << s)) * unit);
          b = Fr::mul(b, w);
        }
        const Fr a = x[t];
        x[t] = Fr::add(a, b);
        x[t + (1u << q)] = Fr::sub(a, b);
      }
    }
#pragma unroll
    for (uint32_t t = 0; t < E; t++) lds_put<Fr>(lds, ((base + (t << s)) << log_cols) + c, x[t]);
  }
  __syncthreads();
}

template <class Fr, int R>
__device__ __forceinline__ void lds_fft_tail(uint4* lds, unsigned logS, unsigned s, unsigned log_cols, const Fr* __restrict__ tw,
                                             uint32_t tw_step) {
  // the last logS - s < R stages
  if constexpr (R >= 3) {
    if (logS - s == 2) lds_fft_round<Fr, 2, false>(lds, logS, s, log_cols, tw, tw_step);
  }
  if constexpr (R >= 2) {
    if (logS - s == 1) lds_fft_round<Fr, 1, false>(lds, logS, s, log_cols, tw, tw_step);
  }
}

template <class Fr, int RMAX>
__device__ __forceinline__ void lds_fft(uint4* lds, unsigned logS, unsigned log_cols, const Fr* __restrict__ tw, uint32_t tw_step) {
  if (logS == 0) return;
  if (logS < (unsigned)RMAX) {  // the whole transform is shorter than one round
    if (logS == 2) lds_fft_round<Fr, 2, true>(lds, logS, 0, log_cols, tw, tw_step);
    else lds_fft_round<Fr, 1, true>(lds, logS, 0, log_cols, tw, tw_step);
    return;
  }
  lds_fft_round<Fr, RMAX, true>(lds, logS, 0, log_cols, tw, tw_step);
  unsigned s = RMAX;
  for (; s + RMAX <= logS; s += RMAX) lds_fft_round<Fr, RMAX, false>(lds, logS, s, log_cols, tw, tw_step);
  lds_fft_tail<Fr, RMAX>(lds, logS, s, log_cols, tw, tw_step);
}

ZG_HD uint32_t bitrev32(uint32_t x, unsigned bits) { return bits ? (__builtin_bitreverse32(x) >> (32 - bits)) : 0; }

// Both passes move a tile between HBM and LDS with 2^RMAX elements per thread, all of a thread's loads in flight at once.
// blockIdx.y selects the vector of a batch (the prover transforms its three evaluation vectors in one launch);
// consecutive vectors are `vstride` elements apart.

// pass A: grid = (N2 / TC, batch) blocks, TC = 2^log_tc adjacent columns per tile
template <class Fr, bool COSET_IN, int RMAX>
__global__ void __launch_bounds__(ntt_max_threads(RMAX)) k_ntt_pass_a(Fr* __restrict__ a, NttTables<Fr> T, int inverse, unsigned log_tc,
                                                             size_t vstride) {
  extern __shared__ uint4 lds[];
  constexpr uint32_t PER = 1u << RMAX;
  a += (size_t)blockIdx.y * vstride;
  const uint32_t N2 = 1u << T.log_n2;
  const uint32_t half_m = (1u << T.log_m) >> 1;
  const Fr* tw = inverse ? T.tw_inv : T.tw;
  const uint32_t c0 = blockIdx.x << log_tc;
  const uint32_t total = 1u << (T.log_n1 + log_tc);
  const uint32_t cmask = (1u << log_tc) - 1u;
  for (uint32_t i0 = threadIdx.x; i0 < total; i0 += PER * blockDim.x) {
    Fr x[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t idx = i0 + k * blockDim.x;
      if (idx < total) x[k] = glob_get(a + (size_t)(idx >> log_tc) * N2 + c0 + (idx & cmask));
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t idx = i0 + k * blockDim.x;
      if (idx < total) {
        const uint32_t j1 = idx >> log_tc, c = idx & cmask;
        if (COSET_IN) x[k] = Fr::mul(Fr::mul(x[k], glob_get(T.g_lo + c0 + c)), glob_get(T.g_hi + j1));
        lds_put<Fr>(lds, (bitrev32(j1, T.log_n1) << log_tc) + c, x[k]);
      }
    }
  }
  __syncthreads();
  lds_fft<Fr, RMAX>(lds, T.log_n1, log_tc, tw, N2);
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t k1 = idx >> log_tc, c = idx & cmask;
    Fr x = lds_get<Fr>(lds, idx);
    const uint32_t e = (c0 + c) * k1;  // < m
    if (e != 0 && N2 > 1) x = Fr::mul(x, tw_lookup(tw, e, half_m ? half_m : 1));
    glob_put(a + (size_t)k1 * N2 + c0 + c, x);
  }
}

// pass B: grid = (N1 / TR, batch) blocks, TR = 2^log_tr adjacent rows per tile.  OUT_MODE 0: none, 1: * m^-1,
// 2: * g^-k m^-1 (two-level tables)
template <class Fr, int OUT_MODE, int RMAX>
__global__ void __launch_bounds__(ntt_max_threads(RMAX)) k_ntt_pass_b(const Fr* __restrict__ a, Fr* __restrict__ out, NttTables<Fr> T,
                                                             int inverse, unsigned log_tr, size_t vstride) {
  extern __shared__ uint4 lds[];
  constexpr uint32_t PER = 1u << RMAX;
  a += (size_t)blockIdx.y * vstride;
  out += (size_t)blockIdx.y * vstride;
  const uint32_t N1 = 1u << T.log_n1, N2 = 1u << T.log_n2;
  const Fr* tw = inverse ? T.tw_inv : T.tw;
  const uint32_t r0 = blockIdx.x << log_tr;
  const uint32_t total = 1u << (T.log_n2 + log_tr);
  const uint32_t rmask = (1u << log_tr) - 1u;
  for (uint32_t i0 = threadIdx.x; i0 < total; i0 += PER * blockDim.x) {
    Fr x[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t idx = i0 + k * blockDim.x;
      if (idx < total) x[k] = glob_get(a + (size_t)(r0 + (idx >> T.log_n2)) * N2 + (idx & (N2 - 1u)));
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t idx = i0 + k * blockDim.x;
      if (idx < total) lds_put<Fr>(lds, (bitrev32(idx & (N2 - 1u), T.log_n2) << log_tr) + (idx >> T.log_n2), x[k]);
    }
  }
  __syncthreads();
  lds_fft<Fr, RMAX>(lds, T.log_n2, log_tr, tw, N1);
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t k2 = idx >> log_tr, r = idx & rmask;
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
