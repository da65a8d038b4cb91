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
// LDS (up to 4096 elements per workgroup) and runs all of their radix-2 stages there:
//   pass A  tiles of TC adjacent columns: N1-point transforms down the columns (stride-N2 loads of TC*32-byte
//           runs), then the inter-pass twiddle w^(j2 k1), written back in place
//   pass B  tiles of TR adjacent rows: N2-point transforms along contiguous rows, fused output scaling
//           (m^-1 and/or g^-k), written transposed (TR*32-byte runs) to the output buffer
// Coset pre-scaling g^j is fused into pass A's load through two small tables g^(j1 N2), g^(j2).
//
// Arithmetic.  The transforms are bound by the field multiplier (profiles/r2_ntt_radix.txt: the time does not move with
// the number of LDS round trips), so between the load and the store of a pass the elements live in the reduced-radix
// form of field29.cuh: nine 29-bit limbs (36 bytes in LDS), Montgomery radix R' = 2^261.  A butterfly is one product
// b w (column accumulators, no carry instructions), a limb-wise a + bw and a limb-wise a - bw + 3p: no carry chains and
// no conditional subtractions.  Values are lazy: the bound grows by 3p per stage (a stage's "a" input is never
// multiplied) from 32p at the load, which stays below the 168p (alt_bn128; 438p BLS12-377) the multiplier admits over
// the 12 stages a pass can have -- checked at compile time, see Ntt29.  Conversions cost nothing extra: the load reads
// the stored x R shifted left by 5 bits (= x R'), and the product every element receives on its way out anyway (the
// inter-pass twiddle, m^-1, the coset factor, or R mod p when there is none) is taken against the STORED form of that
// factor, which lands the result in the stored form x R again.
#pragma once
#include <hip/hip_runtime.h>

#include "field.cuh"
#include "field29.cuh"
#include "knobs.h"

namespace zg {

static constexpr int NTT_LDS_ELEMS = 4096;  // 144 KB of 36-byte elements
static constexpr int NTT_THREADS = 1024;
static constexpr int NTT_MAX_STAGES = 12;   // log2 NTT_LDS_ELEMS: the stages one pass can run
constexpr int ntt_max_threads(int) { return NTT_THREADS; }

template <class Fr>
struct NttTables {
  unsigned log_m;
  const Fr* tw;      // w^i, i < m/2 (forward) -- stored (Montgomery R) form: the inter-pass twiddles
  const Fr* tw_inv;  // w^-i
  // two-level coset tables, N1 + N2 entries each
  const Fr* g_hi;    // g^(j1 N2), j1 < N1      (input scaling of cosetFFT; the kernels read g_hi29)
  const Fr* g_lo;    // g^(j2), j2 < N2
  const Fr* gi_lo;   // g^(-k1) * m^-1, k1 < N1  (output scaling of icosetFFT)
  const Fr* gi_hi;   // g^(-N1 k2), k2 < N2
  // the butterflies' twiddles and g_hi in the reduced-radix form: 9 words per entry, x R' mod p in [0, p).  One compact
  // table per pass and direction -- (w^N2)^j, j < N1/2 for pass A and (w^N1)^j, j < N2/2 for pass B -- so that the
  // entries a workgroup gathers are neighbours in memory (36 KB for a 2048-point pass) instead of a stride apart
  const uint32_t* twa29;
  const uint32_t* twa_inv29;
  const uint32_t* twb29;
  const uint32_t* twb_inv29;
  const uint32_t* g_hi29;
  Fr m_inv;
  unsigned log_n1, log_n2;
  // Full-size tables (round 3; null: the two-level forms above are used -- the allocation-free entry point, whose static
  // scratch has no room for them).  The transforms are bound by the field multiplier, not by memory, so every place that
  // built a factor from two small tables with two products now reads it from an m-entry table laid out like the data
  // (coalesced, 32 bytes an element) and pays one: the inter-pass twiddle, the coset input and the coset output.
  const Fr* twf;      // [k1][col] = w^(col k1), stored form: the inter-pass twiddle of pass A
  const Fr* twf_inv;  // the same for w^-1
  const Fr* in_tab;   // [j] = g^j R' (the stored form of 32 g^j): the coset input scaling of pass A
  const Fr* out_tab;  // [j] = the factor of output j in the stored form, OUT_MODE 3 of pass B (set per launch: g^-j m^-1
                      //       for icosetFFT, or g^j m^-1 for an iFFT whose result feeds a coset FFT)
  Fr qap_c;           // QAP_IN: 1024 / Z(g) in the stored form (see k_ntt_pass_a)
};

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

// Types and bounds of the reduced-radix transform.  An element in LDS is typed XL: value < VW p, limbs <= LL.
//   load            value < V0 p = 32 p (unpack), or < 2 p behind the coset products
//   first round     its first stage has only unit twiddles and skips the products: a + b < 64 p, a - b + 33 p < 65 p;
//                   its second stage (radix 4) multiplies every pair (unit twiddles by the constant one): + 3 p
//   later stages    + 3 p each (b w < 2 p: a + bw, a - bw + 3 p)
// so the input of the round that starts at stage s is below VFIRST + 3 (s - 1) p <= VW p, and the output of the last
// round below VE p.  A round is compiled ONCE, for an input of type XL, and proves that its output is within
// XL::vb + 3 R; narrowing that back to XL (assume_bounds) is justified by the count above -- at most NTT_MAX_STAGES
// stages -- which the static_asserts here tie to the multiplier's bound.
template <class Fr>
struct Ntt29 {
  using P = typename Fr::Params;
  using K = K29<P>;
  static constexpr int NL = K::NL;
  static constexpr uint64_t V0 = 1ull << K::SHIFT;
  static constexpr uint64_t VFIRST = 2 * V0 + 1 + 3;
  static constexpr uint64_t VW = VFIRST + 3 * (NTT_MAX_STAGES - 1);
  static constexpr uint64_t VE = VW + 3 * 2;
  static constexpr uint64_t LL = K::LNORM + 5;
  static_assert(VE <= K::mont_room(), "a pass of 12 stages would leave the multiplier's bound");
  using W = E29<P, 1, K::LNORM>;    // a twiddle: canonical
  using X0 = E29<P, V0, K::LNORM>;  // fresh from unpack
  using XL = E29<P, VW, LL>;        // in LDS between rounds
  using XE = E29<P, VE, LL>;        // in LDS after the last round
};

// An element is 9 words, so two positions share their banks exactly when they are a multiple of 64 apart -- which the
// bit-reversed scatter of the load and the first rounds' strides produce all the time.  Position p therefore lives in
// slot p + p / 64: positions 64 k apart land 65 k apart (64 extra slots for 4096 elements).
static constexpr int NTT_LDS_SLOTS = NTT_LDS_ELEMS + NTT_LDS_ELEMS / 64;
__device__ __forceinline__ uint32_t lds_slot(uint32_t pos) { return pos + (pos >> 6); }

template <class X>
__device__ __forceinline__ X lds_get29(const uint32_t* lds, uint32_t pos) {
  X x;
  const uint32_t* p = lds + lds_slot(pos) * (uint32_t)X::K::NL;
#pragma unroll
  for (int i = 0; i < X::K::NL; i++) x.v[i] = p[i];
  return x;
}
template <class X>
__device__ __forceinline__ void lds_put29(uint32_t* lds, uint32_t pos, const X& x) {
  uint32_t* p = lds + lds_slot(pos) * (uint32_t)X::K::NL;
#pragma unroll
  for (int i = 0; i < X::K::NL; i++) p[i] = x.v[i];
}
template <class Fr>
__device__ __forceinline__ typename Ntt29<Fr>::W tw_get29(const uint32_t* __restrict__ t, size_t i) {
  typename Ntt29<Fr>::W w;
  const uint32_t* p = t + i * (size_t)Ntt29<Fr>::NL;
#pragma unroll
  for (int k = 0; k < Ntt29<Fr>::NL; k++) w.v[k] = p[k];
  return w;
}

// (a, b) -> (a + b w, a - b w), both under the difference's (wider) bounds
template <class XA, class XB, class WT>
__device__ __forceinline__ void bfly29(const XA& a, const XB& b, const WT& w, decltype(f29::sub(a, f29::mul(b, w)))& lo,
                                       decltype(f29::sub(a, f29::mul(b, w)))& hi) {
  using D = decltype(f29::sub(a, f29::mul(b, w)));
  const auto t = f29::mul(b, w);
  lo = f29::widen<D::vb, D::lmax>(f29::add(a, t));
  hi = f29::sub(a, t);
}
// the same with w = 1 and no product
template <class X>
__device__ __forceinline__ void bfly29_unit(const X& a, const X& b, decltype(f29::sub(a, b))& lo, decltype(f29::sub(a, b))& hi) {
  using D = decltype(f29::sub(a, b));
  lo = f29::widen<D::vb, D::lmax>(f29::add(a, b));
  hi = f29::sub(a, b);
}

// `cols` = 2^log_cols interleaved transforms of size 2^logS held in LDS as lds[pos * cols + c], input in bit-reversed
// position order, output in natural order (decimation in time).  tw: this size's twiddles, S/2 entries.  The stages run R at a time in registers (R = 2 where the transform has lanes to spare): a work item owns
// the 2^R elements base + t 2^s of one column, does the stages s .. s+R-1 on them and writes them back -- one LDS round
// trip and one barrier per R stages.  XIN: the type of the elements this round finds in LDS (Ntt29).
template <class Fr, int R, bool FIRST, class XIN>
__device__ __forceinline__ void lds_fft_round(uint32_t* lds, unsigned logS, unsigned s, unsigned log_cols,
                                              const uint32_t* __restrict__ tw) {
  using N = Ntt29<Fr>;
  using XL = typename N::XL;
  static_assert(R == 1 || R == 2, "rounds of one or two stages");
  constexpr uint32_t E = 1u << R;
  const uint32_t S = 1u << logS;
  const uint32_t items = (S >> R) << log_cols;
  const uint32_t cmask = (1u << log_cols) - 1u;
  for (uint32_t idx = threadIdx.x; idx < items; idx += blockDim.x) {
    const uint32_t g = idx >> log_cols, c = idx & cmask;
    const uint32_t j0 = g & ((1u << s) - 1u);
    const uint32_t base = ((g >> s) << (s + R)) + j0;
    XIN x[E];
#pragma unroll
    for (uint32_t t = 0; t < E; t++) x[t] = lds_get29<XIN>(lds, ((base + (t << s)) << log_cols) + c);
    // stage s: pairs (t, t + 1), twiddle exponent j0 (zero in the first round: s == 0)
    const uint32_t unit0 = S >> (s + 1);
    if constexpr (FIRST) {
      using D0 = decltype(f29::sub(x[0], x[0]));
      D0 y[E];
#pragma unroll
      for (uint32_t t = 0; t < E; t += 2) bfly29_unit(x[t], x[t + 1], y[t], y[t + 1]);
      if constexpr (R == 1) {
#pragma unroll
        for (uint32_t t = 0; t < E; t++) {
          const auto o = f29::carry(y[t]);
          static_assert(decltype(o)::vb <= N::VFIRST && decltype(o)::lmax <= N::LL, "first round leaves its bounds");
          lds_put29(lds, ((base + (t << s)) << log_cols) + c, f29::widen<XL::vb, XL::lmax>(o));
        }
      } else {
        // stage 1: pairs (0, 2) with the unit twiddle -- multiplied all the same, to bring that element below 2 p --
        // and (1, 3) with w^(S/4)
        const uint32_t unit1 = S >> 2;
        using D1 = decltype(f29::sub(y[0], f29::mul(y[0], f29::one<typename N::P>())));
        D1 z[E];
        bfly29(y[0], y[2], f29::one<typename N::P>(), z[0], z[2]);
        bfly29(y[1], y[3], tw_get29<Fr>(tw, unit1), z[1], z[3]);
#pragma unroll
        for (uint32_t t = 0; t < E; t++) {
          const auto o = f29::carry(z[t]);
          static_assert(decltype(o)::vb <= N::VFIRST && decltype(o)::lmax <= N::LL, "first round leaves its bounds");
          lds_put29(lds, ((base + (t << s)) << log_cols) + c, f29::widen<XL::vb, XL::lmax>(o));
        }
      }
    } else {
      using D0 = decltype(f29::sub(x[0], f29::mul(x[0], typename N::W{})));
      D0 y[E];
      const typename N::W w0 = tw_get29<Fr>(tw, (size_t)j0 * unit0);
#pragma unroll
      for (uint32_t t = 0; t < E; t += 2) bfly29(x[t], x[t + 1], w0, y[t], y[t + 1]);
      if constexpr (R == 1) {
#pragma unroll
        for (uint32_t t = 0; t < E; t++) {
          const auto o = f29::carry(y[t]);
          static_assert(decltype(o)::vb <= XIN::vb + 3 * R && decltype(o)::lmax <= N::LL, "round grows by more than 3 p a stage");
          // at most NTT_MAX_STAGES stages run in a pass: see Ntt29
          lds_put29(lds, ((base + (t << s)) << log_cols) + c, f29::assume_bounds<XL::vb, XL::lmax>(o));
        }
      } else {
        // stage s + 1: pairs (t, t + 2), exponent j0 + t 2^s, scaled by S / 2^(s+2)
        const uint32_t unit1 = S >> (s + 2);
        using D1 = decltype(f29::sub(y[0], f29::mul(y[0], typename N::W{})));
        D1 z[E];
        bfly29(y[0], y[2], tw_get29<Fr>(tw, (size_t)j0 * unit1), z[0], z[2]);
        bfly29(y[1], y[3], tw_get29<Fr>(tw, (size_t)(j0 + (1u << s)) * unit1), z[1], z[3]);
#pragma unroll
        for (uint32_t t = 0; t < E; t++) {
          const auto o = f29::carry(z[t]);
          static_assert(decltype(o)::vb <= XIN::vb + 3 * R && decltype(o)::lmax <= N::LL, "round grows by more than 3 p a stage");
          lds_put29(lds, ((base + (t << s)) << log_cols) + c, f29::assume_bounds<XL::vb, XL::lmax>(o));
        }
      }
    }
  }
  __syncthreads();
}

// V0IN: the elements the load left in LDS are below V0IN p (Ntt29::V0, or 2 behind the coset products)
template <class Fr, int RMAX>
__device__ __forceinline__ void lds_fft(uint32_t* lds, unsigned logS, unsigned log_cols, const uint32_t* __restrict__ tw) {
  using N = Ntt29<Fr>;
  using X0 = typename N::X0;
  using XL = typename N::XL;
  if (logS == 0) return;
  if (logS < (unsigned)RMAX) {  // the whole transform is shorter than one round
    lds_fft_round<Fr, 1, true, X0>(lds, logS, 0, log_cols, tw);
    return;
  }
  lds_fft_round<Fr, RMAX, true, X0>(lds, logS, 0, log_cols, tw);
  unsigned s = RMAX;
  for (; s + RMAX <= logS; s += RMAX) lds_fft_round<Fr, RMAX, false, XL>(lds, logS, s, log_cols, tw);
  if constexpr (RMAX == 2) {
    if (logS - s == 1) lds_fft_round<Fr, 1, false, XL>(lds, logS, s, log_cols, tw);
  }
}

ZG_HD uint32_t bitrev32(uint32_t x, unsigned bits) { return bits ? (__builtin_bitreverse32(x) >> (32 - bits)) : 0; }

// Both passes move a tile between HBM and LDS with 2^RMAX elements per thread, all of a thread's loads in flight at once.
// blockIdx.y selects the vector of a batch (the prover transforms its three evaluation vectors in one launch);
// consecutive vectors are `vstride` elements apart.

// Tiles narrower than a 128-byte line (4 elements) share their lines with the neighbouring tiles: pass A's column reads
// and pass B's transposed writes then touch every line 4 / width times.  Workgroups are dealt to the 8 XCDs in turn
// (block b runs on XCD b mod 8, each XCD with its own L2), so the tiles that share lines are given to blocks b, b + 8,
// ... -- neighbours in time on ONE XCD, where the second touch is an L2 hit.
__device__ __forceinline__ uint32_t ntt_tile_of_block(uint32_t b, uint32_t nblocks, unsigned log_width) {
  if (log_width >= 2) return b;
  const uint32_t share = 4u >> log_width;  // tiles per line: 2 or 4
  if (nblocks % (8u * share)) return b;
  const uint32_t xcd = b & 7u, slot = b >> 3;
  return ((slot / share) * 8u + xcd) * share + (slot % share);
}

// pass A: grid = (N2 / TC, batch) blocks, TC = 2^log_tc adjacent columns per tile.
// IN_MODE 0: the vector as it is; 1: coset input, x g^j; 2 ("QAP_IN", batch 1): the element is built from THREE vectors
// a, a + vstride, a + 2 vstride (the evaluations of A, B, C on the coset) as (A B - C) / Z(g) -- the pointwise step of
// r1cs_to_qap_witness_map between the coset FFTs and the final icosetFFT, which used to be a kernel of its own
// (read three vectors, write one).  The result of the pass is written over the first vector.
template <class Fr, int IN_MODE, int RMAX>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass_a(Fr* __restrict__ a, NttTables<Fr> T, int inverse, unsigned log_tc,
                                                            size_t vstride) {
  ZG_WAVE_PRIO(2);
  constexpr bool COSET_IN = IN_MODE == 1;
  constexpr bool QAP_IN = IN_MODE == 2;
  extern __shared__ uint32_t lds[];
  using N = Ntt29<Fr>;
  using P = typename N::P;
  constexpr uint32_t PER = 1u << RMAX;
  if (!QAP_IN) a += (size_t)blockIdx.y * vstride;
  const uint32_t N2 = 1u << T.log_n2;
  const Fr* tw = inverse ? T.tw_inv : T.tw;
  const Fr* twf = inverse ? T.twf_inv : T.twf;
  const uint32_t* tw29 = inverse ? T.twa_inv29 : T.twa29;
  const uint32_t c0 = ntt_tile_of_block(blockIdx.x, gridDim.x, log_tc) << log_tc;
  const uint32_t total = 1u << (T.log_n1 + log_tc);
  const uint32_t cmask = (1u << log_tc) - 1u;
  for (uint32_t i0 = threadIdx.x; i0 < total; i0 += PER * blockDim.x) {
    Fr x[PER];
    Fr xb[QAP_IN ? PER : 1], xc[QAP_IN ? PER : 1];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t idx = i0 + k * blockDim.x;
      if (idx < total) {
        const size_t at = (size_t)(idx >> log_tc) * N2 + c0 + (idx & cmask);
        x[k] = glob_get(a + at);
        if constexpr (QAP_IN) {
          xb[k] = glob_get(a + vstride + at);
          xc[k] = glob_get(a + 2 * vstride + at);
        }
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t idx = i0 + k * blockDim.x;
      if (idx < total) {
        const uint32_t j1 = idx >> log_tc, c = idx & cmask;
        const uint32_t pos = (bitrev32(j1, T.log_n1) << log_tc) + c;
        if constexpr (QAP_IN) {
          // a R (as it is) against b R' (shifted read): a b R, below 2 p; minus c R; against the stored 1024 / Z: the product
          // lands on ((a b - c) / Z) 32 R = h R', the form the rounds work in
          const auto t = f29::mul(f29::unpack_raw(x[k]), f29::unpack(xb[k]));
          const auto u = f29::carry(f29::sub(t, f29::unpack_raw(xc[k])));
          const auto h = f29::mul(u, f29::unpack_raw(T.qap_c));
          lds_put29(lds, pos, f29::widen<N::X0::vb, N::X0::lmax>(h));
        } else if (COSET_IN && T.in_tab) {
          // x R' against the table's g^j R': x g^j R', below 2 p
          const auto u = f29::mul(f29::unpack(x[k]), f29::unpack_raw(glob_get(T.in_tab + (size_t)j1 * N2 + c0 + c)));
          lds_put29(lds, pos, f29::widen<N::X0::vb, N::X0::lmax>(u));
        } else if (COSET_IN) {
          // x R' g_hi R' / R' = x g_hi R' (< 2 p), then against g_lo R' (< 32 p): x g R' < 2 p
          const auto t = f29::mul(f29::unpack(x[k]), tw_get29<Fr>(T.g_hi29, j1));
          const auto u = f29::mul(t, f29::unpack(glob_get(T.g_lo + c0 + c)));
          lds_put29(lds, pos, f29::widen<N::X0::vb, N::X0::lmax>(u));
        } else {
          lds_put29(lds, pos, f29::unpack(x[k]));
        }
      }
    }
  }
  __syncthreads();
  lds_fft<Fr, RMAX>(lds, T.log_n1, log_tc, tw29);
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t k1 = idx >> log_tc, c = idx & cmask;
    const auto x = lds_get29<typename N::XE>(lds, idx);
    const uint32_t e = (c0 + c) * k1;  // < m
    // The inter-pass twiddle w^e, e = e_hi N2 + e_lo, from two small tables instead of a gather over the m/2-entry one
    // (one line of HBM traffic per element): x R' against (w^N2)^e_hi R' (pass A's own butterfly table; its upper half by
    // sign), then against the STORED w^e_lo R, which lands the product in the stored form (R mod p when there is no
    // twiddle).
    typename N::W r;
    if (twf) {
      // one product against the stored w^(col k1) (1 for the first row and column): the result is in the stored form
      r = f29::canon(f29::mul(x, f29::unpack_raw(glob_get(twf + (size_t)k1 * N2 + c0 + c))));
    } else if (e != 0 && N2 > 1) {
      const uint32_t e_hi = e >> T.log_n2, e_lo = e & (N2 - 1u), hn1 = (1u << T.log_n1) >> 1;
      const auto t = f29::mul(x, tw_get29<Fr>(tw29, e_hi & (hn1 - 1u)));
      r = f29::canon(f29::mul(t, f29::unpack_raw(glob_get(tw + e_lo))));
      if (e_hi & hn1) r = f29::neg_canon(r);  // (w^N2)^(N1/2) = -1
    } else {
      r = f29::canon(f29::mul(x, f29::r_conv<P>()));
    }
    glob_put(a + (size_t)k1 * N2 + c0 + c, f29::to_words(r));
  }
}

// pass B: grid = (N1 / TR, batch) blocks, TR = 2^log_tr adjacent rows per tile.  OUT_MODE 0: none, 1: * m^-1,
// 2: * g^-k m^-1 (two-level tables), 3: * out_tab[k] (full table, one product)
template <class Fr, int OUT_MODE, int RMAX>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass_b(const Fr* __restrict__ a, Fr* __restrict__ out, NttTables<Fr> T,
                                                            int inverse, unsigned log_tr, size_t vstride) {
  ZG_WAVE_PRIO(2);
  extern __shared__ uint32_t lds[];
  using N = Ntt29<Fr>;
  using P = typename N::P;
  constexpr uint32_t PER = 1u << RMAX;
  a += (size_t)blockIdx.y * vstride;
  out += (size_t)blockIdx.y * vstride;
  const uint32_t N1 = 1u << T.log_n1, N2 = 1u << T.log_n2;
  const uint32_t* tw29 = inverse ? T.twb_inv29 : T.twb29;
  const uint32_t r0 = ntt_tile_of_block(blockIdx.x, gridDim.x, log_tr) << log_tr;
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
      if (idx < total) lds_put29(lds, (bitrev32(idx & (N2 - 1u), T.log_n2) << log_tr) + (idx >> T.log_n2), f29::unpack(x[k]));
    }
  }
  __syncthreads();
  lds_fft<Fr, RMAX>(lds, T.log_n2, log_tr, tw29);
  for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const uint32_t k2 = idx >> log_tr, r = idx & rmask;
    const auto x = lds_get29<typename N::XE>(lds, idx);
    typename N::W y;
    if (OUT_MODE == 0) y = f29::canon(f29::mul(x, f29::r_conv<P>()));
    if (OUT_MODE == 1) y = f29::canon(f29::mul(x, f29::unpack_raw(T.m_inv)));
    if (OUT_MODE == 3) y = f29::canon(f29::mul(x, f29::unpack_raw(glob_get(T.out_tab + (size_t)k2 * N1 + r0 + r))));
    if (OUT_MODE == 2)  // x R' against the stored gi_lo R: x gi_lo R (< 2 p); that against gi_hi R' (< 32 p): x gi R
      y = f29::canon(f29::mul(f29::mul(x, f29::unpack_raw(glob_get(T.gi_lo + r0 + r))), f29::unpack(glob_get(T.gi_hi + k2))));
    glob_put(out + (size_t)k2 * N1 + r0 + r, f29::to_words(y));
  }
}

// dst[9 i ..] = the reduced-radix form of src[i * stride]: x R' mod p in [0, p)
template <class Fr>
__global__ void __launch_bounds__(256) k_to_radix29(const Fr* __restrict__ src, uint32_t stride, uint32_t* __restrict__ dst,
                                                    uint32_t count) {
  using P = typename Fr::Params;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const auto w = f29::canon(f29::mul(f29::unpack(glob_get(src + (size_t)i * stride)), f29::one<P>()));
  uint32_t* p = dst + (size_t)i * K29<P>::NL;
#pragma unroll
  for (int k = 0; k < K29<P>::NL; k++) p[k] = w.v[k];
}

// ---- table construction -------------------------------------------------------------------------
// twf[k1 * N2 + col] = w^(col k1) from the half table tw[i] = w^i, i < m/2 (w^(m/2) = -1)
template <class Fr>
__global__ void __launch_bounds__(256) k_twf_fill(const Fr* __restrict__ tw, unsigned log_m, unsigned log_n2, Fr* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t m = (size_t)1 << log_m;
  if (idx >= m) return;
  const size_t k1 = idx >> log_n2, col = idx & (((size_t)1 << log_n2) - 1);
  const size_t e = (k1 * col) & (m - 1);
  const size_t half = m > 1 ? m / 2 : 1;
  Fr v = glob_get(tw + (e < half ? e : e - half));
  if (e >= half && m > 1) v = Fr::neg(v);
  glob_put(out + idx, v);
}

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
