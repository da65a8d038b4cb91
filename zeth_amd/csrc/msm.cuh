// Pippenger / bucket multi-scalar multiplication for gfx950.
//
// Replaces libff::multi_exp / multi_exp_with_mixed_addition / multi_exp_filter_one_zero with method
// BDLO12 (absent submodule; Zeth's call sites libzeth/core/multi_exp.tcc:20-22,32-34 and, inside the
// libsnark prover, the A/B/H/L query evaluations -- SURVEY.md section 3.1 step 5c).  The result is the group
// element sum_i f_i * g_i, returned as its unique affine point, so it is bit-identical to any correct
// evaluation order.
//
// Pipeline (all on the device, one HIP stream):
//   k_digits   scalar -> W signed c-bit digits (int16), via s + sum_w 2^(c-1) 2^(wc) then per-window re-centre
//   k_hist     per (chunk, window) LDS histogram of |digit| (128 KB of LDS at c = 16) -> global counts
//   k_scan     per window exclusive scan -> bucket offsets, work-unit offsets (a unit = <= L entries of one bucket)
//   k_scatter  counting-sort scatter of (point index | sign) through LDS cursors; from 512 buckets per window the
//              two-level form k_scatter_hi / k_scatter_lo (tiles ranked in LDS, written out in runs)
//   k_units, k_units_heavy   unit table and the longest-first order of the units
//   k_accum    one lane per unit: gather affine bases (64-byte AoS records), XYZZ mixed add      <- dominant
//   k_combine_light, k_combine_tree   fold the units of split buckets into their first unit
//   k_reduce1  per (window, bucket segment): running sums, weight by segment start
//   k_reduce2  per window tree reduction -> W window sums
// Host tail: Horner over the W window sums (c doublings each) and normalisation to affine.
// Fixed bases with a table of precomputed multiples: one bucket set for all digits, msm_table.cuh.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <utility>
#include <vector>

#include "accum29.cuh"
#include "accum29x2.cuh"
#include "curve.cuh"
#include "curve_quad.cuh"
#include "curve_quad29.cuh"
#include "knobs.h"

namespace zg {

static constexpr int MSM_MAX_C = 16;
// units of a bucket folded serially by the first unit's lane (MsmPlan::smax); more: tree (k_combine_tree).  The tree is
// built for a few buckets of thousands of units (the "1" of a witness); the thousands of moderately split buckets of a
// table-mode top digit (11 units at 2^25 points, 22 at 2^26) took it 8 ms against 0.4-0.8 ms folded serially.
static constexpr uint32_t COMBINE_SERIAL_MAX = 8, COMBINE_SERIAL_MAX_TABLE = 32;
static constexpr uint32_t COMBINE_CHUNK = 512;     // units per workgroup and launch of the tree
static constexpr uint32_t COMBINE_HEAVY_SLOTS = 32;  // heavy buckets the tree takes side by side (grid.y)
// Units are handed to the accumulation kernel longest first, by length class: class UNIT_CLASSES = full-length units,
// classes 0..1023 = partial units by (len - 1) * 1024 / L -- one class per length, since L <= 1024: the lanes of a wave
// run in lockstep to the longest unit among them, and classes a few entries wide cost the accumulation ~10 % at 2^24
// points (measured: 21.8 ms with classes 6 entries wide against 19.2 ms with exact lengths, same entries).
static constexpr uint32_t UNIT_CLASSES = 1024;
static constexpr uint32_t UNIT_CLASS_STRIDE = 1280;
__host__ __device__ inline uint32_t unit_class(uint32_t len, uint32_t L) { return ((len - 1) * UNIT_CLASSES) / L; }
static constexpr int MSM_MAX_W = 48;

struct AddConst {
  uint32_t k[9];
};

struct UnitDesc {
  uint64_t start;   // absolute index into sorted[]
  uint32_t len;     // entries in this unit
  uint32_t bucket;  // w * nb + b
};

struct MsmPlan {
  size_t n;
  int c, W;
  uint32_t nb;     // buckets per window = 2^(c-1); bucket b has weight b + 1
  uint32_t L;      // max entries per work unit
  uint32_t cap;    // unit slots per window
  uint32_t nchunk; // histogram/scatter chunks per window
  uint32_t T;      // bucket segments per window (reduce1 lanes per window)
  uint32_t k;      // buckets per segment
  uint32_t smax;   // units folded serially (COMBINE_SERIAL_MAX or COMBINE_SERIAL_MAX_TABLE)
  AddConst K;
  // fixed-base table mode (msm_make_table_plan): c = digit width (up to 22), D = digits per scalar, all digits share ONE
  // set of 2^(c-1) buckets, cut into W "bucket groups" of nb <= 2^15 buckets that the sort, the unit table, the
  // accumulation and the reduction treat the way they treat the windows of the plain mode; E = n * D entries
  int table, D, shift /* log2 nb */, gbits /* log2 W */;
  size_t E;
};

inline int ilog2_floor(size_t n) {
  int l = 0;
  while ((n >> (l + 1)) != 0) l++;
  return l;
}

// The multiples 2^(j c) P_i, j < D, of a fixed-base table: D for a scalar field of `scalar_bits` bits
constexpr int msm_table_digits(int scalar_bits, int c) { return (scalar_bits + 2 + c - 1) / c; }
static constexpr int MSM_TABLE_MIN_C = 10, MSM_TABLE_MAX_C = 22;

// Digit width of the table for N fixed bases, measured on MI355X (BN254 G1, tools/bench_sweep.py g1t): 22 from 2^23
// bases (12 digits, 2^21 buckets), 20 below (13 digits, 2^19 buckets: best at every size from 2^16 to 2^22 -- 2^20 points
// 2.56 ms against 3.1 ms at c = 17 and 3.0 ms windowed).  Widths whose top digit is nearly empty are avoided: at c = 21
// the 13th digit has 2 bits, four buckets take a quarter of all scalars each, and the accumulation runs 60 % longer.
inline int msm_table_pick_c(size_t N) {
  int lg = ilog2_floor(N ? N : 1) + ((N & (N - 1)) ? 1 : 0);
  // (up to 2^16 bases 17 bits: two more additions per scalar, but an eighth of the buckets to reduce -- 0.50 against
  // 0.57 ms of kernels at 2^16 bases, a tie at 2^17, 20 bits ahead from 2^18: profiles/r3_tune_table_c.jsonl)
  return lg >= 23 ? 22 : (lg <= 16 ? 17 : 20);
}

// window width for n scalars that are not 0 or 1, tuned on MI355X (tools/tune_c.py): odd widths keep the
// per-window reduction short
inline int msm_pick_c(size_t n) {
  int lg = ilog2_floor(n ? n : 1) + ((n & (n - 1)) ? 1 : 0);
  int c;
  if (lg <= 12) c = lg - 3;
  else if (lg <= 16) c = 13;
  else if (lg <= 20) c = 15;
  else c = 16;
  if (c < 6) c = 6;
  if (c > MSM_MAX_C) c = MSM_MAX_C;
  return c;
}

// The reduction runs one quad per segment: beyond about two waves a SIMD (32768 quads) more segments only add the c-bit
// weighting of each one to the work (2^24 points, c = 22, 64 groups: 2048 segments a group 1.39 ms, 512 segments 1.00 ms)
inline void msm_cap_segments(MsmPlan& p) {
  while (p.T > 256 && (size_t)p.T * (size_t)p.W > 40000) p.T >>= 1;
  // ... and with few windows or groups, more and shorter segments while the quads stay within a wave per SIMD: the chain
  // of a segment (2 k additions + the weighting) is the latency of the whole reduction then
  // (measurement knob: with few windows or groups, more and shorter segments up to this many quads -- inside a proof it
  // made no difference: c = 17, 2048 / 4096 / 8192 segments a group 384 / 388 / 388 proofs/s)
  if (const char* e = zg_knob("ZG_MSM_MORE_SEGMENTS")) {
    const size_t lim = strtoul(e, nullptr, 0);
    while (p.T < p.nb / 4 && (size_t)p.T * (size_t)p.W * 2 <= lim) p.T <<= 1;
  }
}

inline MsmPlan msm_make_plan(size_t n, int scalar_bits, int force_c = 0) {
  MsmPlan p{};
  p.n = n;
  p.smax = COMBINE_SERIAL_MAX;
  int c = msm_pick_c(n);
  if (force_c) c = force_c;
  p.c = c;
  // need W*c >= scalar_bits + 2 so that s + K never overflows W windows
  p.W = (scalar_bits + 2 + c - 1) / c;
  p.nb = 1u << (c - 1);
  // Unit length: long enough that the pairwise combine tree stays small next to the accumulation, short enough
  // that (i) at least about 2^18 units are in flight and (ii) one lane's chain of dependent mixed additions (~4.5 us each
  // when a wave runs alone) does not set the kernel's duration when a few buckets hold most entries.
  size_t L = ((size_t)p.W * n) >> 18;
  if (L < 32) L = 32;
  if (L > 1024) L = 1024;
  if (const char* e = zg_knob("ZG_MSM_UNIT_LEN")) {  // tuning knob
    size_t v = strtoul(e, nullptr, 0);
    if (v >= 1 && v <= 1024) L = v;
  }
  p.L = (uint32_t)L;
  p.cap = p.nb + (uint32_t)(n / L) + 1;
  uint32_t nchunk = (uint32_t)((n + (1u << 16) - 1) >> 16);  // >= 64K entries per chunk
  if (nchunk < 1) nchunk = 1;
  uint32_t maxchunk = 512u / (uint32_t)p.W;  // about two blocks per CU in total
  if (maxchunk < 1) maxchunk = 1;
  if (nchunk > maxchunk) nchunk = maxchunk;
  p.nchunk = nchunk;
  // bucket segments per window: a reduce1 lane runs 2k dependent additions plus a c-bit double-and-add (~7 us each),
  // reduce2 then folds the T segment sums of a window; measured best: 1024 segments up to 2^14 buckets, 2048 above
  p.T = p.nb < 1024 ? p.nb : (p.nb >= 32768 ? 2048 : 1024);
  msm_cap_segments(p);
  if (const char* e = zg_knob("ZG_MSM_SEGMENTS")) {  // tuning knob (power of two)
    uint32_t v = (uint32_t)strtoul(e, nullptr, 0);
    if (v >= 1 && v <= p.nb && (v & (v - 1)) == 0) p.T = v;
  }
  p.k = p.nb / p.T;
  for (int i = 0; i < 9; i++) p.K.k[i] = 0;
  for (int w = 0; w < p.W; w++) {
    int bit = w * c + (c - 1);
    p.K.k[bit / 32] |= 1u << (bit % 32);
  }
  return p;
}

inline MsmPlan msm_make_table_plan(size_t n, int scalar_bits, int c, int D) {
  MsmPlan p{};
  p.n = n;
  p.table = 1;
  p.smax = COMBINE_SERIAL_MAX_TABLE;
  p.c = c;
  p.D = D;
  p.E = n * (size_t)D;
  const int bucket_bits = c - 1;
  p.shift = bucket_bits < 15 ? bucket_bits : 15;
  p.nb = 1u << p.shift;
  p.gbits = bucket_bits - p.shift;
  p.W = 1 << p.gbits;
  size_t L = p.E >> 18;
  // at least a quarter of the mean bucket load: ordinary buckets then split into about four units (the serial fold of
  // k_combine_light), not into more than COMBINE_SERIAL_MAX (every bucket on the tree path: 46 ms at 2^19 points, c = 16)
  const size_t load4 = (p.E >> bucket_bits) / 4;
  if (L < load4) L = load4;
  if (L < 32) L = 32;
  if (L > 1024) L = 1024;
  if (const char* e = zg_knob("ZG_MSM_UNIT_LEN")) {
    size_t v = strtoul(e, nullptr, 0);
    if (v >= 1 && v <= 1024) L = v;
  }
  p.L = (uint32_t)L;
  // any group may hold all entries (skewed scalars), so every group's slot range is sized for that
  p.cap = p.nb + (uint32_t)(p.E / L) + 1;
  size_t per_group = p.E / (size_t)p.W;
  uint32_t nchunk = (uint32_t)((per_group + (1u << 16) - 1) >> 16);
  if (nchunk < 1) nchunk = 1;
  uint32_t maxchunk = 512u / (uint32_t)p.W;
  if (maxchunk < 1) maxchunk = 1;
  if (nchunk > maxchunk) nchunk = maxchunk;
  p.nchunk = nchunk;
  // Segments cost no weighting in table mode (k_bitsum), so a group is cut into as many as keep the device at one to two
  // waves of quads per SIMD (32768 quads in all; more only lengthens the tree over the segment sums), down to two
  // buckets a segment: 2^16 buckets (c = 17) run as 16384 segments of 4 -- a chain of 8 additions -- instead of 4096 of 16
  p.T = p.nb < 1024 ? p.nb : (p.nb >= 32768 ? 2048 : 1024);
  while (p.T * 4 <= p.nb && (size_t)p.T * 2 * (size_t)p.W <= 32768) p.T <<= 1;
  msm_cap_segments(p);
  if (const char* e = zg_knob("ZG_MSM_SEGMENTS")) {
    uint32_t v = (uint32_t)strtoul(e, nullptr, 0);
    if (v >= 1 && v <= p.nb && (v & (v - 1)) == 0) p.T = v;
  }
  p.k = p.nb / p.T;
  for (int i = 0; i < 9; i++) p.K.k[i] = 0;
  for (int w = 0; w < D; w++) {
    int bit = w * c + (c - 1);
    p.K.k[bit / 32] |= 1u << (bit % 32);
  }
  (void)scalar_bits;
  return p;
}

// ---------------------------------------------------------------------------------------------
// scalar recoding
// ---------------------------------------------------------------------------------------------
// signed digits d_j in [-2^(c-1), 2^(c-1)), sum_j d_j 2^(j c) = s, by adding K = sum_j 2^(j c + c - 1) first; emit(j, d)
template <class Fr, class Emit>
__device__ __forceinline__ void recode_scalar(const uint32_t* __restrict__ scalars, size_t i, int mont, int c, int D,
                                              const AddConst& K, bool* dense, Emit emit) {
  Fr s;
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
  uint4 lo = sp[0], hi = sp[1];
  s.v[0] = lo.x; s.v[1] = lo.y; s.v[2] = lo.z; s.v[3] = lo.w;
  s.v[4] = hi.x; s.v[5] = hi.y; s.v[6] = hi.z; s.v[7] = hi.w;
  if (mont) s = Fr::from_mont(s);
  uint32_t hi_bits = s.v[0] >> 1;
#pragma unroll
  for (int j = 1; j < 8; j++) hi_bits |= s.v[j];
  *dense = hi_bits != 0;
  uint32_t t[9];
  uint64_t cy = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    cy += (uint64_t)s.v[j] + K.k[j];
    t[j] = (uint32_t)cy;
    cy >>= 32;
  }
  t[8] = (uint32_t)cy + K.k[8];
  const uint32_t mask = (1u << c) - 1u;
  const int half = 1 << (c - 1);
  uint64_t buf = 0;
  int have = 0, w = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    buf |= (uint64_t)t[j] << have;
    have += 32;
    while (have >= c && w < D) {
      emit(w, (int)(buf & mask) - half);
      buf >>= c;
      have -= c;
      w++;
    }
  }
}

// The same digits for a digit width known at compile time: every digit sits at a fixed bit offset, so all of them stay in
// registers (dg[j], j < D) and the caller makes ONE pass over them
template <class Fr, int C, int D>
__device__ __forceinline__ void recode_scalar_static(const uint32_t* __restrict__ scalars, size_t i, int mont, const AddConst& K,
                                                     bool* dense, int (&dg)[D]) {
  Fr s;
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
  uint4 lo = sp[0], hi = sp[1];
  s.v[0] = lo.x; s.v[1] = lo.y; s.v[2] = lo.z; s.v[3] = lo.w;
  s.v[4] = hi.x; s.v[5] = hi.y; s.v[6] = hi.z; s.v[7] = hi.w;
  if (mont) s = Fr::from_mont(s);
  uint32_t hi_bits = s.v[0] >> 1;
#pragma unroll
  for (int j = 1; j < 8; j++) hi_bits |= s.v[j];
  *dense = hi_bits != 0;
  uint32_t t[9];
  uint64_t cy = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    cy += (uint64_t)s.v[j] + K.k[j];
    t[j] = (uint32_t)cy;
    cy >>= 32;
  }
  t[8] = (uint32_t)cy + K.k[8];
  static_assert(C * D <= 288 && C <= 32, "digits outside the 288-bit sum");
#pragma unroll
  for (int j = 0; j < D; j++) {
    const int bit = j * C, q = bit >> 5, sh = bit & 31;  // constants once the loop is unrolled
    const uint32_t lo = t[q], hi = q + 1 < 9 ? t[q + 1] : 0u;
    const uint32_t w = sh ? (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) : lo;
    dg[j] = (int)(w & (uint32_t)((1ull << C) - 1ull)) - (1 << (C - 1));
  }
}

template <class Fr>
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, size_t n, int mont, int c,
                                                 int W, AddConst K, int16_t* __restrict__ dig,
                                                 uint32_t* __restrict__ dense_count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;  // keep every lane in the block-wide count below; nothing is stored for it
  Fr s;
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
  uint4 lo = sp[0], hi = sp[1];
  s.v[0] = lo.x; s.v[1] = lo.y; s.v[2] = lo.z; s.v[3] = lo.w;
  s.v[4] = hi.x; s.v[5] = hi.y; s.v[6] = hi.z; s.v[7] = hi.w;
  if (mont) s = Fr::from_mont(s);
  {
    // scalars other than 0 and 1 ("dense" ones): the caller sizes the next call's windows by this count
    uint32_t hi_bits = s.v[0] >> 1;
#pragma unroll
    for (int j = 1; j < 8; j++) hi_bits |= s.v[j];
    if (dense_count) {  // only the prover asks for it (uniform branch)
      int cnt = __syncthreads_count(live && hi_bits != 0);
      if (threadIdx.x == 0 && cnt) atomicAdd(dense_count + (blockIdx.x & 63u), (uint32_t)cnt);
    }
  }
  if (!live) return;
  uint32_t t[9];
  uint64_t cy = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    cy += (uint64_t)s.v[j] + K.k[j];
    t[j] = (uint32_t)cy;
    cy >>= 32;
  }
  t[8] = (uint32_t)cy + K.k[8];
  const uint32_t mask = (1u << c) - 1u;
  const int half = 1 << (c - 1);
  uint64_t buf = 0;
  int have = 0, w = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    buf |= (uint64_t)t[j] << have;
    have += 32;
    while (have >= c && w < W) {
      int d = (int)(buf & mask) - half;
      dig[(size_t)w * n + i] = (int16_t)d;
      buf >>= c;
      have -= c;
      w++;
    }
  }
}

static __global__ void __launch_bounds__(256) k_zero_u32(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b,
                                                          size_t nb_, uint32_t* __restrict__ one_word, uint32_t* __restrict__ c,
                                                          size_t nc /* third array (table mode's group counts) or 0 */,
                                                          uint32_t* __restrict__ gse /* two-ended table mode (msm_table.cuh) or null */,
                                                          uint32_t entries) {
  ZG_WAVE_PRIO(3);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) a[i] = 0;
  if (i < nb_) b[i] = 0;
  if (i < nc) c[i] = 0;
  if (i == 0) one_word[0] = 0;
  if (i == 0 && gse) {  // starts {0, E}, ends {0, E}: group 0 grows up from 0, group 1 down from E
    gse[0] = 0;
    gse[1] = entries;
    gse[2] = 0;
    gse[3] = entries;
  }
}

// ---------------------------------------------------------------------------------------------
// counting sort by |digit| per window
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(1024) k_hist(const int16_t* __restrict__ dig, size_t n, uint32_t nb, uint32_t nchunk,
                                                uint32_t* __restrict__ cnt) {
  extern __shared__ uint32_t lh[];
  const uint32_t w = blockIdx.y, chunk = blockIdx.x;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  size_t per = (n + nchunk - 1) / nchunk;
  size_t lo = (size_t)chunk * per, hi = lo + per;
  if (hi > n) hi = n;
  // the bulk eight digits (16 bytes) per lane and load, two loads in flight; the unaligned ends one digit per lane (with
  // one 2-byte load per lane the kernel waited on load latency: k_hist_table, msm_table.cuh)
  auto count1 = [&](int v) {
    if (v != 0) atomicAdd(&lh[(v < 0 ? -v : v) - 1], 1u);
  };
  auto count8 = [&](const uint4& k) {
    count1((int16_t)(k.x & 0xFFFFu)), count1((int16_t)(k.x >> 16)), count1((int16_t)(k.y & 0xFFFFu)), count1((int16_t)(k.y >> 16));
    count1((int16_t)(k.z & 0xFFFFu)), count1((int16_t)(k.z >> 16)), count1((int16_t)(k.w & 0xFFFFu)), count1((int16_t)(k.w >> 16));
  };
  const size_t glo = (size_t)w * n + lo, ghi = (size_t)w * n + hi;  // positions in dig[]
  const size_t a0 = (glo + 7) & ~(size_t)7, a1 = ghi & ~(size_t)7;
  if (a0 < a1) {
    for (size_t g = glo + threadIdx.x; g < a0; g += blockDim.x) count1(dig[g]);
    const uint4* v4 = reinterpret_cast<const uint4*>(dig + a0);
    const size_t nv = (a1 - a0) >> 3;
    size_t i = threadIdx.x;
    for (; i + blockDim.x < nv; i += 2 * (size_t)blockDim.x) {
      const uint4 k0 = v4[i], k1 = v4[i + blockDim.x];
      count8(k0);
      count8(k1);
    }
    if (i < nv) count8(v4[i]);
    for (size_t g = a1 + threadIdx.x; g < ghi; g += blockDim.x) count1(dig[g]);
  } else {
    for (size_t g = glo + threadIdx.x; g < ghi; g += blockDim.x) count1(dig[g]);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
    uint32_t v = lh[b];
    if (v) atomicAdd(&cnt[(size_t)w * nb + b], v);
  }
}

// inclusive prefix sum across the 64 lanes of a wave
// (DPP moves: shifts inside the rows of 16 lanes, then the last lane of a row broadcast to the next row(s); through
// __shfl_up each of the six steps was a trip through the LDS crossbar, and k_scan runs ~100 of these scans in a row)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
  // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes with no source (or outside row_mask) get `old` = 0
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return x;
}

// One block (16 waves) per window: exclusive scans of the bucket counts (entry offsets), of the unit counts and of the
// full-length unit counts, plus the histogram of unit lengths that orders the units longest first.  Each wave scans a
// contiguous sixteenth of the buckets 64 at a time (coalesced loads and stores, shuffles, no barriers inside), the
// waves' totals are combined through LDS.
static __global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ cnt, size_t n, uint32_t nb, uint32_t L,
                                                uint32_t cap, uint32_t* __restrict__ cursor /* absolute positions */,
                                                uint32_t* __restrict__ boff /* bucket start, absolute */,
                                                uint32_t* __restrict__ ufirst /* first unit, absolute */,
                                                uint32_t* __restrict__ wunits /* [W] unit totals, [W] = max units per bucket */,
                                                uint32_t nhi /* partitions of 256 buckets (0: single-level sort) */,
                                                uint32_t* __restrict__ cursor1 /* [W][nhi] */,
                                                uint32_t* __restrict__ pstart /* [W][nhi + 1] */,
                                                uint32_t* __restrict__ lencur /* [W][UNIT_CLASS_STRIDE]: first slot of each length class */,
                                                uint32_t* __restrict__ ffirst /* [W][nb]: full-length units before the bucket */,
                                                const uint32_t* __restrict__ gstart /* table mode: first entry of each bucket group; else null */) {
  ZG_WAVE_PRIO(3);
  // the histogram of unit length classes in SCAN_HIST_COPIES copies, lane l counting into copy l % COPIES: bucket loads
  // cluster around their mean, so a wave's 64 increments fall on a handful of classes, and increments of one address
  // are served one after the other (k_scan without them: 35 instead of 47-64 us, profiles/r3_k_scan.jsonl)
  constexpr uint32_t COPIES = 8;
  __shared__ uint32_t lh[COPIES][UNIT_CLASSES + 1], wtot[16][3], csum[16];
  const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  uint32_t* const my_lh = lh[lane % COPIES];
  const uint32_t per_wave = (nb + 15) / 16;
  const uint32_t lo = wv * per_wave < nb ? wv * per_wave : nb;
  const uint32_t hi = lo + per_wave < nb ? lo + per_wave : nb;
  const uint32_t* __restrict__ c = cnt + (size_t)w * nb;
  for (uint32_t q = tid; q < COPIES * (UNIT_CLASSES + 1); q += blockDim.x) (&lh[0][0])[q] = 0;
  __syncthreads();
  // the wave's counts, read once with all loads in flight together (a load per loop iteration in each of the two passes
  // below was 64 dependent round trips to L2: most of this kernel's 65 us)
  constexpr uint32_t MAXV = 32;  // nb <= 2^15 buckets over 16 waves of 64 lanes
  uint32_t vals[MAXV];
#pragma unroll
  for (uint32_t i = 0; i < MAXV; i++) {
    const uint32_t b = lo + i * 64 + lane;
    vals[i] = b < hi ? c[b] : 0u;
  }
  // pass 1: per-wave totals, histogram of unit length classes (v / L full units and one partial unit of v % L)
  uint32_t sa = 0, sb = 0, sc = 0, mx = 0;
#pragma unroll
  for (uint32_t i = 0; i < MAXV; i++) {
    if (lo + i * 64 >= hi) break;  // wave-uniform
    const uint32_t v = vals[i];
    const uint32_t nu = (v + L - 1) / L, fu = v / L;
    sa += v;
    sb += nu;
    sc += fu;
    mx = nu > mx ? nu : mx;
    if (fu) atomicAdd(&my_lh[UNIT_CLASSES], fu);
    if (v % L) atomicAdd(&my_lh[unit_class(v % L, L)], 1u);
  }
#pragma unroll
  for (uint32_t d = 32; d >= 1; d >>= 1) {
    sa += __shfl_xor(sa, d);
    sb += __shfl_xor(sb, d);
    sc += __shfl_xor(sc, d);
    const uint32_t o = __shfl_xor(mx, d);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) {
    wtot[wv][0] = sa;
    wtot[wv][1] = sb;
    wtot[wv][2] = sc;
    if (mx > 1) atomicMax(&wunits[gridDim.x], mx);
  }
  __syncthreads();
  // slot ranges of the length classes, longest first (full-length units, then the partial classes from longest to
  // shortest): lencur[cls] = the units of every class above cls.  One class per thread, taken from the top (thread t has
  // class UNIT_CLASSES - 1 - t), an inclusive scan over the threads (every thread summing the classes above its own was
  // 8-12 us of this kernel)
  {
    static_assert(UNIT_CLASSES == 1024, "one class per thread of the block");
    const uint32_t cls = UNIT_CLASSES - 1u - tid;
    uint32_t mine = 0, full = 0;
#pragma unroll
    for (uint32_t k = 0; k < COPIES; k++) {
      mine += lh[k][cls];
      full += lh[k][UNIT_CLASSES];
    }
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) csum[wv] = incl;
    __syncthreads();
    uint32_t above = full;  // the full-length units come first
    for (uint32_t q = 0; q < wv; q++) above += csum[q];
    lencur[(size_t)w * UNIT_CLASS_STRIDE + cls] = above + incl - mine;
    if (tid == 0) lencur[(size_t)w * UNIT_CLASS_STRIDE + UNIT_CLASSES] = 0;
  }
  // pass 2: exclusive prefixes
  uint32_t ea = 0, eb = 0, ef = 0, ta = 0, tb = 0;
  for (uint32_t q = 0; q < 16; q++) {
    if (q < wv) {
      ea += wtot[q][0];
      eb += wtot[q][1];
      ef += wtot[q][2];
    }
    ta += wtot[q][0];
    tb += wtot[q][1];
  }
  const uint32_t abase = gstart ? gstart[w] : (uint32_t)((size_t)w * n);
  const uint32_t ubase = w * cap;
#pragma unroll
  for (uint32_t i = 0; i < MAXV; i++) {
    const uint32_t b0 = lo + i * 64;
    if (b0 >= hi) break;  // wave-uniform
    const uint32_t b = b0 + lane;
    const uint32_t v = vals[i];
    const uint32_t nu = (v + L - 1) / L, fu = v / L;
    const uint32_t ia = wave_inclusive_scan(v), ib = wave_inclusive_scan(nu), ic = wave_inclusive_scan(fu);
    if (b < hi) {
      const size_t g = (size_t)w * nb + b;
      const uint32_t xa = abase + ea + ia - v;
      cursor[g] = xa;
      boff[g] = xa;
      ufirst[g] = ubase + eb + ib - nu;
      ffirst[g] = ef + ic - fu;
      if (nhi && (b & 255u) == 0) {
        cursor1[(size_t)w * nhi + (b >> 8)] = xa;
        pstart[(size_t)w * (nhi + 1) + (b >> 8)] = xa;
      }
    }
    ea += __shfl(ia, 63);
    eb += __shfl(ib, 63);
    ef += __shfl(ic, 63);
  }
  if (tid == 0) {
    wunits[w] = tb;
    if (nhi) pstart[(size_t)w * (nhi + 1) + nhi] = abase + ta;
  }
}

static __global__ void __launch_bounds__(1024) k_scatter(const int16_t* __restrict__ dig, size_t n, uint32_t nb,
                                                   uint32_t nchunk, uint32_t* __restrict__ cursor,
                                                   uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lh[];
  const uint32_t w = blockIdx.y, chunk = blockIdx.x;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  size_t per = (n + nchunk - 1) / nchunk;
  size_t lo = (size_t)chunk * per, hi = lo + per;
  if (hi > n) hi = n;
  const int16_t* d = dig + (size_t)w * n;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int v = d[i];
    if (v != 0) atomicAdd(&lh[(v < 0 ? -v : v) - 1], 1u);
  }
  __syncthreads();
  // claim this block's slice of every bucket it touches; lh[b] becomes the block-private write cursor
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
    uint32_t v = lh[b];
    if (v) lh[b] = atomicAdd(&cursor[(size_t)w * nb + b], v);
  }
  __syncthreads();
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int v = d[i];
    if (v != 0) {
      uint32_t pos = atomicAdd(&lh[(v < 0 ? -v : v) - 1], 1u);
      sorted[pos] = (uint32_t)i | (v < 0 ? 0x80000000u : 0u);
    }
  }
}

// Two-level (most-significant-digit first) counting sort used when a window has >= 512 buckets.  Scattering
// straight into 2^15 bucket streams writes 4-byte entries to 2^15 different cache lines per block (measured 6.7 ms
// at 2^24 points); here level 1 splits a tile over <= 128 partitions of 256 buckets (runs of ~128 entries) and level 2
// finishes each partition on its low 8 bits (runs of ~32 entries), so both levels write in long contiguous runs.
#ifndef ZG_SORT_WG
#define ZG_SORT_WG 1024
#endif
#ifndef ZG_SORT_PER
#define ZG_SORT_PER 8
#endif
static constexpr uint32_t SORT_WG = ZG_SORT_WG;      // lanes of k_scatter_hi_table and k_scatter_lo
static constexpr uint32_t SORT_PER = ZG_SORT_PER;    // consecutive entries per lane (a multiple of eight)
static_assert(SORT_PER % 8 == 0 && (SORT_PER & (SORT_PER - 1)) == 0, "entries are read eight at a time; tiles start on multiples of SORT_PER");
static constexpr uint32_t SORT_TILE1 = 8192;                // k_scatter_hi (1024 lanes)
static constexpr uint32_t SORT_TILE1T = SORT_PER * SORT_WG; // k_scatter_hi_table
static constexpr uint32_t SORT_TILE2 = SORT_PER * SORT_WG;  // k_scatter_lo

// Both levels sort their tile inside LDS first (histogram, exclusive scan, ranked placement) and then copy the LDS
// image out linearly, so that a wave writes 64 consecutive entries of (mostly) one run: full-line global stores.
static __global__ void __launch_bounds__(1024) k_scatter_hi(const int16_t* __restrict__ dig, size_t n, uint32_t nhi,
                                                      uint32_t* __restrict__ cursor1, uint32_t* __restrict__ mid_ent,
                                                      uint8_t* __restrict__ mid_lo, uint32_t w0 /* first window of the group */) {
  __shared__ uint32_t h[128], lstart[128], gbase[128];
  __shared__ uint32_t buf_ent[SORT_TILE1];
  __shared__ uint8_t buf_lo[SORT_TILE1], buf_bin[SORT_TILE1];
  const uint32_t w = w0 + blockIdx.y;
  const size_t lo = (size_t)blockIdx.x * SORT_TILE1;
  size_t hi = lo + SORT_TILE1;
  if (hi > n) hi = n;
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  __syncthreads();
  const int16_t* d = dig + (size_t)w * n;
  // one pass: every lane keeps its (up to) 8 digits and their ranks within their bins in registers, so each entry
  // costs one LDS atomic and the digits are read once
  constexpr uint32_t PER = SORT_TILE1 / 1024;
  static_assert(PER == 8, "eight digits per lane: one 16-byte load where the window's digits are 16-byte aligned");
  int my_v[PER];
  uint32_t my_rank[PER];
  // lane t takes digits [lo + 8 t, lo + 8 t + 8): one 16-byte load when the window's row of dig[] is aligned (n a multiple
  // of eight), else eight 2-byte loads; which lane ranks which digit makes no difference to the sort
  const size_t i0 = lo + (size_t)threadIdx.x * PER;
  if ((((size_t)w * n) & 7u) == 0 && i0 + PER <= hi) {
    const uint4 k = *reinterpret_cast<const uint4*>(d + i0);
    my_v[0] = (int16_t)(k.x & 0xFFFFu), my_v[1] = (int16_t)(k.x >> 16), my_v[2] = (int16_t)(k.y & 0xFFFFu), my_v[3] = (int16_t)(k.y >> 16);
    my_v[4] = (int16_t)(k.z & 0xFFFFu), my_v[5] = (int16_t)(k.z >> 16), my_v[6] = (int16_t)(k.w & 0xFFFFu), my_v[7] = (int16_t)(k.w >> 16);
  } else {
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) my_v[r] = i0 + r < hi ? (int)d[i0 + r] : 0;
  }
#pragma unroll
  for (uint32_t r = 0; r < PER; r++)
    if (my_v[r] != 0) my_rank[r] = atomicAdd(&h[(uint32_t)((my_v[r] < 0 ? -my_v[r] : my_v[r]) - 1) >> 8], 1u);
  __syncthreads();
  {
    // exclusive scan over <= 128 bins: wave prefix sums by the first two waves, then the first wave's total
    __shared__ uint32_t wsum;
    uint32_t v = 0, incl = 0;
    if (threadIdx.x < 128) {
      v = threadIdx.x < nhi ? h[threadIdx.x] : 0;
      incl = wave_inclusive_scan(v);
      if (threadIdx.x == 63) wsum = incl;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const uint32_t excl = incl - v + (threadIdx.x >= 64 ? wsum : 0);
      lstart[threadIdx.x] = excl;
      gbase[threadIdx.x] = v ? atomicAdd(&cursor1[(size_t)w * nhi + threadIdx.x], v) : 0;
    }
    __syncthreads();
  }
#pragma unroll
  for (uint32_t r = 0; r < PER; r++) {
    const int v = my_v[r];
    if (v != 0) {
      const uint32_t b = (uint32_t)((v < 0 ? -v : v) - 1);
      const uint32_t pos = lstart[b >> 8] + my_rank[r];
      buf_ent[pos] = (uint32_t)(i0 + r) | (v < 0 ? 0x80000000u : 0u);
      buf_lo[pos] = (uint8_t)(b & 255u);
      buf_bin[pos] = (uint8_t)(b >> 8);
    }
  }
  __syncthreads();
  const uint32_t total = lstart[nhi - 1] + h[nhi - 1];
  for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
    uint32_t bin = buf_bin[i];
    uint32_t g = gbase[bin] + (i - lstart[bin]);
    mid_ent[g] = buf_ent[i];
    mid_lo[g] = buf_lo[i];
  }
}

// grid (tiles, nhi, W): partition p of window w, tiles taken grid-stride so that any distribution is covered
static __global__ void __launch_bounds__(SORT_WG) k_scatter_lo(const uint32_t* __restrict__ mid_ent,
                                                     const uint8_t* __restrict__ mid_lo,
                                                     const uint32_t* __restrict__ pstart, uint32_t nhi, uint32_t nb,
                                                     uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted,
                                                     uint32_t w0 /* first window of the group */) {
  __shared__ uint32_t h[256], lstart[256], gbase[256], wsum[4];
  __shared__ uint32_t buf_ent[SORT_TILE2];
  __shared__ uint8_t buf_bin[SORT_TILE2];
  const uint32_t p = blockIdx.y, w = w0 + blockIdx.z;
  const uint32_t start = pstart[(size_t)w * (nhi + 1) + p], end = pstart[(size_t)w * (nhi + 1) + p + 1];
  uint32_t* cur = cursor + (size_t)w * nb + (size_t)p * 256;
  // tiles on multiples of SORT_PER entries, consecutive entries per lane read eight at a time (see k_scatter_hi_table)
  for (uint32_t lo = (start & ~(SORT_PER - 1u)) + blockIdx.x * SORT_TILE2; lo < end; lo += gridDim.x * SORT_TILE2) {
    uint32_t hi = lo + SORT_TILE2;
    if (hi > end) hi = end;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    // one pass over the tile: every lane keeps its entries and their ranks within their bins in registers, so each
    // entry costs one LDS atomic and one read of the intermediate arrays
    constexpr uint32_t PER = SORT_PER;
    uint32_t my_ent[PER], my_rank[PER];
    uint32_t my_bin[PER];
    const uint32_t q0 = lo + threadIdx.x * PER;
#pragma unroll
    for (uint32_t c = 0; c < PER; c += 8) {
      if (q0 + c < hi) {  // reads up to seven entries past `end`: inside the workspace
        const uint2 b = *reinterpret_cast<const uint2*>(mid_lo + q0 + c);
        const uint4 e0 = *reinterpret_cast<const uint4*>(mid_ent + q0 + c);
        const uint4 e1 = *reinterpret_cast<const uint4*>(mid_ent + q0 + c + 4);
        my_bin[c + 0] = b.x & 255u, my_bin[c + 1] = (b.x >> 8) & 255u, my_bin[c + 2] = (b.x >> 16) & 255u, my_bin[c + 3] = b.x >> 24;
        my_bin[c + 4] = b.y & 255u, my_bin[c + 5] = (b.y >> 8) & 255u, my_bin[c + 6] = (b.y >> 16) & 255u, my_bin[c + 7] = b.y >> 24;
        my_ent[c + 0] = e0.x, my_ent[c + 1] = e0.y, my_ent[c + 2] = e0.z, my_ent[c + 3] = e0.w;
        my_ent[c + 4] = e1.x, my_ent[c + 5] = e1.y, my_ent[c + 6] = e1.z, my_ent[c + 7] = e1.w;
      }
    }
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
      const uint32_t i = q0 + r;
      if (i >= start && i < hi)
        my_rank[r] = atomicAdd(&h[my_bin[r]], 1u);
      else
        my_bin[r] = 0xFFFFFFFFu;
    }
    __syncthreads();
    {
      // exclusive scan of the 256 counts: wave prefix sums by the first four waves, then the waves' totals
      uint32_t v = 0, incl = 0;
      if (threadIdx.x < 256) {
        v = h[threadIdx.x];
        incl = wave_inclusive_scan(v);
        if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
      }
      __syncthreads();
      if (threadIdx.x < 256) {
        uint32_t off = 0;
        for (uint32_t q = 0; q < (threadIdx.x >> 6); q++) off += wsum[q];
        const uint32_t excl = off + incl - v;
        lstart[threadIdx.x] = excl;
        gbase[threadIdx.x] = v ? atomicAdd(&cur[threadIdx.x], v) : 0;
      }
      __syncthreads();
    }
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
      if (my_bin[r] != 0xFFFFFFFFu) {
        const uint32_t pos = lstart[my_bin[r]] + my_rank[r];
        buf_ent[pos] = my_ent[r];
        buf_bin[pos] = (uint8_t)my_bin[r];
      }
    }
    __syncthreads();
    const uint32_t total = hi - (lo > start ? lo : start);
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
      uint32_t bin = buf_bin[i];
      sorted[gbase[bin] + (i - lstart[bin])] = buf_ent[i];
    }
    __syncthreads();
  }
}

// Unit table and the longest-first order.  k_units: one lane per bucket writes the bucket's first COMBINE_SERIAL_MAX
// units and lists the bucket as heavy when it has more; k_units_heavy: the remaining units of the heavy buckets,
// one lane each (a bucket holding half of all entries -- the "1" scalars of a witness -- has thousands).
// Slots of the order: full-length units are ranked by a scan (no atomics); the one partial unit of a bucket takes a
// slot of its length class, claimed per workgroup through LDS (one global atomic per class and workgroup: the
// same-address atomics of 2^15 buckets per window on ~100 counters were 0.1 ms).
__device__ __forceinline__ void emit_full_unit(uint32_t k, uint32_t g, uint32_t w, uint32_t u0, uint32_t start,
                                               uint32_t full_rank, uint32_t L, uint32_t cap, UnitDesc* __restrict__ units,
                                               const uint32_t* __restrict__ lencur, uint32_t* __restrict__ order) {
  units[u0 + k] = UnitDesc{(uint64_t)start + (uint64_t)k * L, L, g};
  order[(size_t)w * cap + lencur[(size_t)w * UNIT_CLASS_STRIDE + UNIT_CLASSES] + full_rank + k] = u0 + k;
}

static __global__ void __launch_bounds__(256) k_units(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ boff,
                                                const uint32_t* __restrict__ ufirst, const uint32_t* __restrict__ ffirst,
                                                uint32_t nb, uint32_t W, uint32_t L, uint32_t cap,
                                                UnitDesc* __restrict__ units, uint32_t* __restrict__ lencur,
                                                uint32_t* __restrict__ order /* [W][cap]: unit ids, longest first */,
                                                uint32_t* __restrict__ heavy /* [0] = count, then ids of heavy buckets */,
                                                uint32_t smax) {
  ZG_WAVE_PRIO(3);
  __shared__ uint32_t lc[UNIT_CLASSES], gb[UNIT_CLASSES];
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = g < (size_t)nb * W;
  const bool one_window = (nb % 256u) == 0;  // then the 256 buckets of this workgroup share a window
  const uint32_t v = in_range ? cnt[g] : 0;
  const uint32_t w = in_range ? (uint32_t)(g / nb) : 0;
  const uint32_t full = v / L, rem = v % L;
  const uint32_t u0 = v ? ufirst[g] : 0, start = v ? boff[g] : 0, fr = v ? ffirst[g] : 0;
  for (uint32_t q = threadIdx.x; q < UNIT_CLASSES; q += blockDim.x) lc[q] = 0;
  __syncthreads();
  if (v) {
    if (full + (rem ? 1 : 0) > smax) heavy[1 + atomicAdd(&heavy[0], 1u)] = (uint32_t)g;
    const uint32_t mine = full < smax ? full : smax;
    for (uint32_t k = 0; k < mine; k++) emit_full_unit(k, (uint32_t)g, w, u0, start, fr, L, cap, units, lencur, order);
  }
  // the partial unit (index `full`), unless it belongs to k_units_heavy
  const bool partial = v && rem && full < smax;
  const uint32_t cls = partial ? unit_class(rem, L) : 0;
  uint32_t rank = 0;
  if (partial && one_window) rank = atomicAdd(&lc[cls], 1u);
  __syncthreads();
  if (one_window) {
    const uint32_t wb = (uint32_t)(((size_t)blockIdx.x * blockDim.x) / nb);
    for (uint32_t q = threadIdx.x; q < UNIT_CLASSES; q += blockDim.x) {
      const uint32_t c = lc[q];
      if (c) gb[q] = atomicAdd(&lencur[(size_t)wb * UNIT_CLASS_STRIDE + q], c);
    }
  }
  __syncthreads();
  if (partial) {
    const uint32_t slot = one_window ? gb[cls] + rank : atomicAdd(&lencur[(size_t)w * UNIT_CLASS_STRIDE + cls], 1u);
    units[u0 + full] = UnitDesc{(uint64_t)start + (uint64_t)full * L, rem, (uint32_t)g};
    order[(size_t)w * cap + slot] = u0 + full;
  }
}

static __global__ void __launch_bounds__(256) k_units_heavy(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ boff,
                                                      const uint32_t* __restrict__ ufirst, const uint32_t* __restrict__ ffirst,
                                                      uint32_t nb, uint32_t L, uint32_t cap, UnitDesc* __restrict__ units,
                                                      uint32_t* __restrict__ lencur, uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ heavy, uint32_t smax) {
  ZG_WAVE_PRIO(3);
  const uint32_t nheavy = heavy[0];
  for (uint32_t hi = blockIdx.y; hi < nheavy; hi += gridDim.y) {
    const uint32_t g = heavy[1 + hi];
    const uint32_t v = cnt[g], w = g / nb;
    const uint32_t full = v / L, rem = v % L;
    const uint32_t nu = full + (rem ? 1 : 0);
    const uint32_t u0 = ufirst[g], start = boff[g], fr = ffirst[g];
    for (uint32_t k = smax + blockIdx.x * blockDim.x + threadIdx.x; k < nu; k += gridDim.x * blockDim.x) {
      if (k < full) {
        emit_full_unit(k, g, w, u0, start, fr, L, cap, units, lencur, order);
      } else {
        units[u0 + k] = UnitDesc{(uint64_t)start + (uint64_t)k * L, rem, g};
        order[(size_t)w * cap + atomicAdd(&lencur[(size_t)w * UNIT_CLASS_STRIDE + unit_class(rem, L)], 1u)] = u0 + k;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bucket accumulation (dominant kernel)
// ---------------------------------------------------------------------------------------------
template <class F>
ZG_HD Affine<F> load_affine(const Affine<F>* __restrict__ bases, size_t idx) {
  constexpr int NW = sizeof(Affine<F>) / 16;
  union {
    uint4 q[NW];
    Affine<F> p;
  } u;
  const uint4* src = reinterpret_cast<const uint4*>(bases + idx);
#pragma unroll
  for (int i = 0; i < NW; i++) u.q[i] = src[i];
  return u.p;
}

// acc += p for the lanes of one wave.  The general addition (xyzz_madd) carries four exceptional cases; as divergent
// branches they cost ~2,500 of its ~15,100 cycles per wave (tools/bench_madd.hip: the straight-line formula alone is
// ~12,600) in exec-mask bookkeeping and register copies at the joins.  Here the common case is one straight line and
// the exceptional cases are taken by the WHOLE wave (ballot), which is rare: an identity base, an accumulator that
// cancelled to the identity, or an addition of equal / opposite points.
template <class F>
__device__ __forceinline__ void xyzz_madd_wave(XYZZ<F>& acc, const Affine<F>& p) {
  const F U2 = F::mul(p.x, acc.zz);
  const F S2 = F::mul(p.y, acc.zzz);
  const F Pd = F::sub(U2, acc.x);
  const F R = F::sub(S2, acc.y);
  const int exceptional = (int)p.is_inf() | (int)acc.is_inf() | (int)Pd.is_zero();
  if (__ballot(exceptional) != 0ull) {
    xyzz_madd(acc, p);
    return;
  }
  const F PP = F::sqr(Pd);
  const F PPP = F::mul(Pd, PP);
  const F Q = F::mul(acc.x, PP);
  const F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
  acc.y = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(acc.y, PPP));
  acc.x = X3;
  acc.zz = F::mul(acc.zz, PP);
  acc.zzz = F::mul(acc.zzz, PPP);
}

// which coordinate fields accumulate in the reduced-radix form: Fq (G1, accum29.cuh) and Fq2 (G2, accum29x2.cuh)
#ifndef ZG_ACCUM_RADIX29
#define ZG_ACCUM_RADIX29 1
#endif
#ifndef ZG_ACCUM_RADIX29_G2
#define ZG_ACCUM_RADIX29_G2 1
#endif
template <class F>
struct Radix29For {
  static constexpr bool value = false;
};
template <class P>
struct Radix29For<Fp<P>> {
  static constexpr bool value = ZG_ACCUM_RADIX29 != 0;
};
template <class P, int NR>
struct Radix29For<Fp2<Fp<P>, NR>> {
  static constexpr bool value = ZG_ACCUM_RADIX29_G2 != 0;
};

// the arithmetic of the quad kernels (k_combine_*, k_reduce*) for coordinate field F, and with it the form of the
// buffers they exchange with k_accum
template <class F>
struct QuadSel {
  using type = QuadOps32<F>;
};
#if ZG_QUAD_RADIX29
template <class P>
struct QuadSel<Fp<P>> {
  using type = std::conditional_t<Radix29For<Fp<P>>::value, QuadOps29<P>, QuadOps32<Fp<P>>>;
};
#endif
template <class F>
using QuadOps = typename QuadSel<F>::type;

#ifndef ZG_ACCUM_TOUCH
#define ZG_ACCUM_TOUCH 0
#endif
#ifndef ZG_ACCUM_PIN_LOADS
#define ZG_ACCUM_PIN_LOADS 0
#endif
#ifndef ZG_ACCUM_MIN_WAVES
#define ZG_ACCUM_MIN_WAVES 1
#endif
#ifndef ZG_ACCUM_BLOCK
#define ZG_ACCUM_BLOCK 128
#endif
template <class F>
__global__ void __launch_bounds__(ZG_ACCUM_BLOCK, ZG_ACCUM_MIN_WAVES) k_accum(const UnitDesc* __restrict__ units, const uint32_t* __restrict__ wunits,
                                                uint32_t cap, const uint32_t* __restrict__ order,
                                                const uint32_t* __restrict__ sorted,
                                                const Affine<F>* __restrict__ bases, XYZZ<F>* __restrict__ part,
                                                const uint32_t idx_mask /* 0x7FFFFFFF: strips the sign bit */,
                                                const uint32_t w_end /* windows [w_end - gridDim.y, w_end) */,
                                                const uint32_t groups_fastest /* table mode: grid (groups, slot blocks) */) {
  // Workgroups are dispatched in blockIdx order.  The top window comes first: its digits cover fewer values (the scalar's
  // leading bits), so its buckets -- and units -- are the longest, and a long unit that starts late is the kernel's tail.
  // Table mode (groups_fastest): the groups are statistically alike, each with its share of the long units of the
  // scalars' top digit, so the grid is (group, slot block) and every group's longest units are dispatched first.
  const uint32_t w = groups_fastest ? blockIdx.x : w_end - 1 - blockIdx.y;
  const uint32_t ul = (groups_fastest ? blockIdx.y : blockIdx.x) * blockDim.x + threadIdx.x;
  if (ul >= wunits[w]) return;
  const size_t u = order[(size_t)w * cap + ul];  // longest units first; the lanes of a wave get (nearly) equal lengths
  const UnitDesc d = units[u];
  const uint32_t* __restrict__ ent = sorted + d.start;
  const uint32_t last = d.len - 1;
  // software pipeline: the base of entry t + 1 and the index of entry t + 2 are in flight while entry t is added
  uint32_t e = ent[0];
  uint32_t en = ent[last < 1 ? last : 1];
  Affine<F> p = load_affine(bases, e & idx_mask);
  if constexpr (Radix29For<F>::value) {
    // the accumulator lives in the reduced-radix form (accum29.cuh, accum29x2.cuh) and returns to the stored form once per
    // unit
    using A = AccumOf<F>;
    typename A::Acc acc = A::from_affine(p, (e >> 31) != 0);
    p = load_affine(bases, en & idx_mask);
    e = en;
    en = ent[last < 2 ? last : 2];
#if ZG_ACCUM_TOUCH
    uint32_t touch_prev = 0;
#endif
    for (uint32_t t = 1; t <= last; t++) {
      const Affine<F> pn = load_affine(bases, en & idx_mask);
      const uint32_t en2 = ent[t + 2 <= last ? t + 2 : last];
#if ZG_ACCUM_TOUCH
      // one word of the base after next: brings its line (and its page translation) in a whole addition early
      const uint32_t touch = reinterpret_cast<const uint32_t*>(bases + (en2 & idx_mask))[0];
#endif
#if ZG_ACCUM_PIN_LOADS
      __builtin_amdgcn_sched_barrier(0);
#endif
      acc = A::madd(acc, p, (e >> 31) != 0);
      e = en;
      en = en2;
      p = pn;
#if ZG_ACCUM_TOUCH
      asm volatile("" ::"v"(touch_prev));
      touch_prev = touch;
#endif
    }
    if constexpr (QuadOps<F>::rprime_buffers)
      part[u] = A::to_part(acc);
    else
      part[u] = A::to_xyzz(acc);
  } else {
    p.y = F::cneg(p.y, (e >> 31) != 0);
    XYZZ<F> acc = XYZZ<F>::from_affine(p);
    p = load_affine(bases, en & idx_mask);
    e = en;
    en = ent[last < 2 ? last : 2];
    for (uint32_t t = 1; t <= last; t++) {
      const Affine<F> pn = load_affine(bases, en & idx_mask);
      const uint32_t en2 = ent[t + 2 <= last ? t + 2 : last];
      p.y = F::cneg(p.y, (e >> 31) != 0);
      xyzz_madd_wave(acc, p);
      e = en;
      en = en2;
      p = pn;
    }
    part[u] = acc;
  }
}

// Buckets that were split into several units (skewed scalars: 0/1-heavy witnesses, short top windows) are folded into
// their first unit.  Up to COMBINE_SERIAL_MAX units: the first unit's QUAD adds the others one after another
// (k_combine_light).  More ("heavy" buckets, listed by k_units -- a JoinSplit witness has one, holding half of all
// entries in thousands of units): a tree, 512 units per workgroup and launch (k_combine_tree with stride 1, then stride
// 512 over the chunk heads).
//
// These kernels and the bucket reduction below are chains of dependent full additions with little parallel work: every
// point is handled by the four lanes of a quad (curve_quad.cuh), one coordinate and one field product per lane and
// round, which shortens each addition from 14 dependent products to 4.

// threads of a workgroup of the quad kernels: wider fields hold more registers per lane
template <class F>
constexpr int quad_block() {
  return sizeof(F) <= 32 ? 1024 : (sizeof(F) <= 64 ? 512 : 256);
}

// one quad per BUCKET (not per unit slot: a window has nb buckets but up to nb + entries / L slots, nearly all of which
// had nothing to do -- 75 M lanes launched at 2^24 points, 0.2 ms)
template <class F>
__global__ void __launch_bounds__(256) k_combine_light(const uint32_t* __restrict__ wunits, uint32_t W, uint32_t nb,
                                                        const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ ufirst,
                                                        uint32_t L, XYZZ<F>* __restrict__ part, uint32_t smax) {
  ZG_WAVE_PRIO(3);
  if (wunits[W] <= 1) return;  // no bucket has more than one unit
  const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, q = threadIdx.x & 3u;
  if (b >= nb) return;
  const size_t g = (size_t)blockIdx.y * nb + b;
  const uint32_t nu = (cnt[g] + L - 1) / L;
  if (nu < 2 || nu > smax) return;
  const uint32_t u = ufirst[g];
  using Q = QuadOps<F>;
  typename Q::Coord a = Q::load(part + u, q);
  for (uint32_t k = 1; k < nu; k++) a = Q::add(a, Q::load(part + u + k, q), q);
  Q::store(part + u, q, a);
}

// grid (chunk slots, heavy slots); elements of this level: part[first + stride * j], j < ceil(units / stride)
template <class F>
__global__ void __launch_bounds__(quad_block<F>()) k_combine_tree(const uint32_t* __restrict__ heavy /* [0] = count, then bucket ids */,
                                                                   const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ ufirst,
                                                                   uint32_t L, uint32_t stride, XYZZ<F>* __restrict__ part,
                                                                   const uint32_t* __restrict__ max_units /* most units in any bucket */) {
  ZG_WAVE_PRIO(3);
  // chunk slots beyond the longest bucket have nothing to do in any heavy bucket
  if ((size_t)blockIdx.x * COMBINE_CHUNK * stride >= max_units[0]) return;
  using Q = QuadOps<F>;
  const uint32_t nheavy = heavy[0];
  const uint32_t qi = threadIdx.x >> 2, nq = blockDim.x >> 2, q = threadIdx.x & 3u;
  for (uint32_t hi = blockIdx.y; hi < nheavy; hi += gridDim.y) {
    const uint32_t g = heavy[1 + hi];
    const uint32_t nu = (cnt[g] + L - 1) / L;
    const uint32_t nelem = (nu + stride - 1) / stride;
    if (nelem < 2) continue;
    XYZZ<F>* base = part + ufirst[g];
    for (uint32_t c0 = blockIdx.x * COMBINE_CHUNK; c0 < nelem; c0 += gridDim.x * COMBINE_CHUNK) {
      const uint32_t c1 = c0 + COMBINE_CHUNK < nelem ? c0 + COMBINE_CHUNK : nelem;
      for (uint32_t s = 1; s < COMBINE_CHUNK && c0 + s < c1; s <<= 1) {
        for (uint32_t j0 = c0 + 2 * s * qi; j0 + s < c1; j0 += 2 * s * nq) {
          const typename Q::Coord a = Q::load(base + (size_t)stride * j0, q);
          const typename Q::Coord c = Q::load(base + (size_t)stride * (j0 + s), q);
          Q::store(base + (size_t)stride * j0, q, Q::add(a, c, q));
        }
        __syncthreads();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bucket reduction
// ---------------------------------------------------------------------------------------------
// quad (w, j): buckets [j*k, (j+1)*k) of window w; seg[w*T + j] = sum_b (b + 1) * B_b over the segment
template <class F>
__global__ void __launch_bounds__(128) k_reduce1(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ ufirst,
                                                  const XYZZ<F>* __restrict__ part, uint32_t nb, uint32_t T, uint32_t k,
                                                  uint32_t L, uint32_t W, XYZZ<F>* __restrict__ seg,
                                                  XYZZ<F>* __restrict__ seg_plain /* table mode: unweighted segment sums; else null */,
                                                  uint32_t weighted /* 0 (table mode): seg gets sum (i + 1) B_{jk+i} only */) {
  ZG_WAVE_PRIO(3);
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, q = threadIdx.x & 3u;
  if (g >= T * W) return;
  const uint32_t w = g / T, j = g % T;
  using Q = QuadOps<F>;
  using Coord = typename Q::Coord;
  Coord run = Q::zero(), tot = Q::zero();  // the identity: every coordinate zero
  // (the three dependent loads at the head of each step -- count, first unit, bucket sum -- are not what bounds this
  // kernel: with the loads running two buckets ahead of the additions it took the same time, 0.28-0.36 ms for 2^19
  // buckets; at two waves per SIMD it is bound by the issue of the additions themselves, ~6,500 cycles each)
  for (int bb = (int)k - 1; bb >= 0; bb--) {
    const size_t gid = (size_t)w * nb + (size_t)j * k + (uint32_t)bb;
    if (cnt[gid]) run = Q::add(run, Q::load(part + ufirst[gid], q), q);  // k_combine left the bucket sum in its first unit
    tot = Q::add(tot, run, q);
  }
  if (weighted) {
    // tot = sum (b - j*k + 1) B_b ; add (j*k) * run, j*k < nb
    const uint32_t start = j * k;
    Coord r = Q::zero();
    for (int b = 31 - __builtin_clz(nb > 1 ? nb - 1 : 1); b >= 0; b--) {
      r = Q::dbl(r, q);
      if ((start >> b) & 1u) r = Q::add(r, run, q);
    }
    tot = Q::add(tot, r, q);
  }
  Q::store(seg + g, q, tot);
  if (seg_plain) Q::store(seg_plain + g, q, run);
}

// Tree-sum of the segment sums of a row (a window; table mode: a bucket group's weighted or unweighted sums).  Block
// (row, chunk) adds the n entries s[j * stride], j < n, of its chunk s = seg + row * T + chunk * n * stride into s[0].  One
// launch with n = T when a block's quads cover the row; else two: chunks of one entry per quad (stride 1), then the chunk
// heads (stride = chunk length, `final`) -- a row of 2048 sums is then 11 dependent additions instead of 15 (Fq), 22
// (Fq2, 128 quads a block) or 37 (BLS12-377 Fq2, 64 quads).  `final`: the result is what the host reads (stored form).
template <class F>
__global__ void __launch_bounds__(quad_block<F>()) k_reduce2(XYZZ<F>* __restrict__ seg, uint32_t T, uint32_t n, uint32_t stride,
                                                            int final) {
  ZG_WAVE_PRIO(3);
  XYZZ<F>* s = seg + (size_t)blockIdx.x * T + (size_t)blockIdx.y * n * stride;
  using Q = QuadOps<F>;
  const uint32_t qi = threadIdx.x >> 2, nq = blockDim.x >> 2, q = threadIdx.x & 3u;
  if (n == 1) {  // nothing to add: the one sum only changes form
    if (qi == 0 && final) Q::store_final(s, q, Q::load(s, q));
    return;
  }
  if (n > nq) {  // fold the tail onto the first nq entries
    typename Q::Coord a = Q::load(s + (size_t)qi * stride, q);
    for (uint32_t j = qi + nq; j < n; j += nq) a = Q::add(a, Q::load(s + (size_t)j * stride, q), q);
    Q::store(s + (size_t)qi * stride, q, a);
    __syncthreads();
    n = nq;
  }
  for (uint32_t h = n >> 1; h >= 1; h >>= 1) {
    if (qi < h) {
      const typename Q::Coord r = Q::add(Q::load(s + (size_t)qi * stride, q), Q::load(s + (size_t)(qi + h) * stride, q), q);
      if (h == 1 && final)
        Q::store_final(s, q, r);
      else
        Q::store(s + (size_t)qi * stride, q, r);
    }
    __syncthreads();
  }
}

// Table mode: the weights of the segment sums without a multiplication per segment.  k_reduce1 leaves, for the flat
// segment index I = group * T + j, the plain sum X[I] and the locally weighted sum Y[I] of its k buckets.  What the host
// needs is  sum_I Y[I],  S = sum_I X[I]  and  sum_I I' X[I]  for I' = j (bucket weights) and I' = group (group weights):
// every such weighted sum is  sum_t 2^t BIT_t  with  BIT_t = sum over the I whose bit t is set of X[I].  All BIT_t come out
// of ONE pairwise tree over X, in place: at level u a node (2^(u+1) entries from i) adds its right half's head into its
// left half's head -- X[i] += X[i + 2^u] -- and does the same for the bit sums below u, which live at X[i + 2^t]:
// X[i + 2^t] += X[i + 2^u + 2^t], t < u; the right head X[i + 2^u] itself IS the node's bit-u sum and stays where it is.
// After log2 N levels X[0] = S and X[2^t] = BIT_t.  2 N additions in all instead of the ~1.5 log2(buckets) doublings and
// additions every segment paid for its weight (40-65 % of the reduction's instructions), in a chain of log2 N additions;
// the host then runs two Horner chains of log2 N steps on single points, which it does ten times faster than a quad.
//   grid.x blocks of 2^hi entries each run the levels [lo, hi); Y (the other sums) is tree-summed alongside.
//   rows (last launch only): Y total, S, BIT_0 .. BIT_(hi-1) in the form the host reads.
template <class F>
__global__ void __launch_bounds__(quad_block<F>()) k_bitsum(XYZZ<F>* __restrict__ X, XYZZ<F>* __restrict__ Y, uint32_t lo, uint32_t hi,
                                                           XYZZ<F>* __restrict__ rows /* null except in the last launch */) {
  ZG_WAVE_PRIO(3);
  using Q = QuadOps<F>;
  const uint32_t qi = threadIdx.x >> 2, nq = blockDim.x >> 2, q = threadIdx.x & 3u;
  XYZZ<F>* x = X + ((size_t)blockIdx.x << hi);
  XYZZ<F>* y = Y + ((size_t)blockIdx.x << hi);
  for (uint32_t u = lo; u < hi; u++) {
    const uint32_t nodes = 1u << (hi - u - 1), per = u + 2;  // per node: S, the u bit sums below, and Y
    for (uint32_t item = qi; item < nodes * per; item += nq) {
      const uint32_t node = item / per, e = item % per;
      const size_t i = (size_t)node << (u + 1);
      XYZZ<F>* base = e == u + 1 ? y : x;
      const size_t o = (e == 0 || e == u + 1) ? 0 : ((size_t)1 << (e - 1));
      const typename Q::Coord r = Q::add(Q::load(base + i + o, q), Q::load(base + i + ((size_t)1 << u) + o, q), q);
      Q::store(base + i + o, q, r);
    }
    __syncthreads();
  }
  if (rows && blockIdx.x == 0) {
    for (uint32_t item = qi; item < hi + 2; item += nq) {
      const XYZZ<F>* src = item == 0 ? y : (item == 1 ? x : x + ((size_t)1 << (item - 2)));
      Q::store_final(rows + item, q, Q::load(src, q));
    }
  }
}

}  // namespace zg
