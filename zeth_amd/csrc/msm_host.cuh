// Host driver for the MSM pipeline plus base-point set-up kernels, generic over the group.
// One translation unit per (curve, group) instantiates this through ZG_DEFINE_GROUP_OPS.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.cuh"
#include "msm.cuh"
#include "msm_table.cuh"

namespace zg {

template <class F>
struct BaseField {
  using type = F;
};
template <class B, int K>
struct BaseField<Fp2<B, K>> {
  using type = B;
};

// ---------------------------------------------------------------------------------------------
// set-up kernels
// ---------------------------------------------------------------------------------------------
template <class B>
__global__ void __launch_bounds__(256) k_to_mont(B* __restrict__ elems, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  elems[i] = B::to_mont(elems[i]);
}

static constexpr uint32_t GEN_CHUNK = 64;

// Measurement knob, compiled in only with -DZG_DEBUG_KNOBS (tools/ builds): ZG_DEBUG_BASE_MASK=0xffff makes every gather
// of the accumulation kernel hit the first 2^16 bases -- wrong results by design -- which separates the kernel's
// arithmetic time from its memory time.  The shipped library ignores the variable.
inline uint32_t zg_debug_base_mask() {
#ifdef ZG_DEBUG_KNOBS
  static const uint32_t mask = [] {
    const char* e = getenv("ZG_DEBUG_BASE_MASK");
    if (e) fprintf(stderr, "zeth_gpu: ZG_DEBUG_BASE_MASK is set: every MSM result of this process is WRONG by design\n");
    return e ? (uint32_t)strtoul(e, nullptr, 0) & 0x7FFFFFFFu : 0x7FFFFFFFu;
  }();
  return mask;
#else
  return 0x7FFFFFFFu;
#endif
}

// lane j: points [j*64, j*64+64): start (t0 + j*64*d) * G by double-and-add, then repeated + D
template <class F, class Fr>
__global__ void __launch_bounds__(64) k_gen_points(Affine<F> G, Affine<F> D, Fr t0m, Fr dm, size_t n,
                                                    XYZZ<F>* __restrict__ tmp) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t lo = j * GEN_CHUNK;
  if (lo >= n) return;
  size_t hi = lo + GEN_CHUNK;
  if (hi > n) hi = n;
  Fr idx = Fr::zero();
  idx.v[0] = (uint32_t)lo;
  idx.v[1] = (uint32_t)((uint64_t)lo >> 32);
  Fr s = Fr::from_mont(Fr::add(t0m, Fr::mul(Fr::to_mont(idx), dm)));
  uint32_t k[Fr::N];
#pragma unroll
  for (int i = 0; i < Fr::N; i++) k[i] = s.v[i];
  XYZZ<F> acc = xyzz_mul_limbs(XYZZ<F>::from_affine(G), k, Fr::N);
  for (size_t i = lo; i < hi; i++) {
    tmp[i] = acc;
    xyzz_madd(acc, D);
  }
}

// lane i: tmp[i] = s_i * G by double-and-add (fixed-base batch exponentiation, the shape of libff::batch_exp)
template <class F, class Fr>
__global__ void __launch_bounds__(64) k_batch_exp(Affine<F> G, const Fr* __restrict__ scalars, size_t n,
                                                   XYZZ<F>* __restrict__ tmp) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = scalars[i];
  uint32_t k[Fr::N];
#pragma unroll
  for (int j = 0; j < Fr::N; j++) k[j] = s.v[j];
  tmp[i] = xyzz_mul_limbs(XYZZ<F>::from_affine(G), k, Fr::N);
}

// lane j normalises its 64 points with one shared inversion (Montgomery's trick on zzz)
template <class F>
__global__ void __launch_bounds__(64) k_normalize(const XYZZ<F>* __restrict__ tmp, F* __restrict__ pref, size_t n,
                                                   Affine<F>* __restrict__ out) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t lo = j * GEN_CHUNK;
  if (lo >= n) return;
  size_t hi = lo + GEN_CHUNK;
  if (hi > n) hi = n;
  F prod = F::one();
  for (size_t i = lo; i < hi; i++) {
    F z = tmp[i].zzz;
    if (z.is_zero()) z = F::one();
    pref[i] = prod;
    prod = F::mul(prod, z);
  }
  F inv = F::inv(prod);
  for (size_t i = hi; i-- > lo;) {
    XYZZ<F> p = tmp[i];
    bool isinf = p.zzz.is_zero();
    F z = isinf ? F::one() : p.zzz;
    F izzz = F::mul(inv, pref[i]);
    inv = F::mul(inv, z);
    F iz = F::mul(p.zz, izzz);
    F izz = F::sqr(iz);
    Affine<F> a{F::mul(p.x, izz), F::mul(p.y, izzz)};
    out[i] = isinf ? Affine<F>::inf() : a;
  }
}

// ---------------------------------------------------------------------------------------------
// host-side helpers (the O(1) tail of an MSM / proof assembly)
// ---------------------------------------------------------------------------------------------
template <class F>
inline void host_affine_convert(Affine<F>* pts, size_t n, int to_mont) {
  using B = typename BaseField<F>::type;
  B* e = reinterpret_cast<B*>(pts);
  size_t cnt = n * (sizeof(Affine<F>) / sizeof(B));
  for (size_t i = 0; i < cnt; i++) e[i] = to_mont ? B::to_mont(e[i]) : B::from_mont(e[i]);
}

template <class F>
inline XYZZ<F> host_horner(const XYZZ<F>* win, int W, int c) {
  XYZZ<F> acc = win[W - 1];
  for (int w = W - 2; w >= 0; w--) {
    for (int i = 0; i < c; i++) acc = xyzz_dbl(acc);
    xyzz_add(acc, win[w]);
  }
  return acc;
}

template <class F, class Fr>
int host_group_op(int op, const void* a, const void* b, int form, void* out) {
  Affine<F> pa;
  memcpy(&pa, a, sizeof(pa));
  if (form == ZG_FORM_PLAIN) host_affine_convert(&pa, 1, 1);
  XYZZ<F> r;
  if (op == 0) {
    Affine<F> pb;
    memcpy(&pb, b, sizeof(pb));
    if (form == ZG_FORM_PLAIN) host_affine_convert(&pb, 1, 1);
    r = XYZZ<F>::from_affine(pa);
    xyzz_madd(r, pb);
  } else if (op == 1) {
    uint32_t k[Fr::N];
    memcpy(k, b, sizeof(k));
    r = xyzz_mul_limbs(XYZZ<F>::from_affine(pa), k, Fr::N);
  } else {
    return ZG_ERR_INVALID_ARGUMENT;
  }
  Affine<F> o = xyzz_to_affine(r);
  if (form == ZG_FORM_PLAIN) host_affine_convert(&o, 1, 0);
  memcpy(out, &o, sizeof(o));
  return ZG_OK;
}

// ---------------------------------------------------------------------------------------------
// MSM driver
// ---------------------------------------------------------------------------------------------
// Queues the device pipeline on ctx's stream; the W window sums land in ctx->pinned (asynchronous copy).
// msm_finish() waits for them and runs the host tail.  Several contexts ("lanes") can be in flight at once.
// `tab` (may be null): d_pts is a fixed-base table of tab->D levels of tab->N points, T_j[i] = 2^(j c) P_i at
// d_pts[j * N + i], and the MSM runs over the bases [first, first + n) in table mode (msm_table.cuh).  Without it d_pts
// points at the n bases themselves.
template <class F, class Fr>
int msm_launch(zg_ctx* ctx, const void* d_pts, const uint32_t* d_scalars, size_t n, int scalars_mont, int force_c,
               const MsmTable* tab) {
  ctx->job_n = n;
  if (n == 0) return ZG_OK;
  const Affine<F>* d_bases = reinterpret_cast<const Affine<F>*>(d_pts);
  if (tab && (force_c || n < 1024)) {  // an explicit window width, or too few points to sort by group: the plain form on level 0
    d_bases += tab->first;
    tab = nullptr;
  }
  MsmPlan plan = tab ? msm_make_table_plan(n, Fr::Params::BITS, tab->c, tab->D) : msm_make_plan(n, Fr::Params::BITS, force_c);
  if (!tab && ctx->reduce_segments && ctx->reduce_segments < plan.T) {  // fewer, longer bucket segments: less work, longer chain (table mode: segments cost no weighting)
    plan.T = ctx->reduce_segments;
    plan.k = plan.nb / plan.T;
  }
  if (tab && ctx->table_unit_len && ctx->table_unit_len < plan.L) {
    plan.L = ctx->table_unit_len;
    plan.cap = plan.nb + (uint32_t)(plan.E / plan.L) + 1;
  }
  if (tab && ctx->table_segments && ctx->table_segments < plan.T) {  // table mode: the bit-sum tree's work goes with the segment count
    plan.T = ctx->table_segments;
    plan.k = plan.nb / plan.T;
  }
  const MsmPlan& p = plan;
  ctx->job_W = p.W;
  ctx->job_c = p.c;
  ctx->job_table_gbits = tab ? p.gbits : -1;
  const int rows = tab ? 2 * p.W : p.W;  // rows of segment sums; table mode: locally weighted sums, then plain sums
  // what the host reads: the W window sums, or (table mode) the sum of the weighted rows, the plain total and one sum per
  // bit of the flat segment index (k_bitsum)
  unsigned log_nt = 0;
  while (((size_t)1 << log_nt) < (size_t)p.W * p.T) log_nt++;
  const int out_rows = tab ? (int)log_nt + 2 : p.W;
  ctx->job_rows = out_rows;
  ctx->job_log_t = (int)ilog2_floor(p.T);
  ctx->job_log_k = (int)ilog2_floor(p.k);
  ZG_CRUMB("msm_launch: pinned_reserve");
  ZG_TRY(ctx->pinned_reserve((size_t)out_rows * sizeof(XYZZ<F>) + 256));
  ZG_CRUMB("msm_launch: after pinned_reserve");
  const size_t WN = tab ? p.E : (size_t)p.W * n;  // entries
  if ((uint64_t)WN >= (1ull << 31)) return ZG_ERR_UNSUPPORTED;  // 32-bit entry positions with headroom
  const size_t WB = (size_t)p.W * p.nb, WC = (size_t)p.W * p.cap;
  const uint32_t nhi = p.nb >= 512 ? (p.nb >> 8) : 0;  // two-level sort when a window has >= 512 buckets
  size_t need = (nhi ? zg_ctx::ws_round(WN * 4) + zg_ctx::ws_round(WN) + 2 * zg_ctx::ws_round((size_t)p.W * (nhi + 1) * 4) : 0) +
                zg_ctx::ws_round(WN * 2) + zg_ctx::ws_round(WN * 4) + 5 * zg_ctx::ws_round(WB * 4) + zg_ctx::ws_round((WB + 1) * 4) +
                zg_ctx::ws_round((p.W + 1 + 64) * 4) + zg_ctx::ws_round(WC * sizeof(UnitDesc)) + zg_ctx::ws_round(WC * 4) +
                zg_ctx::ws_round((size_t)p.W * UNIT_CLASS_STRIDE * 4) +
                zg_ctx::ws_round(WC * sizeof(XYZZ<F>)) + zg_ctx::ws_round((size_t)rows * p.T * sizeof(XYZZ<F>)) +
                zg_ctx::ws_round((size_t)(log_nt + 2) * sizeof(XYZZ<F>)) +
                (tab ? zg_ctx::ws_round(WN * 4) + 3 * zg_ctx::ws_round(129 * 4) : 0);
  ZG_TRY(ctx->ws_reserve(need));
  int16_t* dig = ctx->ws_take<int16_t>(WN);
  uint32_t* sorted = ctx->ws_take<uint32_t>(WN);
  uint32_t* cnt = ctx->ws_take<uint32_t>(WB);
  uint32_t* cursor = ctx->ws_take<uint32_t>(WB);
  uint32_t* boff = ctx->ws_take<uint32_t>(WB);
  uint32_t* ufirst = ctx->ws_take<uint32_t>(WB);
  uint32_t* ffirst = ctx->ws_take<uint32_t>(WB);
  uint32_t* heavy = ctx->ws_take<uint32_t>(WB + 1);
  uint32_t* wunits = ctx->ws_take<uint32_t>(p.W + 1 + 64);  // [W] unit totals, [W] max units per bucket, then 64 dense-scalar counters
  UnitDesc* units = ctx->ws_take<UnitDesc>(WC);
  uint32_t* order = ctx->ws_take<uint32_t>(WC);
  uint32_t* lencur = ctx->ws_take<uint32_t>((size_t)p.W * UNIT_CLASS_STRIDE);
  XYZZ<F>* part = ctx->ws_take<XYZZ<F>>(WC);
  XYZZ<F>* seg = ctx->ws_take<XYZZ<F>>((size_t)rows * p.T);
  XYZZ<F>* bit_rows = ctx->ws_take<XYZZ<F>>(log_nt + 2);
  uint32_t *fent = nullptr, *gcnt = nullptr, *gstart = nullptr, *gcursor = nullptr;
  if (tab) {
    fent = ctx->ws_take<uint32_t>(WN);
    gcnt = ctx->ws_take<uint32_t>(129);
    gstart = ctx->ws_take<uint32_t>(129);
    gcursor = ctx->ws_take<uint32_t>(129);
    if (!fent || !gcnt || !gstart || !gcursor) return ZG_ERR_OUT_OF_MEMORY;
  }
  uint16_t* fkey = reinterpret_cast<uint16_t*>(dig);  // table mode: bucket-within-group of every entry (same size as dig)
  uint32_t *mid_ent = nullptr, *cursor1 = nullptr, *pstart = nullptr;
  uint8_t* mid_lo = nullptr;
  if (nhi) {
    mid_ent = ctx->ws_take<uint32_t>(WN);
    mid_lo = ctx->ws_take<uint8_t>(WN);
    cursor1 = ctx->ws_take<uint32_t>((size_t)p.W * (nhi + 1));
    pstart = ctx->ws_take<uint32_t>((size_t)p.W * (nhi + 1));
    if (!mid_ent || !mid_lo || !cursor1 || !pstart) return ZG_ERR_OUT_OF_MEMORY;
  }
  if (!dig || !sorted || !cnt || !cursor || !boff || !ufirst || !ffirst || !heavy || !wunits || !units || !order || !lencur || !part || !seg || !bit_rows)
    return ZG_ERR_OUT_OF_MEMORY;

  hipStream_t st = ctx->stream;
  const size_t lds = (size_t)p.nb * 4;
  static std::once_flag attr_once;  // several lanes of a proof may arrive here at once
  hipError_t attr_err = hipSuccess;
  ZG_CRUMB("msm_launch: call_once hipFuncSetAttribute");
  std::call_once(attr_once, [&] {
    attr_err = hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 17);
    if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 17);
    if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_hist_table, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 17);
    if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_scatter_table, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 17);
  });
  ZG_HIP(attr_err);
  ctx->prof_begin();
  ctx->mark("digits");
  ZG_CRUMB("msm_launch: first kernel launch (k_zero_u32)");
  // table mode with at most two bucket groups: the two-ended layout (msm_table.cuh), no counting pass
  static const bool no_two_ended = zg_knob("ZG_MSM_NO_TWO_ENDED") != nullptr;  // measurement knob
  const bool two_ended = tab && p.W <= 2 && !no_two_ended;
  uint32_t* gse = two_ended ? gcnt : (uint32_t*)nullptr;  // (the counts' array is free in this layout)
  const uint32_t* gstart_r = two_ended ? gse : gstart;
  const uint32_t* gend_r = two_ended ? gse + 2 : gstart + 1;
  hipLaunchKernelGGL(k_zero_u32, dim3((unsigned)((WB + 255) / 256)), dim3(256), 0, st, cnt, WB, wunits,
                     (size_t)(p.W + 1 + 64), heavy, gcnt, (size_t)(tab && !two_ended ? 129 : 0), gse, (uint32_t)WN);
  uint32_t* dense_counters = ctx->want_dense ? wunits + p.W + 1 : (uint32_t*)nullptr;
  if (tab) {
    // level 0 of the sort: entries by bucket group
    uint32_t tile_scalars = TABLE_TILE / (uint32_t)p.D;
    if (tile_scalars > 1024) tile_scalars = 1024;
    size_t cblocks = (n + 255) / 256;
    if (cblocks > 2048) cblocks = 2048;
    // (k_table_starts stays a launch of its own: folding it into the last workgroup of k_table_count needs a device-wide
    // fence per workgroup, which on this part writes back and invalidates an XCD's L2 -- digits phase 0.78 -> 1.07 ms)
    if (!two_ended) {
      hipLaunchKernelGGL((k_table_count<Fr>), dim3((unsigned)cblocks), dim3(256), 0, st, d_scalars, n, scalars_mont, p.c, p.D, p.K,
                         (uint32_t)p.W - 1u, gcnt, dense_counters);
      hipLaunchKernelGGL(k_table_starts, dim3(1), dim3(64), 0, st, gcnt, (uint32_t)p.W, gstart, gcursor);
    }
    const dim3 sgrid((unsigned)((n + tile_scalars - 1) / tile_scalars));
    if (p.c == 22 && p.D == msm_table_digits(Fr::Params::BITS, 22))
      hipLaunchKernelGGL((k_table_scatter_static<Fr, 22>), sgrid, dim3(1024), 0, st, d_scalars, n, scalars_mont, p.K, p.gbits,
                         (uint32_t)p.W, tile_scalars, (uint32_t)tab->N, (uint32_t)tab->first, gcursor, gse, fent, fkey);
    else if (p.c == 20 && p.D == msm_table_digits(Fr::Params::BITS, 20))
      hipLaunchKernelGGL((k_table_scatter_static<Fr, 20>), sgrid, dim3(1024), 0, st, d_scalars, n, scalars_mont, p.K, p.gbits,
                         (uint32_t)p.W, tile_scalars, (uint32_t)tab->N, (uint32_t)tab->first, gcursor, gse, fent, fkey);
    else if (p.c == 16 && p.D == msm_table_digits(Fr::Params::BITS, 16))
      hipLaunchKernelGGL((k_table_scatter_static<Fr, 16>), sgrid, dim3(1024), 0, st, d_scalars, n, scalars_mont, p.K, p.gbits,
                         (uint32_t)p.W, tile_scalars, (uint32_t)tab->N, (uint32_t)tab->first, gcursor, gse, fent, fkey);
    else if (p.c == 17 && p.D == msm_table_digits(Fr::Params::BITS, 17))  // (the prover's tables up to 2^19 points)
      hipLaunchKernelGGL((k_table_scatter_static<Fr, 17>), sgrid, dim3(1024), 0, st, d_scalars, n, scalars_mont, p.K, p.gbits,
                         (uint32_t)p.W, tile_scalars, (uint32_t)tab->N, (uint32_t)tab->first, gcursor, gse, fent, fkey);
    else
      hipLaunchKernelGGL((k_table_scatter<Fr>), sgrid, dim3(1024), 0, st, d_scalars, n, scalars_mont, p.c, p.D, p.K, p.gbits,
                         (uint32_t)p.W, tile_scalars, (uint32_t)tab->N, (uint32_t)tab->first, gcursor, gse, fent, fkey);
    ctx->mark("hist");
    hipLaunchKernelGGL(k_hist_table, dim3(p.nchunk, p.W), dim3(1024), lds, st, fkey, gstart_r, gend_r, p.nb, cnt);
  } else {
    hipLaunchKernelGGL((k_digits<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n, scalars_mont,
                       p.c, p.W, p.K, dig, dense_counters);
    ctx->mark("hist");
    hipLaunchKernelGGL(k_hist, dim3(p.nchunk, p.W), dim3(1024), lds, st, dig, n, p.nb, p.nchunk, cnt);
  }
  ctx->mark("scan");
  ZG_CRUMB("msm_launch: k_scan launch");
  hipLaunchKernelGGL(k_scan, dim3(p.W), dim3(1024), 0, st, cnt, n, p.nb, p.L, p.cap, cursor, boff, ufirst, wunits, nhi,
                     cursor1, pstart, lencur, ffirst, tab ? gstart_r : gstart);
  ctx->mark("scatter");
  // Window groups (experiment, off unless ZG_MSM_SPLIT is set): the top `g0` windows are sorted first and accumulated at
  // once while the remaining windows are sorted on a second stream.  Measured at 2^24 points: 35.0 ms against 30.4 ms
  // for the plain sequence -- the sort's 1024-lane workgroups displace accumulation waves for longer than the sort
  // takes on its own (k_accum 25.1 -> 31.6 ms), so the sequence stays sort-then-accumulate.
  const uint32_t Wn = (uint32_t)p.W;
  const bool split = !tab && nhi && Wn >= 8 && n >= ((size_t)1 << 21) && zg_knob("ZG_MSM_SPLIT");
  const uint32_t g0 = split ? (Wn + 3) / 4 : Wn;  // windows [Wn - g0, Wn) form the first group
  const uint32_t w_lo = Wn - g0;
  if (split) ZG_TRY(ctx->side_init());
  const size_t per_group = WN / Wn;  // entries of a window / an average bucket group
  auto scatter_group = [&](hipStream_t s, uint32_t w0, uint32_t cntw) {
    if (tab) {
      if (nhi) {
        size_t t1 = (2 * per_group + SORT_TILE1T - 1) / SORT_TILE1T;
        if (t1 < 1) t1 = 1;
        if (t1 > 8192) t1 = 8192;
        hipLaunchKernelGGL(k_scatter_hi_table, dim3((unsigned)t1, cntw), dim3(SORT_WG), 0, s, fent, fkey, gstart_r, gend_r, nhi, cursor1, mid_ent,
                           mid_lo);
        size_t tiles = (2 * per_group / nhi + SORT_TILE2 - 1) / SORT_TILE2;
        if (tiles < 1) tiles = 1;
        if (tiles > 64) tiles = 64;
        hipLaunchKernelGGL(k_scatter_lo, dim3((unsigned)tiles, nhi, cntw), dim3(SORT_WG), 0, s, mid_ent, mid_lo, pstart, nhi, p.nb,
                           cursor, sorted, w0);
      } else {
        hipLaunchKernelGGL(k_scatter_table, dim3(p.nchunk, p.W), dim3(1024), lds, s, fent, fkey, gstart_r, gend_r, p.nb, cursor, sorted);
      }
    } else if (nhi) {
      hipLaunchKernelGGL(k_scatter_hi, dim3((unsigned)((n + SORT_TILE1 - 1) / SORT_TILE1), cntw), dim3(1024), 0, s, dig, n, nhi,
                         cursor1, mid_ent, mid_lo, w0);
      size_t tiles = (2 * n / nhi + SORT_TILE2 - 1) / SORT_TILE2;
      if (tiles < 1) tiles = 1;
      if (tiles > 64) tiles = 64;
      hipLaunchKernelGGL(k_scatter_lo, dim3((unsigned)tiles, nhi, cntw), dim3(SORT_WG), 0, s, mid_ent, mid_lo, pstart, nhi, p.nb,
                         cursor, sorted, w0);
    } else {
      hipLaunchKernelGGL(k_scatter, dim3(p.nchunk, p.W), dim3(1024), lds, s, dig, n, p.nb, p.nchunk, cursor, sorted);
    }
  };
  if (split) {
    ZG_HIP(hipEventRecord(ctx->ev_fork, st));
    ZG_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
  }
  scatter_group(st, w_lo, g0);
  if (split) {
    scatter_group(ctx->side_stream, 0, w_lo);
    ZG_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
  }
  ctx->mark("units");
  hipLaunchKernelGGL(k_units, dim3((unsigned)((WB + 255) / 256)), dim3(256), 0, st, cnt, boff, ufirst, ffirst, p.nb,
                     (uint32_t)p.W, p.L, p.cap, units, lencur, order, heavy, p.smax);
  hipLaunchKernelGGL(k_units_heavy, dim3(8, 128), dim3(256), 0, st, cnt, boff, ufirst, ffirst, p.nb, p.L, p.cap, units, lencur,
                     order, heavy, p.smax);
  ctx->mark("accum");
  ZG_CRUMB("msm_launch: k_accum launch");
  const dim3 ablk(ZG_ACCUM_BLOCK);
  const uint32_t ablocks = (p.cap + ZG_ACCUM_BLOCK - 1) / ZG_ACCUM_BLOCK;
  hipLaunchKernelGGL((k_accum<F>), tab ? dim3(g0, ablocks) : dim3(ablocks, g0), ablk, 0, st, units, wunits, p.cap, order, sorted,
                     d_bases, part, zg_debug_base_mask(), Wn, tab ? 1u : 0u);
  if (split) {
    ZG_HIP(hipStreamWaitEvent(st, ctx->ev_join, 0));
    hipLaunchKernelGGL((k_accum<F>), dim3((p.cap + ZG_ACCUM_BLOCK - 1) / ZG_ACCUM_BLOCK, w_lo), ablk, 0, st, units, wunits, p.cap,
                       order, sorted, d_bases, part, zg_debug_base_mask(), w_lo, 0u);
  }
  ctx->mark("combine");
  hipLaunchKernelGGL((k_combine_light<F>), dim3((4 * p.nb + 255) / 256, p.W), dim3(256), 0, st, wunits, (uint32_t)p.W, p.nb, cnt,
                     ufirst, p.L, part, p.smax);
  {
    uint32_t chunks = (p.cap + COMBINE_CHUNK - 1) / COMBINE_CHUNK;
    if (chunks > 64) chunks = 64;
    // COMBINE_HEAVY_SLOTS heavy buckets side by side, the rest in turn (the 4096 values of a 12-bit top digit are heavy
    // from 2^25 points on); workgroups beyond the heavy count or a bucket's chunk count leave at once -- a JoinSplit
    // witness has ONE heavy bucket, so the grid is kept small: a million waves launched to exit cost the proof's other
    // streams their dispatch slots
    hipLaunchKernelGGL((k_combine_tree<F>), dim3(chunks, COMBINE_HEAVY_SLOTS), dim3(quad_block<F>()), 0, st, heavy, cnt, ufirst, p.L, 1u, part, wunits + p.W);
    hipLaunchKernelGGL((k_combine_tree<F>), dim3(1, COMBINE_HEAVY_SLOTS), dim3(quad_block<F>()), 0, st, heavy, cnt, ufirst, p.L, COMBINE_CHUNK, part, wunits + p.W);
    if (WN / Wn / p.L >= (size_t)COMBINE_CHUNK * COMBINE_CHUNK || (tab && WN / p.L >= (size_t)COMBINE_CHUNK * COMBINE_CHUNK))  // more than 2^18 units in one bucket: a third level
      hipLaunchKernelGGL((k_combine_tree<F>), dim3(1, 16), dim3(quad_block<F>()), 0, st, heavy, cnt, ufirst, p.L,
                         COMBINE_CHUNK * COMBINE_CHUNK, part, wunits + p.W);
  }
  ctx->mark("reduce1");
  ZG_CRUMB("msm_launch: k_reduce1 launch");
  hipLaunchKernelGGL((k_reduce1<F>), dim3((4 * p.T * p.W + 127) / 128), dim3(128), 0, st, cnt, ufirst, part, p.nb, p.T,
                     p.k, p.L, (uint32_t)p.W, seg, tab ? seg + (size_t)p.W * p.T : (XYZZ<F>*)nullptr, tab ? 0u : 1u);
  ctx->mark("reduce2");
  if (tab) {
    // the weights come from the bit sums of the plain segment sums (k_bitsum): blocks of two entries per quad first,
    // then one block over the block heads
    const uint32_t nq = (uint32_t)quad_block<F>() / 4;
    unsigned chunk_log = 0;
    while ((1u << chunk_log) < 2 * nq) chunk_log++;
    XYZZ<F>* X = seg + (size_t)p.W * p.T;
    if (log_nt <= chunk_log) {
      hipLaunchKernelGGL((k_bitsum<F>), dim3(1), dim3(4 * nq), 0, st, X, seg, 0u, log_nt, bit_rows);
    } else {
      hipLaunchKernelGGL((k_bitsum<F>), dim3(1u << (log_nt - chunk_log)), dim3(4 * nq), 0, st, X, seg, 0u, chunk_log, (XYZZ<F>*)nullptr);
      hipLaunchKernelGGL((k_bitsum<F>), dim3(1), dim3(4 * nq), 0, st, X, seg, chunk_log, log_nt, bit_rows);
    }
    ctx->mark("end");
    ZG_HIP(hipGetLastError());
    ZG_CRUMB("msm_launch: hipMemcpyAsync D2H");
    ZG_HIP(hipMemcpyAsync(ctx->pinned, bit_rows, (size_t)out_rows * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
  } else {
    const uint32_t nq = (uint32_t)quad_block<F>() / 4;  // quads of a full block
    if (p.T > nq && p.T % nq == 0) {
      // chunks of one entry per quad, then the chunk heads
      hipLaunchKernelGGL((k_reduce2<F>), dim3(rows, p.T / nq), dim3(4 * nq), 0, st, seg, p.T, nq, 1u, 0);
      uint32_t t2 = 4 * (p.T / nq);
      if (t2 > 4 * nq) t2 = 4 * nq;
      if (t2 < 64) t2 = 64;
      hipLaunchKernelGGL((k_reduce2<F>), dim3(rows, 1), dim3(t2), 0, st, seg, p.T, p.T / nq, nq, 1);
    } else {
      uint32_t t2 = 4 * p.T;  // a quad per entry, up to the block the field's registers allow; at least a wave
      if (t2 > 4 * nq) t2 = 4 * nq;
      if (t2 < 64) t2 = 64;
      hipLaunchKernelGGL((k_reduce2<F>), dim3(rows, 1), dim3(t2), 0, st, seg, p.T, p.T, 1u, 1);
    }
    ctx->mark("end");
    ZG_HIP(hipGetLastError());
    ZG_CRUMB("msm_launch: hipMemcpy2DAsync D2H");
    // gather the W window sums (strided by T) with one 2D copy into pinned host memory
    ZG_HIP(hipMemcpy2DAsync(ctx->pinned, sizeof(XYZZ<F>), seg, (size_t)p.T * sizeof(XYZZ<F>), sizeof(XYZZ<F>), rows,
                            hipMemcpyDeviceToHost, st));
  }
  ZG_CRUMB("msm_launch: hipMemcpyAsync D2H (dense counters)");
  ZG_HIP(hipMemcpyAsync((char*)ctx->pinned + (size_t)out_rows * sizeof(XYZZ<F>), wunits + p.W + 1, 256, hipMemcpyDeviceToHost, st));
  ZG_CRUMB("msm_launch: queued");
  return ZG_OK;
}

static constexpr int ZG_FORM_XYZZ_INTERNAL = 2;  // msm_finish only: leave the sum as an XYZZ point (Montgomery)

template <class F>
int msm_finish(zg_ctx* ctx, int out_form, void* out_affine_host) {
  Affine<F> res = Affine<F>::inf();
  XYZZ<F> acc = XYZZ<F>::inf();
  if (ctx->job_n > 0) {
    ZG_CRUMB("msm_finish: hipStreamSynchronize");
    ZG_HIP(zg_stream_wait(ctx->stream));
    ZG_CRUMB("msm_finish: synchronised");
    ctx->prof_end();
    const XYZZ<F>* rows = reinterpret_cast<const XYZZ<F>*>(ctx->pinned);
    const int nrows = ctx->job_rows;
    if (ctx->job_table_gbits >= 0) {
      // rows: Y = sum of the segments' locally weighted sums, S = sum of all buckets, BIT_t = sum of the plain segment sums
      // whose flat index (group * T + j) has bit t set.  Bucket (group g, key kappa = j k + i) weighs kappa * W + g + 1:
      //   sum = W * (R - S) + S + sum_t 2^t BIT_(log T + t),   R = Y + k * sum_(t < log T) 2^t BIT_t
      const XYZZ<F>& Y = rows[0];
      const XYZZ<F>& S = rows[1];
      const XYZZ<F>* BIT = rows + 2;
      const int log_t = ctx->job_log_t, gbits = ctx->job_table_gbits;
      XYZZ<F> r = XYZZ<F>::inf();
      for (int t = log_t - 1; t >= 0; t--) {
        r = xyzz_dbl(r);
        xyzz_add(r, BIT[t]);
      }
      for (int i = 0; i < ctx->job_log_k; i++) r = xyzz_dbl(r);
      xyzz_add(r, Y);
      xyzz_add(r, xyzz_neg(S));  // R - S
      for (int i = 0; i < gbits; i++) r = xyzz_dbl(r);
      XYZZ<F> gsum = XYZZ<F>::inf();
      for (int t = gbits - 1; t >= 0; t--) {
        gsum = xyzz_dbl(gsum);
        xyzz_add(gsum, BIT[log_t + t]);
      }
      xyzz_add(r, gsum);
      xyzz_add(r, S);
      acc = r;
    } else {
      acc = host_horner(rows, ctx->job_W, ctx->job_c);
    }
    const uint32_t* dc = reinterpret_cast<const uint32_t*>((const char*)ctx->pinned + (size_t)nrows * sizeof(XYZZ<F>));
    ctx->job_dense = 0;
    for (int i = 0; i < 64; i++) ctx->job_dense += dc[i];
  }
  if (out_form == ZG_FORM_XYZZ_INTERNAL) {
    memcpy(out_affine_host, &acc, sizeof(acc));
    return ZG_OK;
  }
  res = xyzz_to_affine(acc);
  if (out_form == ZG_FORM_PLAIN) host_affine_convert(&res, 1, 0);
  memcpy(out_affine_host, &res, sizeof(res));
  return ZG_OK;
}

template <class F, class Fr>
int msm_run(zg_ctx* ctx, const void* d_pts, const uint32_t* d_scalars, size_t n, int scalars_mont, int force_c,
            int out_form, void* out_affine_host, const MsmTable* tab) {
  ZG_TRY((msm_launch<F, Fr>(ctx, d_pts, d_scalars, n, scalars_mont, force_c, tab)));
  return msm_finish<F>(ctx, out_form, out_affine_host);
}

// digit width (0: chosen from N) and level count of a fixed-base table over N bases
template <class Fr>
int table_params(size_t N, int window_bits, int* c, int* D) {
  if (window_bits != 0 && (window_bits < MSM_TABLE_MIN_C || window_bits > MSM_TABLE_MAX_C)) return ZG_ERR_INVALID_ARGUMENT;
  *c = window_bits ? window_bits : msm_table_pick_c(N);
  *D = msm_table_digits(Fr::Params::BITS, *c);
  return ZG_OK;
}

// Levels 1 .. D-1 of a fixed-base table from level 0 (d_table[0 .. N)): T_j = 2^c T_{j-1}, normalised to affine
template <class F>
int table_build(zg_ctx* ctx, void* d_table, size_t N, int c, int D) {
  if (N == 0) return ZG_OK;
  size_t need = zg_ctx::ws_round(N * sizeof(XYZZ<F>)) + zg_ctx::ws_round(N * sizeof(F));
  ZG_TRY(ctx->ws_reserve(need));
  XYZZ<F>* tmp = ctx->ws_take<XYZZ<F>>(N);
  F* pref = ctx->ws_take<F>(N);
  if (!tmp || !pref) return ZG_ERR_OUT_OF_MEMORY;
  Affine<F>* t = reinterpret_cast<Affine<F>*>(d_table);
  const size_t lanes = (N + GEN_CHUNK - 1) / GEN_CHUNK;
  for (int j = 1; j < D; j++) {
    hipLaunchKernelGGL((k_table_next<F>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, t + (size_t)(j - 1) * N, N, c,
                       tmp);
    hipLaunchKernelGGL((k_normalize<F>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, ctx->stream, tmp, pref, N,
                       t + (size_t)j * N);
  }
  ZG_HIP(hipGetLastError());
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  return ZG_OK;
}

template <class F>
int bases_to_mont(zg_ctx* ctx, void* d_pts, size_t n) {
  using B = typename BaseField<F>::type;
  size_t cnt = n * (sizeof(Affine<F>) / sizeof(B));
  if (cnt == 0) return ZG_OK;
  hipLaunchKernelGGL((k_to_mont<B>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                     reinterpret_cast<B*>(d_pts), cnt);
  ZG_HIP(hipGetLastError());
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  return ZG_OK;
}

template <class F, class Fr, class Gen>
int bases_generate(zg_ctx* ctx, void* d_pts, size_t n, const uint32_t* t0, const uint32_t* d) {
  if (n == 0) return ZG_OK;
  Fr t0p, dp;
  memcpy(&t0p, t0, sizeof(Fr));
  memcpy(&dp, d, sizeof(Fr));
  Affine<F> G = Gen::gen();
  uint32_t k[Fr::N];
  memcpy(k, d, sizeof(k));
  Affine<F> D = xyzz_to_affine(xyzz_mul_limbs(XYZZ<F>::from_affine(G), k, Fr::N));
  size_t need = zg_ctx::ws_round(n * sizeof(XYZZ<F>)) + zg_ctx::ws_round(n * sizeof(F));
  ZG_TRY(ctx->ws_reserve(need));
  XYZZ<F>* tmp = ctx->ws_take<XYZZ<F>>(n);
  F* pref = ctx->ws_take<F>(n);
  if (!tmp || !pref) return ZG_ERR_OUT_OF_MEMORY;
  size_t lanes = (n + GEN_CHUNK - 1) / GEN_CHUNK;
  unsigned blocks = (unsigned)((lanes + 63) / 64);
  hipLaunchKernelGGL((k_gen_points<F, Fr>), dim3(blocks), dim3(64), 0, ctx->stream, G, D, Fr::to_mont(t0p),
                     Fr::to_mont(dp), n, tmp);
  hipLaunchKernelGGL((k_normalize<F>), dim3(blocks), dim3(64), 0, ctx->stream, tmp, pref, n,
                     reinterpret_cast<Affine<F>*>(d_pts));
  ZG_HIP(hipGetLastError());
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  return ZG_OK;
}

// P_i = s_i * G for n plain scalars in device memory
template <class F, class Fr, class Gen>
int bases_batch_exp(zg_ctx* ctx, void* d_pts, size_t n, const uint32_t* d_scalars) {
  if (n == 0) return ZG_OK;
  size_t need = zg_ctx::ws_round(n * sizeof(XYZZ<F>)) + zg_ctx::ws_round(n * sizeof(F));
  ZG_TRY(ctx->ws_reserve(need));
  XYZZ<F>* tmp = ctx->ws_take<XYZZ<F>>(n);
  F* pref = ctx->ws_take<F>(n);
  if (!tmp || !pref) return ZG_ERR_OUT_OF_MEMORY;
  hipLaunchKernelGGL((k_batch_exp<F, Fr>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, Gen::gen(),
                     reinterpret_cast<const Fr*>(d_scalars), n, tmp);
  size_t lanes = (n + GEN_CHUNK - 1) / GEN_CHUNK;
  hipLaunchKernelGGL((k_normalize<F>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, ctx->stream, tmp, pref, n,
                     reinterpret_cast<Affine<F>*>(d_pts));
  ZG_HIP(hipGetLastError());
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  return ZG_OK;
}

template <class F>
void host_convert_thunk(void* pts, size_t n, int to_mont) {
  host_affine_convert(reinterpret_cast<Affine<F>*>(pts), n, to_mont);
}

// sum of n affine points on the host (one normalisation): the combine step of a point-range sharded MSM, after the
// ranks' partial results have been all-gathered
template <class F>
int host_group_sum(const void* pts, size_t n, int form, void* out) {
  const char* src = reinterpret_cast<const char*>(pts);
  XYZZ<F> acc = XYZZ<F>::inf();
  for (size_t i = 0; i < n; i++) {
    Affine<F> p;
    memcpy(&p, src + i * sizeof(p), sizeof(p));
    if (form == ZG_FORM_PLAIN) host_affine_convert(&p, 1, 1);
    xyzz_madd(acc, p);
  }
  Affine<F> o = xyzz_to_affine(acc);
  if (form == ZG_FORM_PLAIN) host_affine_convert(&o, 1, 0);
  memcpy(out, &o, sizeof(o));
  return ZG_OK;
}

// libff's G::is_well_formed() on affine input: the identity, or Y^2 = X^3 + b (host only; used by the key loader
// the way groth16_snark.tcc:193-224 checks a proving key after reading it).  *first_bad = n when all points pass.
template <class F, class Gen>
int host_points_check(const void* pts, size_t n, int form, size_t* first_bad) {
  const F b = Gen::coeff_b();
  const char* src = reinterpret_cast<const char*>(pts);
  *first_bad = n;
  for (size_t i = 0; i < n; i++) {
    Affine<F> p;
    memcpy(&p, src + i * sizeof(p), sizeof(p));
    if (p.is_inf()) continue;
    if (form == ZG_FORM_PLAIN) host_affine_convert(&p, 1, 1);
    if (F::sqr(p.y) != F::add(F::mul(F::sqr(p.x), p.x), b)) {
      *first_bad = i;
      return ZG_OK;
    }
  }
  return ZG_OK;
}

template <class Fq, class GP>
struct G1Gen {
  static Affine<Fq> gen() {
    Affine<Fq> g;
    for (int i = 0; i < Fq::N; i++) {
      g.x.v[i] = GP::GX_MONT[i];
      g.y.v[i] = GP::GY_MONT[i];
    }
    return g;
  }
  static Fq coeff_b() {
    Fq b;
    for (int i = 0; i < Fq::N; i++) b.v[i] = GP::B_MONT[i];
    return b;
  }
};
template <class Fq2T, class GP>
struct G2Gen {
  static Affine<Fq2T> gen() {
    Affine<Fq2T> g;
    for (int i = 0; i < Fq2T::Base::N; i++) {
      g.x.c0.v[i] = GP::GX_C0_MONT[i];
      g.x.c1.v[i] = GP::GX_C1_MONT[i];
      g.y.c0.v[i] = GP::GY_C0_MONT[i];
      g.y.c1.v[i] = GP::GY_C1_MONT[i];
    }
    return g;
  }
  static Fq2T coeff_b() {
    Fq2T b;
    for (int i = 0; i < Fq2T::Base::N; i++) {
      b.c0.v[i] = GP::B_C0_MONT[i];
      b.c1.v[i] = GP::B_C1_MONT[i];
    }
    return b;
  }
};

}  // namespace zg

#define ZG_DEFINE_GROUP_OPS(NAME, F, FR, GEN)                                                     \
  const GroupOps* NAME() {                                                                        \
    static GroupOps ops = {sizeof(zg::Affine<F>),                                            \
                                (size_t)zg::BaseField<F>::type::N,                                \
                                (int)(sizeof(zg::Affine<F>) / sizeof(zg::BaseField<F>::type)),    \
                                &zg::msm_run<F, FR>,                                              \
                                &zg::msm_launch<F, FR>,                                           \
                                &zg::msm_finish<F>,                                               \
                                &zg::bases_to_mont<F>,                                            \
                                &zg::bases_generate<F, FR, GEN>,                                  \
                                &zg::bases_batch_exp<F, FR, GEN>,                                 \
                                &zg::host_group_op<F, FR>,                                        \
                                &zg::host_convert_thunk<F>,                                       \
                                &zg::host_points_check<F, GEN>,                                   \
                                &zg::group_fft<F, FR>,                                            \
                                &zg::group_scale<F, FR>,                                          \
                                &zg::group_combine<F, FR>,                                        \
                                &zg::host_group_sum<F>,                                           \
                                &zg::table_build<F>,                                              \
                                &zg::table_params<FR>};                                           \
    return &ops;                                                                                  \
  }
