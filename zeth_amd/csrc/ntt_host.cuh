// Host driver for the Fr evaluation-domain transforms: per-(curve, log m) twiddle tables cached in the context
// (the analogue of constructing a libfqfft::basic_radix2_domain<Fr>(m) once), launch geometry, fused scalings.
#pragma once
#include <stdlib.h>

#include <mutex>

#include "common.cuh"
#include "ntt.cuh"

namespace zg {

template <class Fr>
struct NttDomain : NttDomainBase {
  NttTables<Fr> T;
  Fr* storage = nullptr;
  // the full-size tables (ntt.cuh, NttTables::twf ...): five vectors of m elements in one allocation -- twf, twf_inv,
  // in_tab, then the two output tables g^-j m^-1 (icosetFFT) and g^j m^-1 (an iFFT feeding a coset FFT); null when the
  // memory could not be had or the domain lives in the static scratch (the two-level forms serve then)
  Fr* full = nullptr;
  const Fr *out_icoset = nullptr, *out_fuse = nullptr;
  uint32_t TC = 1, TR = 1;
  ~NttDomain() override {
    if (storage) hipFree(storage);
    if (full) hipFree(full);
  }
};

// launch flags of ntt_launch
static constexpr int NTT_FUSE_COSET_NEXT = 1;  // iFFT only: outputs are also scaled by g^j -- the following transform is a plain FFT
static constexpr int NTT_QAP_IN = 2;           // icosetFFT only: the input is (A B - C) / Z(g) of three vectors (k_ntt_pass_a)

template <class Fr>
inline Fr fr_from_limbs(const uint32_t* l) {
  Fr r;
  for (int i = 0; i < Fr::N; i++) r.v[i] = l[i];
  return r;
}

template <class Fr>
inline Fr host_pow_u64(Fr b, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = Fr::mul(r, b);
    b = Fr::sqr(b);
    e >>= 1;
  }
  return r;
}

// Elements of a workgroup's tile where a pass's transforms fit it: 2048 (74 KB of LDS) lets TWO workgroups share a CU,
// so one's loads and stores overlap the other's stages -- 2^22: 0.50 against 0.53 ms with 4096-element tiles, 2^21: 0.26
// against 0.275; passes of 4096-point transforms (m >= 2^23) need the full 4096.  ZG_NTT_TILE_ELEMS = 2048 | 4096 (tuning
// knob).
inline uint32_t ntt_tile_elems() {
  static const uint32_t v = [] {
    const char* e = zg_knob("ZG_NTT_TILE_ELEMS");
    const uint32_t x = e ? (uint32_t)atoi(e) : 0u;
    return (x == 2048u || x == 4096u) ? x : 2048u;
  }();
  return v;
}
inline uint32_t ntt_pick_tile(uint32_t n_this, uint32_t n_other) {
  uint32_t t = (n_this <= ntt_tile_elems() ? ntt_tile_elems() : (uint32_t)NTT_LDS_ELEMS) / n_this;
  if (t > n_other) t = n_other;
  if (t < 1) t = 1;
  // keep at least ~512 workgroups in flight when the transform is large enough to allow it
  while (t > 1 && n_other / t < 512) t >>= 1;
  return t;
}

// split of log m over the two passes; false when this build cannot run the size
inline bool ntt_split(unsigned log_m, unsigned* l1, unsigned* l2) {
  *l1 = log_m <= 12 ? log_m : (log_m + 1) / 2;
  // From 2^21 points pass A takes one stage fewer than the even split where pass B has room for it: pass A's column tiles
  // are then two elements wide instead of one (64-byte runs), which is worth more than pass B's longer rows cost -- 2^21
  // 0.259 -> 0.255 ms, 2^22 0.531 -> 0.521, 2^23 1.145 -> 1.116 (profiles/r3_ntt_split.jsonl; one stage MORE loses 7 %).
  // (tuning knob ZG_NTT_L1_DELTA: pass A's share of the stages, relative to that)
  if (log_m >= 21 && log_m - (*l1 - 1) <= 12) *l1 -= 1;
  static const int delta = zg_knob("ZG_NTT_L1_DELTA") ? atoi(zg_knob("ZG_NTT_L1_DELTA")) : 0;
  if (log_m > 12 && delta && (int)*l1 + delta >= 1 && (int)*l1 + delta <= 12 && log_m - (unsigned)((int)*l1 + delta) <= 12)
    *l1 = (unsigned)((int)*l1 + delta);
  *l2 = log_m - *l1;
  return *l1 <= 12;
}

// Table storage of a domain of 2^log_m points, in 32-byte units (sizeof Fr): w^i and w^-i (m/2 each) and the four coset
// tables in the stored form, then the per-pass butterfly twiddles and g_hi in the reduced-radix form (36 bytes an entry,
// ntt.cuh)
inline size_t ntt_table29_entries(unsigned log_m) {
  unsigned l1 = 0, l2 = 0;
  ntt_split(log_m, &l1, &l2);
  return 2 * ((size_t)1 << l1) + ((size_t)1 << l2);  // twa and twa_inv (N1/2 each), twb and twb_inv (N2/2 each), g_hi (N1)
}
inline size_t ntt_table_elems(unsigned log_m) {
  unsigned l1 = 0, l2 = 0;
  ntt_split(log_m, &l1, &l2);
  const size_t m = (size_t)1 << log_m, half = m > 1 ? m / 2 : 1;
  return 2 * half + 2 * ((size_t)1 << l1) + 2 * ((size_t)1 << l2) + (ntt_table29_entries(log_m) * 36 + 31) / 32;
}

// Points D at `storage` (ntt_table_elems(log_m) elements) and, when `fill` is set, queues the kernels that compute the
// tables there on the context's stream (fill = false: the storage already holds this domain's tables).
template <class Fr>
int ntt_fill_tables(zg_ctx* ctx, unsigned log_m, Fr* storage, NttDomain<Fr>* D, bool fill = true) {
  using P = typename Fr::Params;
  unsigned l1 = 0, l2 = 0;
  if (log_m > (unsigned)P::TWO_ADICITY || log_m > 26 || !ntt_split(log_m, &l1, &l2)) return ZG_ERR_UNSUPPORTED;
  const uint64_t m = 1ull << log_m;
  const uint32_t N1 = 1u << l1, N2 = 1u << l2;
  const size_t half = m > 1 ? m / 2 : 1;
  Fr* p = storage;
  Fr* tw = p; p += half;
  Fr* twi = p; p += half;
  Fr* g_hi = p; p += N1;
  Fr* g_lo = p; p += N2;
  Fr* gi_lo = p; p += N1;
  Fr* gi_hi = p; p += N2;
  uint32_t* twa29 = reinterpret_cast<uint32_t*>(p);
  uint32_t* twai29 = twa29 + 9 * (size_t)(N1 > 1 ? N1 / 2 : 1);
  uint32_t* twb29 = twa29 + 9 * (size_t)N1;
  uint32_t* twbi29 = twb29 + 9 * (size_t)(N2 > 1 ? N2 / 2 : 1);
  uint32_t* g_hi29 = twb29 + 9 * (size_t)N2;
  // w = root_of_unity^(2^(s - log m))  (libfqfft get_root_of_unity)
  Fr w = fr_from_limbs<Fr>(P::ROOT_MONT), wi = fr_from_limbs<Fr>(P::ROOT_INV_MONT);
  for (unsigned i = log_m; i < (unsigned)P::TWO_ADICITY; i++) {
    w = Fr::sqr(w);
    wi = Fr::sqr(wi);
  }
  Fr g = fr_from_limbs<Fr>(P::GEN_MONT), gi = fr_from_limbs<Fr>(P::GEN_INV_MONT);
  Fr minv = Fr::one(), half_f = fr_from_limbs<Fr>(P::TWO_INV_MONT);
  for (unsigned i = 0; i < log_m; i++) minv = Fr::mul(minv, half_f);
  hipStream_t st = ctx->stream;
  auto launch = [&](Fr* dst, Fr base, uint64_t step, Fr scale, uint32_t count) {
    hipLaunchKernelGGL((k_pow_table<Fr>), dim3((count + 255) / 256), dim3(256), 0, st, dst, base, step, scale, count);
  };
  if (fill) {
    launch(tw, w, 1, Fr::one(), (uint32_t)half);
    launch(twi, wi, 1, Fr::one(), (uint32_t)half);
    launch(g_hi, g, N2, Fr::one(), N1);
    launch(g_lo, g, 1, Fr::one(), N2);
    launch(gi_lo, gi, 1, minv, N1);
    launch(gi_hi, gi, N1, Fr::one(), N2);
    // dst[i] = src[i * stride] in the reduced-radix form
    auto to29 = [&](const Fr* src, uint32_t stride, uint32_t* dst, uint32_t count) {
      if (count) hipLaunchKernelGGL((k_to_radix29<Fr>), dim3((count + 255) / 256), dim3(256), 0, st, src, stride, dst, count);
    };
    to29(tw, N2, twa29, N1 / 2);  // (w^N2)^j: the N1-point transforms of pass A
    to29(twi, N2, twai29, N1 / 2);
    to29(tw, N1, twb29, N2 / 2);  // (w^N1)^j: the N2-point transforms of pass B
    to29(twi, N1, twbi29, N2 / 2);
    to29(g_hi, 1, g_hi29, N1);
    ZG_HIP(hipGetLastError());
  }
  D->T.log_m = log_m;
  D->T.tw = tw;
  D->T.tw_inv = twi;
  D->T.g_hi = g_hi;
  D->T.g_lo = g_lo;
  D->T.gi_lo = gi_lo;
  D->T.gi_hi = gi_hi;
  D->T.twa29 = twa29;
  D->T.twa_inv29 = twai29;
  D->T.twb29 = twb29;
  D->T.twb_inv29 = twbi29;
  D->T.g_hi29 = g_hi29;
  D->T.m_inv = minv;
  D->T.twf = D->T.twf_inv = D->T.in_tab = D->T.out_tab = nullptr;
  D->T.qap_c = Fr::zero();
  D->T.log_n1 = l1;
  D->T.log_n2 = l2;
  D->TC = ntt_pick_tile(N1, N2);
  D->TR = ntt_pick_tile(N2, N1);
  return ZG_OK;
}

template <class Fr>
int ntt_get_domain(zg_ctx* ctx, int curve, unsigned log_m, NttDomain<Fr>** out) {
  using P = typename Fr::Params;
  unsigned l1 = 0, l2 = 0;
  if (log_m > (unsigned)P::TWO_ADICITY || log_m > 26 || !ntt_split(log_m, &l1, &l2)) return ZG_ERR_UNSUPPORTED;
  const uint32_t key = ((uint32_t)curve << 8) | log_m;
  auto it = ctx->ntt_domains.find(key);
  if (it != ctx->ntt_domains.end()) {
    *out = static_cast<NttDomain<Fr>*>(it->second);
    return ZG_OK;
  }
  auto* D = new NttDomain<Fr>();
  hipError_t e = hipMalloc((void**)&D->storage, ntt_table_elems(log_m) * sizeof(Fr));
  if (e != hipSuccess) {
    delete D;
    return e == hipErrorOutOfMemory ? ZG_ERR_OUT_OF_MEMORY : ZG_ERR_HIP;
  }
  int s = ntt_fill_tables<Fr>(ctx, log_m, D->storage, D);
  if (s == ZG_OK) {
    const size_t m = (size_t)1 << log_m;
    if (hipMalloc((void**)&D->full, 5 * m * sizeof(Fr)) != hipSuccess) {
      (void)hipGetLastError();  // no room: the two-level tables serve
      D->full = nullptr;
    } else {
      using P = typename Fr::Params;
      Fr* twf = D->full;
      Fr* twfi = twf + m;
      Fr* in_tab = twfi + m;
      Fr* out_ic = in_tab + m;
      Fr* out_fu = out_ic + m;
      hipStream_t st = ctx->stream;
      const unsigned blocks = (unsigned)((m + 255) / 256);
      hipLaunchKernelGGL((k_twf_fill<Fr>), dim3(blocks), dim3(256), 0, st, D->T.tw, log_m, D->T.log_n2, twf);
      hipLaunchKernelGGL((k_twf_fill<Fr>), dim3(blocks), dim3(256), 0, st, D->T.tw_inv, log_m, D->T.log_n2, twfi);
      const Fr g = fr_from_limbs<Fr>(P::GEN_MONT), gi = fr_from_limbs<Fr>(P::GEN_INV_MONT);
      Fr thirty_two = Fr::zero();
      thirty_two.v[0] = 32;
      hipLaunchKernelGGL((k_pow_table<Fr>), dim3(blocks), dim3(256), 0, st, in_tab, g, (uint64_t)1, Fr::to_mont(thirty_two), (uint32_t)m);
      hipLaunchKernelGGL((k_pow_table<Fr>), dim3(blocks), dim3(256), 0, st, out_ic, gi, (uint64_t)1, D->T.m_inv, (uint32_t)m);
      hipLaunchKernelGGL((k_pow_table<Fr>), dim3(blocks), dim3(256), 0, st, out_fu, g, (uint64_t)1, D->T.m_inv, (uint32_t)m);
      if (hipGetLastError() != hipSuccess) s = ZG_ERR_HIP;
      D->T.twf = twf;
      D->T.twf_inv = twfi;
      D->T.in_tab = in_tab;
      D->out_icoset = out_ic;
      D->out_fuse = out_fu;
      // 1024 / Z(g), Z(g) = g^m - 1 (divide_by_Z_on_coset), for the QAP_IN load
      Fr k1024 = Fr::zero();
      k1024.v[0] = 1024;
      D->T.qap_c = Fr::mul(Fr::to_mont(k1024), Fr::inv(Fr::sub(host_pow_u64(g, (uint64_t)m), Fr::one())));
    }
  }
  if (s == ZG_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = ZG_ERR_HIP;
  if (s != ZG_OK) {
    delete D;
    return s;
  }
  ctx->ntt_domains[key] = D;
  *out = D;
  return ZG_OK;
}

inline unsigned ilog2_u32(uint32_t x) {
  unsigned l = 0;
  while ((1u << (l + 1)) <= x && l < 31) l++;
  return l;
}

template <class Fr, int RMAX>
int ntt_launch_r(zg_ctx* ctx, NttDomain<Fr>* D, Fr* data, Fr* out, int inverse, int coset, unsigned batch, size_t vstride, int flags) {
  static std::once_flag attr_once;  // clones of a key prove from several threads (zg_pk_clone)
  hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [&] {
    const int lim = NTT_LDS_SLOTS * 36;
    const void* kernels[] = {(const void*)k_ntt_pass_a<Fr, 0, RMAX>, (const void*)k_ntt_pass_a<Fr, 1, RMAX>,
                             (const void*)k_ntt_pass_a<Fr, 2, RMAX>,
                             (const void*)k_ntt_pass_b<Fr, 0, RMAX>, (const void*)k_ntt_pass_b<Fr, 1, RMAX>,
                             (const void*)k_ntt_pass_b<Fr, 2, RMAX>, (const void*)k_ntt_pass_b<Fr, 3, RMAX>};
    for (const void* k : kernels)
      if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  });
  ZG_HIP(attr_err);
  const bool fuse_next = (flags & NTT_FUSE_COSET_NEXT) != 0, qap_in = (flags & NTT_QAP_IN) != 0;
  if ((fuse_next && (!inverse || coset)) || (qap_in && (!inverse || !coset || batch != 1)) || ((fuse_next || qap_in) && !D->full))
    return ZG_ERR_INVALID_ARGUMENT;
  const uint32_t N1 = 1u << D->T.log_n1, N2 = 1u << D->T.log_n2;
  const uint32_t TC = D->TC, TR = D->TR;
  const unsigned log_tc = ilog2_u32(TC), log_tr = ilog2_u32(TR);
  hipStream_t st = ctx->stream;
  auto lds_bytes = [](uint32_t elems) { return (size_t)(elems + elems / 64 + 1) * 36; };
  const size_t lds_a = lds_bytes(N1 * TC), lds_b = lds_bytes(N2 * TR);
  auto threads = [](uint32_t elems) {
    uint32_t t = elems >> RMAX;  // lds_fft: 2^RMAX elements per work item
    if (t < 64) t = 64;
    if (t > (uint32_t)ntt_max_threads(RMAX)) t = ntt_max_threads(RMAX);
    return t;
  };
  NttTables<Fr> T = D->T;
  ctx->mark("ntt_pass_a");
  const dim3 ga(N2 / TC, batch), ta(threads(N1 * TC));
  if (qap_in)
    hipLaunchKernelGGL((k_ntt_pass_a<Fr, 2, RMAX>), ga, ta, lds_a, st, data, T, inverse, log_tc, vstride);
  else if (coset && !inverse)
    hipLaunchKernelGGL((k_ntt_pass_a<Fr, 1, RMAX>), ga, ta, lds_a, st, data, T, inverse, log_tc, vstride);
  else
    hipLaunchKernelGGL((k_ntt_pass_a<Fr, 0, RMAX>), ga, ta, lds_a, st, data, T, inverse, log_tc, vstride);
  ctx->mark("ntt_pass_b");
  const dim3 gb(N1 / TR, batch), tb(threads(N2 * TR));
  if (!inverse) {
    hipLaunchKernelGGL((k_ntt_pass_b<Fr, 0, RMAX>), gb, tb, lds_b, st, data, out, T, inverse, log_tr, vstride);
  } else if (!coset && !fuse_next) {
    hipLaunchKernelGGL((k_ntt_pass_b<Fr, 1, RMAX>), gb, tb, lds_b, st, data, out, T, inverse, log_tr, vstride);
  } else if (D->full) {
    T.out_tab = coset ? D->out_icoset : D->out_fuse;
    hipLaunchKernelGGL((k_ntt_pass_b<Fr, 3, RMAX>), gb, tb, lds_b, st, data, out, T, inverse, log_tr, vstride);
  } else {
    hipLaunchKernelGGL((k_ntt_pass_b<Fr, 2, RMAX>), gb, tb, lds_b, st, data, out, T, inverse, log_tr, vstride);
  }
  ctx->mark("ntt_end");
  ZG_HIP(hipGetLastError());
  return ZG_OK;
}

// Stages per register round for `batch` vectors of m elements.  Measured on MI355X (profiles/r2_ntt_radix.txt): the
// transforms are bound by the field multiplier, not by LDS round trips, so the radix moves the time by a few percent
// only -- radix 4 is best from 2^18 elements up, radix 2 below (a radix-4 work item leaves a small transform with too
// few lanes), radix 8 never (150 VGPRs, half the waves).  ZG_NTT_RADIX_LOG = 1 | 2 overrides (tuning knob).
inline int ntt_pick_rmax(size_t total_elems) {
  static const int forced = [] {
    const char* e = zg_knob("ZG_NTT_RADIX_LOG");
    return e ? atoi(e) : 0;
  }();
  if (forced >= 1 && forced <= 2) return forced;
  return total_elems >= ((size_t)1 << 18) ? 2 : 1;
}

// data (`batch` vectors of m elements, `vstride` elements apart, Montgomery form) is overwritten with pass A's
// intermediate; the transforms land in `out` (same layout).
template <class Fr>
int ntt_launch(zg_ctx* ctx, NttDomain<Fr>* D, Fr* data, Fr* out, int inverse, int coset, unsigned batch = 1, size_t vstride = 0,
               int flags = 0) {
  const int r = ntt_pick_rmax(((size_t)batch) << D->T.log_m);
  if (r == 2) return ntt_launch_r<Fr, 2>(ctx, D, data, out, inverse, coset, batch, vstride, flags);
  return ntt_launch_r<Fr, 1>(ctx, D, data, out, inverse, coset, batch, vstride, flags);
}

// C-ABI level: in-place semantics on `data` (device pointer, Montgomery or plain per `form`)
template <class Fr>
int ntt_inplace(zg_ctx* ctx, int curve, void* d_data, unsigned log_m, int inverse, int coset, int form) {
  NttDomain<Fr>* D = nullptr;
  ZG_TRY(ntt_get_domain<Fr>(ctx, curve, log_m, &D));
  const size_t m = (size_t)1 << log_m;
  ZG_TRY(ctx->ws_reserve(zg_ctx::ws_round(m * sizeof(Fr))));
  Fr* tmp = ctx->ws_take<Fr>(m);
  if (!tmp) return ZG_ERR_OUT_OF_MEMORY;
  Fr* data = reinterpret_cast<Fr*>(d_data);
  hipStream_t st = ctx->stream;
  const unsigned blocks = (unsigned)((m + 255) / 256);
  ctx->prof_begin();
  if (form == ZG_FORM_PLAIN) hipLaunchKernelGGL((k_fr_convert<Fr>), dim3(blocks), dim3(256), 0, st, data, m, 1);
  ZG_TRY(ntt_launch<Fr>(ctx, D, data, tmp, inverse, coset));
  ZG_HIP(hipMemcpyAsync(data, tmp, m * sizeof(Fr), hipMemcpyDeviceToDevice, st));
  if (form == ZG_FORM_PLAIN) hipLaunchKernelGGL((k_fr_convert<Fr>), dim3(blocks), dim3(256), 0, st, data, m, 0);
  ZG_HIP(hipGetLastError());
  ZG_HIP(zg_stream_wait(st));
  ctx->prof_end();
  return ZG_OK;
}

// The allocation-free form behind zg_ntt_noalloc: the tables of one (curve, log m) and the transform buffer live in
// `scratch` (the library's static device array, api.hip) instead of hipMalloc'ed memory.  *tag_curve / *tag_log_m say
// which domain's tables the scratch currently holds; they are rebuilt (kernels only) when another size is asked for.
static constexpr unsigned NTT_STATIC_MAX_LOG = 16;
inline size_t ntt_static_table_elems() {
  size_t mx = 0;
  for (unsigned l = 0; l <= NTT_STATIC_MAX_LOG; l++) mx = ntt_table_elems(l) > mx ? ntt_table_elems(l) : mx;
  return mx;
}
inline size_t ntt_static_bytes() { return (ntt_static_table_elems() + ((size_t)1 << NTT_STATIC_MAX_LOG)) * 32; }

template <class Fr>
int ntt_static(zg_ctx* ctx, int curve, void* d_data, unsigned log_m, int inverse, int coset, int form, void* scratch,
               size_t scratch_bytes, int* tag_curve, unsigned* tag_log_m) {
  if (log_m > NTT_STATIC_MAX_LOG) return ZG_ERR_UNSUPPORTED;
  if (scratch_bytes < ntt_static_bytes()) return ZG_ERR_INVALID_ARGUMENT;
  Fr* tables = reinterpret_cast<Fr*>(scratch);
  Fr* tmp = tables + ntt_static_table_elems();
  NttDomain<Fr> D;  // storage stays null: nothing to free
  const bool cached = *tag_curve == curve && *tag_log_m == log_m;
  *tag_curve = -1;  // nothing valid until the fill kernels are KNOWN to have run (set again after the synchronise below)
  ZG_TRY(ntt_fill_tables<Fr>(ctx, log_m, tables, &D, !cached));
  const size_t m = (size_t)1 << log_m;
  Fr* data = reinterpret_cast<Fr*>(d_data);
  hipStream_t st = ctx->stream;
  const unsigned blocks = (unsigned)((m + 255) / 256);
  const bool prof = ctx->profiling;
  ctx->profiling = false;  // phase marks create events on demand: not in this entry point
  if (form == ZG_FORM_PLAIN) hipLaunchKernelGGL((k_fr_convert<Fr>), dim3(blocks), dim3(256), 0, st, data, m, 1);
  int s = ntt_launch<Fr>(ctx, &D, data, tmp, inverse, coset);
  ctx->profiling = prof;
  if (s == ZG_OK && hipMemcpyAsync(data, tmp, m * sizeof(Fr), hipMemcpyDeviceToDevice, st) != hipSuccess) s = ZG_ERR_HIP;
  if (s == ZG_OK && form == ZG_FORM_PLAIN) hipLaunchKernelGGL((k_fr_convert<Fr>), dim3(blocks), dim3(256), 0, st, data, m, 0);
  if (s == ZG_OK && hipGetLastError() != hipSuccess) s = ZG_ERR_HIP;
  // whatever was queued is waited for even after an error: the scratch is shared and the next call may overwrite it
  if (hipStreamSynchronize(st) != hipSuccess && s == ZG_OK) s = ZG_ERR_HIP;
  ZG_TRY(s);
  *tag_curve = curve;
  *tag_log_m = log_m;
  return ZG_OK;
}

}  // namespace zg
