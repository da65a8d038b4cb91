// Host driver for the Fr evaluation-domain transforms: per-(curve, log m) twiddle tables cached in the context
// (the analogue of constructing a libfqfft::basic_radix2_domain<Fr>(m) once), launch geometry, fused scalings.
#pragma once
#include "common.cuh"
#include "ntt.cuh"

namespace zg {

template <class Fr>
struct NttDomain : NttDomainBase {
  NttTables<Fr> T;
  Fr* storage = nullptr;
  uint32_t TC = 1, TR = 1;
  ~NttDomain() override {
    if (storage) hipFree(storage);
  }
};

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

inline uint32_t ntt_pick_tile(uint32_t n_this, uint32_t n_other) {
  uint32_t t = NTT_LDS_ELEMS / n_this;
  if (t > n_other) t = n_other;
  if (t < 1) t = 1;
  // keep at least ~512 workgroups in flight when the transform is large enough to allow it
  while (t > 1 && n_other / t < 512) t >>= 1;
  return t;
}

template <class Fr>
int ntt_get_domain(zg_ctx* ctx, int curve, unsigned log_m, NttDomain<Fr>** out) {
  using P = typename Fr::Params;
  if (log_m > (unsigned)P::TWO_ADICITY || log_m > 26) return ZG_ERR_UNSUPPORTED;
  const uint32_t key = ((uint32_t)curve << 8) | log_m;
  auto it = ctx->ntt_domains.find(key);
  if (it != ctx->ntt_domains.end()) {
    *out = static_cast<NttDomain<Fr>*>(it->second);
    return ZG_OK;
  }
  auto* D = new NttDomain<Fr>();
  const uint64_t m = 1ull << log_m;
  const unsigned l1 = log_m <= 12 ? log_m : (log_m + 1) / 2, l2 = log_m - l1;
  if (l1 > 12) {
    delete D;
    return ZG_ERR_UNSUPPORTED;
  }
  const uint32_t N1 = 1u << l1, N2 = 1u << l2;
  const size_t half = m > 1 ? m / 2 : 1;
  const size_t total = 2 * half + 2 * (size_t)N1 + 2 * (size_t)N2;
  hipError_t e = hipMalloc((void**)&D->storage, total * sizeof(Fr));
  if (e != hipSuccess) {
    delete D;
    return e == hipErrorOutOfMemory ? ZG_ERR_OUT_OF_MEMORY : ZG_ERR_HIP;
  }
  Fr* p = D->storage;
  Fr* tw = p; p += half;
  Fr* twi = p; p += half;
  Fr* g_hi = p; p += N1;
  Fr* g_lo = p; p += N2;
  Fr* gi_lo = p; p += N1;
  Fr* gi_hi = p; p += N2;
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
  launch(tw, w, 1, Fr::one(), (uint32_t)half);
  launch(twi, wi, 1, Fr::one(), (uint32_t)half);
  launch(g_hi, g, N2, Fr::one(), N1);
  launch(g_lo, g, 1, Fr::one(), N2);
  launch(gi_lo, gi, 1, minv, N1);
  launch(gi_hi, gi, N1, Fr::one(), N2);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    delete D;
    return ZG_ERR_HIP;
  }
  D->T.log_m = log_m;
  D->T.tw = tw;
  D->T.tw_inv = twi;
  D->T.g_hi = g_hi;
  D->T.g_lo = g_lo;
  D->T.gi_lo = gi_lo;
  D->T.gi_hi = gi_hi;
  D->T.m_inv = minv;
  D->T.log_n1 = l1;
  D->T.log_n2 = l2;
  D->TC = ntt_pick_tile(N1, N2);
  D->TR = ntt_pick_tile(N2, N1);
  ctx->ntt_domains[key] = D;
  *out = D;
  return ZG_OK;
}

// data (m elements, Montgomery form) is overwritten with pass A's intermediate; the transform lands in `out`.
template <class Fr>
int ntt_launch(zg_ctx* ctx, NttDomain<Fr>* D, Fr* data, Fr* out, int inverse, int coset) {
  static bool attr_done = false;
  if (!attr_done) {
    const int lim = NTT_LDS_ELEMS * 32;
    ZG_HIP(hipFuncSetAttribute((const void*)k_ntt_pass_a<Fr, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    ZG_HIP(hipFuncSetAttribute((const void*)k_ntt_pass_a<Fr, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    ZG_HIP(hipFuncSetAttribute((const void*)k_ntt_pass_b<Fr, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    ZG_HIP(hipFuncSetAttribute((const void*)k_ntt_pass_b<Fr, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    ZG_HIP(hipFuncSetAttribute((const void*)k_ntt_pass_b<Fr, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    attr_done = true;
  }
  const uint32_t N1 = 1u << D->T.log_n1, N2 = 1u << D->T.log_n2;
  const uint32_t TC = D->TC, TR = D->TR;
  hipStream_t st = ctx->stream;
  const size_t lds_a = (size_t)N1 * TC * 32, lds_b = (size_t)N2 * TR * 32;
  auto threads = [](uint32_t elems) {
    uint32_t t = elems / 2;
    if (t < 64) t = 64;
    if (t > (uint32_t)NTT_THREADS) t = NTT_THREADS;
    return t;
  };
  ctx->mark("ntt_pass_a");
  if (coset && !inverse)
    hipLaunchKernelGGL((k_ntt_pass_a<Fr, true>), dim3(N2 / TC), dim3(threads(N1 * TC)), lds_a, st, data, D->T, inverse, TC);
  else
    hipLaunchKernelGGL((k_ntt_pass_a<Fr, false>), dim3(N2 / TC), dim3(threads(N1 * TC)), lds_a, st, data, D->T, inverse, TC);
  ctx->mark("ntt_pass_b");
  const dim3 gb(N1 / TR), tb(threads(N2 * TR));
  if (!inverse)
    hipLaunchKernelGGL((k_ntt_pass_b<Fr, 0>), gb, tb, lds_b, st, data, out, D->T, inverse, TR);
  else if (!coset)
    hipLaunchKernelGGL((k_ntt_pass_b<Fr, 1>), gb, tb, lds_b, st, data, out, D->T, inverse, TR);
  else
    hipLaunchKernelGGL((k_ntt_pass_b<Fr, 2>), gb, tb, lds_b, st, data, out, D->T, inverse, TR);
  ctx->mark("ntt_end");
  ZG_HIP(hipGetLastError());
  return ZG_OK;
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
  ZG_HIP(hipStreamSynchronize(st));
  ctx->prof_end();
  return ZG_OK;
}

}  // namespace zg
