// C ABI (include/zeth_gpu.h): argument checking, contexts, resident bases, MSM entry points.
#include <dlfcn.h>
#include <link.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "common.cuh"

// ---- static scratch of the allocation-free entry point (zg_ntt_noalloc) --------------------------------------
// One array per GPU the code object is loaded on, held until the process ends: the twiddle/coset tables of one
// (curve, log n <= 16) domain followed by the 2^16-element transform buffer (ntt_host.cuh, ntt_static_bytes()).
// `scratch` is a host-side stand-in for the symbol, never a device address: every use below goes through the pointer
// hipGetSymbolAddress returned for the device in question.
static constexpr size_t ZG_STATIC_SCRATCH_BYTES =
    (((size_t)1 << 16) + 4 * 256 + ((3 * 256 * 36 + 31) / 32) + ((size_t)1 << 16)) * 32;  // 4,254,720
static __device__ __align__(256) unsigned char scratch[ZG_STATIC_SCRATCH_BYTES];

static constexpr int ZG_MAX_DEVICES = 64;
struct StaticScratch {
  void* ptr = nullptr;  // this device's copy of `scratch`
  int tag_curve = -1;   // domain whose tables the scratch holds (-1: none)
  unsigned tag_log_m = 0;
};
static StaticScratch g_static[ZG_MAX_DEVICES];
static std::mutex g_static_mu;                       // guards the resolution of `ptr`
static std::mutex g_static_call_mu[ZG_MAX_DEVICES];  // one zg_ntt_noalloc at a time per GPU: the scratch and its tags are shared

// with `device` current: resolve the device address of `scratch` once per device
static int static_scratch_resolve(int device) {
  if (device < 0 || device >= ZG_MAX_DEVICES) return ZG_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(g_static_mu);
  if (g_static[device].ptr) return ZG_OK;
  void* p = nullptr;
  ZG_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  if (!p) return ZG_ERR_HIP;
  g_static[device].ptr = p;
  return ZG_OK;
}

const GroupOps *ZG_OPS_BN254_G1(), *ZG_OPS_BN254_G2(), *ZG_OPS_BLS377_G1(), *ZG_OPS_BLS377_G2();

const GroupOps* zg_group_ops(int curve, int group) {
  if (curve == ZG_CURVE_ALT_BN128) return group == ZG_GROUP_G1 ? ZG_OPS_BN254_G1() : group == ZG_GROUP_G2 ? ZG_OPS_BN254_G2() : nullptr;
  if (curve == ZG_CURVE_BLS12_377) return group == ZG_GROUP_G1 ? ZG_OPS_BLS377_G1() : group == ZG_GROUP_G2 ? ZG_OPS_BLS377_G2() : nullptr;
  return nullptr;
}

extern "C" {

const char* zg_strerror(int s) {
  switch (s) {
    case ZG_OK: return "ok";
    case ZG_ERR_INVALID_ARGUMENT: return "invalid argument";
    case ZG_ERR_LENGTH: return "length error";
    case ZG_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case ZG_ERR_HIP: return "HIP runtime error";
    case ZG_ERR_OUT_OF_MEMORY: return "out of device memory";
    case ZG_ERR_UNSUPPORTED: return "unsupported size or configuration";
    default: return "unknown status";
  }
}

int zg_sizes(int curve, size_t* fr, size_t* fq, size_t* g1, size_t* g2) {
  const GroupOps* o1 = zg_group_ops(curve, ZG_GROUP_G1);
  const GroupOps* o2 = zg_group_ops(curve, ZG_GROUP_G2);
  if (!o1 || !o2) return ZG_ERR_INVALID_ARGUMENT;
  if (fr) *fr = 32;
  if (fq) *fq = o1->fq_limbs * 4;
  if (g1) *g1 = o1->affine_bytes;
  if (g2) *g2 = o2->affine_bytes;
  return ZG_OK;
}

int zg_ctx_create(int device, zg_ctx** out) {
  // (debug builds: ZG_CTX_PRIO = 1 | 0 | -1, for tools/diag_prio.py)
  static const int prio = zg_knob("ZG_CTX_PRIO") ? atoi(zg_knob("ZG_CTX_PRIO")) : +1;
  return zg_ctx_create_prio(device, prio, out);
}


}  // extern "C"

// prio > 0: highest stream priority, prio == 0: the default priority, prio < 0: lowest.  Internal: every caller in the
// shipped library passes +1.  Contexts of DIFFERENT priorities whose streams are fed by several threads at once stall the
// runtime's queues (profiles/r3_stream_priority_bisect.jsonl); the other values exist for the debug build's
// knobs (tools/diag_prio.py), which is how that was bisected.
int zg_ctx_create_prio(int device, int prio, zg_ctx** out) {
  if (!out) return ZG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZG_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(device));
  ZG_TRY(static_scratch_resolve(device));
  zg_ctx* c = new (std::nothrow) zg_ctx();
  if (!c) return ZG_ERR_OUT_OF_MEMORY;
  c->device = device;
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
  hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio > 0 ? greatest : (prio == 0 ? (least + greatest) / 2 : least));
  if (e != hipSuccess) {
    delete c;
    return ZG_ERR_HIP;
  }
  *out = c;
  return ZG_OK;
}

extern "C" {

void zg_ctx_destroy(zg_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->side_stream) {
    hipStreamSynchronize(c->side_stream);
    hipStreamDestroy(c->side_stream);
    hipEventDestroy(c->ev_fork);
    hipEventDestroy(c->ev_join);
  }
  for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
  for (auto& kv : c->ntt_domains) delete kv.second;
  if (c->ws) hipFree(c->ws);
  if (c->pinned) hipHostFree(c->pinned);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int zg_ctx_synchronize(zg_ctx* c) {
  if (!c) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(c->device));
  ZG_HIP(hipStreamSynchronize(c->stream));
  return ZG_OK;
}

int zg_ctx_set_profiling(zg_ctx* c, int enabled) {
  if (!c) return ZG_ERR_INVALID_ARGUMENT;
  c->profiling = enabled != 0;
  return ZG_OK;
}

int zg_ctx_last_timings(zg_ctx* c, const char** names, float* ms, int max_entries, int* n_entries) {
  if (!c || !n_entries) return ZG_ERR_INVALID_ARGUMENT;
  int n = (int)c->last_ms.size();
  if (n > max_entries) n = max_entries;
  for (int i = 0; i < n; i++) {
    if (names) names[i] = c->last_names[i];
    if (ms) ms[i] = c->last_ms[i];
  }
  *n_entries = n;
  return ZG_OK;
}

int zg_dev_malloc(zg_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  ZG_HIP(hipMalloc(dptr, bytes ? bytes : 1));
  return ZG_OK;
}
int zg_dev_free(zg_ctx* ctx, void* dptr) {
  if (!ctx) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  if (dptr) ZG_HIP(hipFree(dptr));
  return ZG_OK;
}
int zg_dev_upload(zg_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  ZG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  return ZG_OK;
}
int zg_dev_download(zg_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  ZG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  return ZG_OK;
}

__device__ __forceinline__ uint64_t zg_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
struct ZgModulus {
  uint64_t p[4];
  uint64_t top_mask;
};
static __global__ void __launch_bounds__(256) k_scalars_random(uint64_t* __restrict__ out, size_t n, uint64_t seed,
                                                                ZgModulus m) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (uint64_t t = 0; t < 256; t++) {
    uint64_t ctr = ((seed + i) << 8) + t;
    uint64_t w[4];
    for (int k = 0; k < 4; k++) w[k] = zg_splitmix64(ctr * 4 + k);
    w[3] &= m.top_mask;
    bool lt = false;
    for (int k = 3; k >= 0; k--) {
      if (w[k] != m.p[k]) {
        lt = w[k] < m.p[k];
        break;
      }
    }
    if (lt || t == 255) {
      if (!lt) w[3] = 0;  // unreachable in practice (probability 2^-256)
      for (int k = 0; k < 4; k++) out[4 * i + k] = w[k];
      return;
    }
  }
}
int zg_scalars_random(zg_ctx* ctx, int curve, void* dptr, size_t n, uint64_t seed) {
  if (!ctx || (n && !dptr)) return ZG_ERR_INVALID_ARGUMENT;
  ZgModulus m;
  if (curve == ZG_CURVE_ALT_BN128) {
    m = ZgModulus{{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
                  0x3fffffffffffffffull};
  } else if (curve == ZG_CURVE_BLS12_377) {
    m = ZgModulus{{0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull},
                  0x1fffffffffffffffull};
  } else {
    return ZG_ERR_INVALID_ARGUMENT;
  }
  if (!n) return ZG_OK;
  ZG_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_scalars_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint64_t*)dptr, n,
                     seed, m);
  ZG_HIP(hipGetLastError());
  ZG_HIP(hipStreamSynchronize(ctx->stream));
  return ZG_OK;
}

static int bases_alloc(zg_ctx* ctx, int curve, int group, size_t n, zg_bases** out) {
  if (!ctx || !out) return ZG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  const GroupOps* ops = zg_group_ops(curve, group);
  if (!ops) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  zg_bases* b = new (std::nothrow) zg_bases();
  if (!b) return ZG_ERR_OUT_OF_MEMORY;
  b->ctx = ctx;
  b->curve = curve;
  b->group = group;
  b->n = n;
  b->ops = ops;
  b->d_pts = nullptr;
  if (n) {
    hipError_t e = hipMalloc(&b->d_pts, n * ops->affine_bytes);
    if (e != hipSuccess) {
      delete b;
      return e == hipErrorOutOfMemory ? ZG_ERR_OUT_OF_MEMORY : ZG_ERR_HIP;
    }
  }
  *out = b;
  return ZG_OK;
}

int zg_bases_create(zg_ctx* ctx, int curve, int group, const void* points, size_t n, int form, int mem,
                    zg_bases** out) {
  if ((n && !points) || (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT) || (mem != ZG_MEM_HOST && mem != ZG_MEM_DEVICE))
    return ZG_ERR_INVALID_ARGUMENT;
  zg_bases* b = nullptr;
  ZG_TRY(bases_alloc(ctx, curve, group, n, &b));
  int s = ZG_OK;
  if (n) {
    hipError_t e = hipMemcpyAsync(b->d_pts, points, n * b->ops->affine_bytes,
                                  mem == ZG_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) s = ZG_ERR_HIP;
    if (s == ZG_OK && form == ZG_FORM_PLAIN) s = b->ops->to_mont(ctx, b->d_pts, n);
    if (s == ZG_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = ZG_ERR_HIP;
  }
  if (s != ZG_OK) {
    zg_bases_free(b);
    return s;
  }
  *out = b;
  return ZG_OK;
}

int zg_bases_generate(zg_ctx* ctx, int curve, int group, size_t n, const void* t0, const void* d, zg_bases** out) {
  if (!t0 || !d) return ZG_ERR_INVALID_ARGUMENT;
  zg_bases* b = nullptr;
  ZG_TRY(bases_alloc(ctx, curve, group, n, &b));
  int s = b->ops->generate(ctx, b->d_pts, n, (const uint32_t*)t0, (const uint32_t*)d);
  if (s != ZG_OK) {
    zg_bases_free(b);
    return s;
  }
  *out = b;
  return ZG_OK;
}

int zg_bases_batch_exp(zg_ctx* ctx, int curve, int group, const void* scalars, size_t n, zg_bases** out) {
  if (n && !scalars) return ZG_ERR_INVALID_ARGUMENT;
  zg_bases* b = nullptr;
  ZG_TRY(bases_alloc(ctx, curve, group, n, &b));
  void* ds = nullptr;
  int s = ZG_OK;
  if (n) {
    if (hipMalloc(&ds, n * 32) != hipSuccess) s = ZG_ERR_OUT_OF_MEMORY;
    if (s == ZG_OK && hipMemcpyAsync(ds, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) s = ZG_ERR_HIP;
    if (s == ZG_OK) s = b->ops->batch_exp(ctx, b->d_pts, n, (const uint32_t*)ds);
    if (ds) hipFree(ds);
  }
  if (s != ZG_OK) {
    zg_bases_free(b);
    return s;
  }
  *out = b;
  return ZG_OK;
}

int zg_bases_count(const zg_bases* b, size_t* n) {
  if (!b || !n) return ZG_ERR_INVALID_ARGUMENT;
  *n = b->n;
  return ZG_OK;
}

int zg_bases_read(zg_bases* b, size_t first, size_t count, int form, void* out) {
  if (!b || (count && !out) || first > b->n || count > b->n - first) return ZG_ERR_INVALID_ARGUMENT;
  if (!count) return ZG_OK;
  ZG_HIP(hipSetDevice(b->ctx->device));
  ZG_HIP(hipMemcpy(out, (const char*)b->d_pts + first * b->ops->affine_bytes, count * b->ops->affine_bytes,
                   hipMemcpyDeviceToHost));
  if (form == ZG_FORM_PLAIN) b->ops->host_convert(out, count, 0);
  return ZG_OK;
}

int zg_bases_precompute(zg_bases* b, int window_bits) {
  if (!b) return ZG_ERR_INVALID_ARGUMENT;
  int c = 0, D = 0;
  ZG_TRY(b->ops->table_params(b->n, window_bits, &c, &D));
  if (b->table_D) return b->table_c == c || window_bits == 0 ? ZG_OK : ZG_ERR_INVALID_ARGUMENT;
  if (b->n == 0) return ZG_OK;
  if ((uint64_t)D * (uint64_t)b->n >= (1ull << 31)) return ZG_ERR_UNSUPPORTED;  // 31-bit table indices in the sort's entries
  zg_ctx* ctx = b->ctx;
  ZG_HIP(hipSetDevice(ctx->device));
  void* table = nullptr;
  const size_t level = b->n * b->ops->affine_bytes;
  if (hipMalloc(&table, level * (size_t)D) != hipSuccess) {
    (void)hipGetLastError();
    return ZG_ERR_OUT_OF_MEMORY;
  }
  int s = ZG_OK;
  if (hipMemcpyAsync(table, b->d_pts, level, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) s = ZG_ERR_HIP;
  if (s == ZG_OK) s = b->ops->table_build(ctx, table, b->n, c, D);
  if (s != ZG_OK) {
    hipStreamSynchronize(ctx->stream);
    hipFree(table);
    return s;
  }
  hipFree(b->d_pts);
  b->d_pts = table;
  b->table_c = c;
  b->table_D = D;
  return ZG_OK;
}

int zg_bases_table_info(const zg_bases* b, int* window_bits, int* levels) {
  if (!b || !window_bits || !levels) return ZG_ERR_INVALID_ARGUMENT;
  *window_bits = b->table_c;
  *levels = b->table_D;
  return ZG_OK;
}

void zg_bases_free(zg_bases* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  if (b->d_pts) hipFree(b->d_pts);
  delete b;
}

int zg_msm_ex(zg_bases* b, size_t first, const void* scalars, size_t n, int scalar_form, int scalars_mem,
              int out_form, void* out_affine, int window_bits) {
  if (!b || !out_affine || (n && !scalars)) return ZG_ERR_INVALID_ARGUMENT;
  if (first > b->n || n > b->n - first) return ZG_ERR_INVALID_ARGUMENT;  // multi_exp.tcc:28-31
  if ((scalar_form != ZG_FORM_PLAIN && scalar_form != ZG_FORM_MONT) || (out_form != ZG_FORM_PLAIN && out_form != ZG_FORM_MONT))
    return ZG_ERR_INVALID_ARGUMENT;
  if (window_bits != 0 && (window_bits < 2 || window_bits > 16)) return ZG_ERR_INVALID_ARGUMENT;
  zg_ctx* ctx = b->ctx;
  ZG_HIP(hipSetDevice(ctx->device));
  const uint32_t* d_scalars = (const uint32_t*)scalars;
  void* staged = nullptr;
  if (n && scalars_mem == ZG_MEM_HOST) {
    ZG_HIP(hipMalloc(&staged, n * 32));
    hipError_t e = hipMemcpyAsync(staged, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
      hipFree(staged);
      return ZG_ERR_HIP;
    }
    d_scalars = (const uint32_t*)staged;
  } else if (scalars_mem != ZG_MEM_DEVICE && scalars_mem != ZG_MEM_HOST) {
    return ZG_ERR_INVALID_ARGUMENT;
  }
  int s;
  if (b->table_D) {
    const MsmTable tab{b->n, first, b->table_c, b->table_D};
    s = b->ops->msm(ctx, b->d_pts, d_scalars, n, scalar_form == ZG_FORM_MONT, window_bits, out_form, out_affine, &tab);
  } else {
    s = b->ops->msm(ctx, (const char*)b->d_pts + first * b->ops->affine_bytes, d_scalars, n, scalar_form == ZG_FORM_MONT,
                    window_bits, out_form, out_affine, nullptr);
  }
  if (staged) {
    hipStreamSynchronize(ctx->stream);
    hipFree(staged);
  }
  return s;
}

int zg_msm(zg_bases* b, size_t first, const void* scalars, size_t n, int scalar_form, int scalars_mem, int out_form,
           void* out_affine) {
  return zg_msm_ex(b, first, scalars, n, scalar_form, scalars_mem, out_form, out_affine, 0);
}

int zg_ntt(zg_ctx* ctx, int curve, void* data, unsigned log_n, int inverse, int coset, int form, int mem) {
  const FrOps* ops = zg_fr_ops(curve);
  if (!ctx || !ops || !data || (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT) ||
      (mem != ZG_MEM_HOST && mem != ZG_MEM_DEVICE))
    return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  const size_t bytes = ((size_t)32) << log_n;
  if (log_n > 28) return ZG_ERR_UNSUPPORTED;
  if (mem == ZG_MEM_DEVICE) return ops->ntt_inplace(ctx, curve, data, log_n, inverse != 0, coset != 0, form);
  void* d = nullptr;
  ZG_HIP(hipMalloc(&d, bytes));
  int s = ZG_OK;
  if (hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) s = ZG_ERR_HIP;
  if (s == ZG_OK) s = ops->ntt_inplace(ctx, curve, d, log_n, inverse != 0, coset != 0, form);
  if (s == ZG_OK && hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) s = ZG_ERR_HIP;
  hipFree(d);
  return s;
}

int zg_ntt_noalloc(zg_ctx* ctx, int curve, void* d_data, unsigned log_n, int inverse, int coset, int form) {
  const FrOps* ops = zg_fr_ops(curve);
  if (!ctx || !ops || !d_data || (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT)) return ZG_ERR_INVALID_ARGUMENT;
  if (log_n > 16) return ZG_ERR_UNSUPPORTED;  // the static scratch holds one domain of at most 2^16 points
  if (ctx->device < 0 || ctx->device >= ZG_MAX_DEVICES) return ZG_ERR_UNSUPPORTED;
  if (ops->ntt_static_bytes() > ZG_STATIC_SCRATCH_BYTES) return ZG_ERR_UNSUPPORTED;
  ZG_HIP(hipSetDevice(ctx->device));
  // The call is synchronous, so holding the device's lock for all of it costs nothing and makes calls from several
  // contexts on one GPU (the serving-loop case) take turns on the shared scratch instead of racing on it.
  std::lock_guard<std::mutex> lock(g_static_call_mu[ctx->device]);
  StaticScratch& sc = g_static[ctx->device];  // ptr was resolved when this context was created
  if (!sc.ptr) return ZG_ERR_HIP;
  return ops->ntt_static(ctx, curve, d_data, log_n, inverse != 0, coset != 0, form, sc.ptr, ZG_STATIC_SCRATCH_BYTES,
                         &sc.tag_curve, &sc.tag_log_m);
}

int zg_pk_create(zg_ctx* ctx, int curve, const zg_pk_view* view, zg_pk** out) {
  const FrOps* ops = zg_fr_ops(curve);
  if (!ctx || !ops || !view || !out) return ZG_ERR_INVALID_ARGUMENT;
  if (view->form != ZG_FORM_PLAIN && view->form != ZG_FORM_MONT) return ZG_ERR_INVALID_ARGUMENT;
  if (!view->alpha_g1 || !view->beta_g1 || !view->beta_g2 || !view->delta_g1 || !view->delta_g2) return ZG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  ZG_HIP(hipSetDevice(ctx->device));
  PkBase* impl = nullptr;
  ZG_TRY(ops->pk_create(ctx, curve, view, &impl));
  zg_pk* pk = new (std::nothrow) zg_pk{ctx, impl, false};
  if (!pk) {
    delete impl;
    return ZG_ERR_OUT_OF_MEMORY;
  }
  *out = pk;
  return ZG_OK;
}

void zg_pk_free(zg_pk* pk) {
  if (!pk) return;
  delete pk->impl;
  if (pk->owns_ctx) zg_ctx_destroy(pk->ctx);
  delete pk;
}

int zg_pk_clone(zg_pk* pk, zg_pk** out) {
  if (!pk || !out) return ZG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  ZG_HIP(hipSetDevice(pk->ctx->device));
  zg_ctx* c = nullptr;
  ZG_TRY(zg_ctx_create(pk->ctx->device, &c));
  PkBase* impl = nullptr;
  int s = pk->impl->clone_onto(c, &impl);
  if (s != ZG_OK) {
    zg_ctx_destroy(c);
    return s;
  }
  zg_pk* q = new (std::nothrow) zg_pk{c, impl, true};
  if (!q) {
    delete impl;
    zg_ctx_destroy(c);
    return ZG_ERR_OUT_OF_MEMORY;
  }
  *out = q;
  return ZG_OK;
}

int zg_pk_domain_size(const zg_pk* pk, size_t* m) {
  if (!pk || !m) return ZG_ERR_INVALID_ARGUMENT;
  *m = pk->impl->domain_size();
  return ZG_OK;
}

int zg_prove(zg_pk* pk, const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form, const void* r,
             const void* s, int out_form, void* proof_out) {
  if (!pk || !r || !s || !proof_out || (n_primary && !primary) || (n_aux && !aux)) return ZG_ERR_INVALID_ARGUMENT;
  if ((form != ZG_FORM_PLAIN && form != ZG_FORM_MONT) || (out_form != ZG_FORM_PLAIN && out_form != ZG_FORM_MONT))
    return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(pk->ctx->device));
  return pk->impl->prove(primary, n_primary, aux, n_aux, form, (const uint32_t*)r, (const uint32_t*)s, out_form, proof_out);
}

int zg_witness_map(zg_pk* pk, const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form,
                   int out_form, void* h_out) {
  if (!pk || !h_out || (n_primary && !primary) || (n_aux && !aux)) return ZG_ERR_INVALID_ARGUMENT;
  if ((form != ZG_FORM_PLAIN && form != ZG_FORM_MONT) || (out_form != ZG_FORM_PLAIN && out_form != ZG_FORM_MONT))
    return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(pk->ctx->device));
  return pk->impl->witness_map_out(primary, n_primary, aux, n_aux, form, out_form, h_out);
}

#ifdef ZG_DEBUG_KNOBS
}  // extern "C"
const char* volatile zg_crumbs[8] = {"-", "-", "-", "-", "-", "-", "-", "-"};
thread_local int zg_crumb_slot = 4;
extern "C" {
int zg_debug_pk_progress(zg_pk* pk, char* buf, size_t cap) {
  if (!pk || !buf || !cap) return ZG_ERR_INVALID_ARGUMENT;
  std::string s = pk->impl->debug_progress();
  s += " crumbs:";
  for (int i = 0; i < 5; i++) s += std::string(" [") + std::to_string(i) + "] " + zg_crumbs[i];
  snprintf(buf, cap, "%s", s.c_str());
  return ZG_OK;
}
#endif

int zg_host_group_op(int curve, int group, int op, const void* a, const void* b, int form, void* out) {
  const GroupOps* ops = zg_group_ops(curve, group);
  if (!ops || !a || !b || !out) return ZG_ERR_INVALID_ARGUMENT;
  return ops->host_op(op, a, b, form, out);
}


int zg_group_fft(zg_ctx* ctx, int curve, int group, void* points, unsigned log_n, int inverse, int form) {
  const GroupOps* ops = zg_group_ops(curve, group);
  if (!ctx || !ops || !points || (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT)) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  return ops->group_fft(ctx, curve, points, log_n, inverse != 0, form);
}

int zg_group_scale(zg_ctx* ctx, int curve, int group, void* points, size_t n, const void* k, int form) {
  const GroupOps* ops = zg_group_ops(curve, group);
  if (!ctx || !ops || (n && !points) || !k || (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT)) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  return ops->group_scale(ctx, points, n, (const uint32_t*)k, form);
}

int zg_group_sparse_combine(zg_ctx* ctx, int curve, int group, const void* points, size_t n_points, const uint64_t* row_ptr,
                            const uint32_t* col, const void* coeff, size_t n_rows, int form, void* out_points) {
  const GroupOps* ops = zg_group_ops(curve, group);
  if (!ctx || !ops || (n_points && !points) || (n_rows && (!row_ptr || !out_points)) ||
      (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT))
    return ZG_ERR_INVALID_ARGUMENT;
  if (n_rows && row_ptr[n_rows] && (!col || !coeff)) return ZG_ERR_INVALID_ARGUMENT;
  ZG_HIP(hipSetDevice(ctx->device));
  return ops->group_combine(ctx, points, n_points, row_ptr, col, coeff, n_rows, form, out_points);
}

int zg_host_group_sum(int curve, int group, const void* points, size_t n, int form, void* out_affine) {
  const GroupOps* ops = zg_group_ops(curve, group);
  if (!ops || (n && !points) || !out_affine || (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT)) return ZG_ERR_INVALID_ARGUMENT;
  return ops->host_sum(points, n, form, out_affine);
}

int zg_host_points_check(int curve, int group, const void* points, size_t n, int form, size_t* first_bad) {
  const GroupOps* ops = zg_group_ops(curve, group);
  if (!ops || (n && !points) || !first_bad || (form != ZG_FORM_PLAIN && form != ZG_FORM_MONT)) return ZG_ERR_INVALID_ARGUMENT;
  return ops->points_check(points, n, form, first_bad);
}

}  // extern "C"

// ---- which ROCm runtime serves this process --------------------------------------------------------------------
namespace {
struct RtScan {
  std::vector<std::string> hip, hsa;
};
int rt_scan_cb(struct dl_phdr_info* info, size_t, void* data) {
  RtScan* r = static_cast<RtScan*>(data);
  const char* n = info->dlpi_name;
  if (!n || !*n) return 0;
  if (strstr(n, "libamdhip64")) r->hip.push_back(n);
  if (strstr(n, "libhsa-runtime64")) r->hsa.push_back(n);
  return 0;
}
std::string dir_of(const std::string& path) {
  const size_t k = path.rfind('/');
  return k == std::string::npos ? std::string() : path.substr(0, k);
}
}  // namespace

extern "C" int zg_runtime_info(char* buf, size_t cap, int* mixed) {
  if (!buf || cap == 0) return ZG_ERR_INVALID_ARGUMENT;
  RtScan r;
  dl_iterate_phdr(rt_scan_cb, &r);
  Dl_info bound;
  std::string mine;
  if (dladdr((void*)&hipGetDeviceCount, &bound) && bound.dli_fname) mine = bound.dli_fname;
  int ver = 0;
  (void)hipRuntimeGetVersion(&ver);
  bool mix = r.hip.size() > 1 || r.hsa.size() > 1;
  if (!mine.empty() && !r.hsa.empty() && dir_of(mine) != dir_of(r.hsa[0])) mix = true;  // HIP of one installation over HSA of another
  std::string out = "{\"hip_bound\": \"" + mine + "\", \"hip_runtime_version\": " + std::to_string(ver) + ", \"hip_loaded\": [";
  for (size_t i = 0; i < r.hip.size(); i++) out += std::string(i ? ", " : "") + "\"" + r.hip[i] + "\"";
  out += "], \"hsa_loaded\": [";
  for (size_t i = 0; i < r.hsa.size(); i++) out += std::string(i ? ", " : "") + "\"" + r.hsa[i] + "\"";
  out += std::string("], \"mixed_installations\": ") + (mix ? "true" : "false") + "}";
  if (mixed) *mixed = mix ? 1 : 0;
  if (out.size() + 1 > cap) return ZG_ERR_LENGTH;
  memcpy(buf, out.c_str(), out.size() + 1);
  return ZG_OK;
}

// ---- one multi-exponentiation over several devices -----------------------------------------------------------------
extern "C" int zg_msm_sharded(zg_bases* const* shards, size_t n_shards, const void* const* scalars, const size_t* counts,
                              int scalar_form, int scalars_mem, int out_form, void* out_affine) {
  if (!shards || !n_shards || !scalars || !counts || !out_affine) return ZG_ERR_INVALID_ARGUMENT;
  if (out_form != ZG_FORM_PLAIN && out_form != ZG_FORM_MONT) return ZG_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < n_shards; i++) {
    if (!shards[i] || (counts[i] && !scalars[i])) return ZG_ERR_INVALID_ARGUMENT;
    if (shards[i]->curve != shards[0]->curve || shards[i]->group != shards[0]->group) return ZG_ERR_INVALID_ARGUMENT;
    for (size_t j = 0; j < i; j++)
      if (shards[j]->ctx == shards[i]->ctx) return ZG_ERR_INVALID_ARGUMENT;  // calls on one context must not overlap
  }
  const size_t ab = shards[0]->ops->affine_bytes;
  std::vector<unsigned char> partial(n_shards * ab);
  std::vector<int> status(n_shards, ZG_OK);
  auto run = [&](size_t i) {
    // each shard: its own device, context and stream; partial result = one affine point (Montgomery)
    status[i] = zg_msm(shards[i], 0, scalars[i], counts[i], scalar_form, scalars_mem, ZG_FORM_MONT, partial.data() + i * ab);
  };
  std::vector<std::thread> workers;
  for (size_t i = 1; i < n_shards; i++) {
    try {
      workers.emplace_back(run, i);
    } catch (const std::system_error&) {
      run(i);  // no thread to be had: this shard runs on the calling thread
    }
  }
  run(0);
  for (auto& w : workers) w.join();
  for (size_t i = 0; i < n_shards; i++) ZG_TRY(status[i]);
  // the "sum over chunks" of libff::multi_exp: n_shards points, on the host
  unsigned char sum[4 * 48 * 2];
  ZG_TRY(shards[0]->ops->host_sum(partial.data(), n_shards, ZG_FORM_MONT, sum));
  if (out_form == ZG_FORM_PLAIN) shards[0]->ops->host_convert(sum, 1, 0);
  memcpy(out_affine, sum, ab);
  return ZG_OK;
}
