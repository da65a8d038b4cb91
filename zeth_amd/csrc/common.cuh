// Context, device workspace and error plumbing shared by the library's translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "../../include/zeth_gpu.h"
#include "knobs.h"

#define ZG_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      fprintf(stderr, "zeth_gpu: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return _e == hipErrorOutOfMemory ? ZG_ERR_OUT_OF_MEMORY : ZG_ERR_HIP;                            \
    }                                                                                                  \
  } while (0)

#define ZG_TRY(expr)          \
  do {                        \
    int _s = (expr);          \
    if (_s != ZG_OK) return _s; \
  } while (0)

// hipStreamSynchronize with a polling phase first.  The runtime's blocking wait wakes its caller 50-150 us after the
// stream has drained (measured: the wall time of a 0.5-1 ms multi-exponentiation moved in steps of ~0.15 ms while its
// kernels' times did not, profiles/r3_tune_table_c.jsonl); a result that the host tail is waiting for is therefore polled
// (hipStreamQuery, ~1 us a call) for up to `spin_us` before the thread gives up its core to the blocking wait.
inline hipError_t zg_stream_wait(hipStream_t st, unsigned spin_us = 4000) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > (long)spin_us) break;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  (void)hipGetLastError();  // hipErrorNotReady is not an error to keep
  return hipStreamSynchronize(st);
}

struct NttDomainBase {
  virtual ~NttDomainBase() {}
};

struct zg_ctx {
  std::map<uint32_t, NttDomainBase*> ntt_domains;  // key = curve << 8 | log m
  int device = 0;
  hipStream_t stream = nullptr;
  // second stream + events of the MSM's window-group pipeline (created on first use)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int side_init() {
    if (side_stream) return ZG_OK;
    ZG_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    ZG_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    ZG_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    return ZG_OK;
  }
  // grow-only workspace, bump-allocated per call
  char* ws = nullptr;
  size_t ws_cap = 0, ws_used = 0;
  // pinned host staging for asynchronous result copies + the MSM job in flight on this context
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  size_t job_n = 0;
  int job_W = 0, job_c = 0;
  int job_table_gbits = -1;  // >= 0: the job ran in fixed-base table mode with 2^gbits bucket groups
  int job_rows = 0, job_log_t = 0, job_log_k = 0;  // rows the host reads; log2 of segments per group and buckets per segment
  bool want_dense = false;  // count scalars other than 0/1 during recoding (prover lanes)
  // Bucket segments per window of the reduction (power of two; 0: the plan's own choice).  A caller whose evaluation is
  // not on its critical path (the prover's side lanes) asks for fewer: each segment's lane then pays the c-bit weighting
  // once for more buckets -- less work for the device, a longer dependent chain for this evaluation.
  uint32_t reduce_segments = 0;
  uint32_t table_unit_len = 0;  // table mode: entries per work unit (0: the plan's own length, sized for dense scalars)
  uint32_t table_segments = 0;  // table mode: at most this many bucket segments per group (0: the plan's own count)
  uint32_t job_dense = 0;  // scalars other than 0/1 seen by the last MSM on this context
  int pinned_reserve(size_t bytes) {
    if (bytes <= pinned_cap) return ZG_OK;
    ZG_CRUMB("pinned_reserve: hipHostFree/hipHostMalloc");
    if (pinned) ZG_HIP(hipHostFree(pinned));
    pinned = nullptr;
    pinned_cap = 0;
    size_t want = bytes < 65536 ? 65536 : bytes;
    ZG_HIP(hipHostMalloc(&pinned, want, hipHostMallocDefault));
    pinned_cap = want;
    return ZG_OK;
  }
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<const char*> ev_names;
  size_t ev_used = 0;
  std::vector<const char*> last_names;
  std::vector<float> last_ms;

  int ws_reserve(size_t bytes) {
    ws_used = 0;
    if (bytes <= ws_cap) return ZG_OK;
    ZG_CRUMB("ws_reserve: hipStreamSynchronize");
    ZG_HIP(hipStreamSynchronize(stream));
    ZG_CRUMB("ws_reserve: hipFree");
    if (ws) ZG_HIP(hipFree(ws));
    ws = nullptr;
    ws_cap = 0;
    size_t want = bytes + (bytes >> 3) + (1u << 20);
    ZG_CRUMB("ws_reserve: hipMalloc");
    ZG_HIP(hipMalloc((void**)&ws, want));
    ZG_CRUMB("ws_reserve: done");
    ws_cap = want;
    return ZG_OK;
  }
  template <class T>
  T* ws_take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (ws_used + bytes > ws_cap) return nullptr;
    T* p = reinterpret_cast<T*>(ws + ws_used);
    ws_used += bytes;
    return p;
  }
  static size_t ws_round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

  void prof_begin() {
    ev_used = 0;
    ev_names.clear();
  }
  // marks the START of phase `name` (and the end of the previous one)
  void mark(const char* name) {
    if (!profiling) return;
    if (ev_used == ev_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      ev_pool.push_back(e);
    }
    hipEventRecord(ev_pool[ev_used++], stream);
    ev_names.push_back(name);
  }
  // call after the stream has been synchronised
  void prof_end() {
    if (!profiling) return;
    last_names.clear();
    last_ms.clear();
    for (size_t i = 0; i + 1 < ev_used; i++) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev_pool[i], ev_pool[i + 1]);
      last_names.push_back(ev_names[i]);
      last_ms.push_back(ms);
    }
  }
};

// fixed-base table behind a set of bases (msm_table.cuh): D levels of N points, level j = 2^(j c) * level 0
struct MsmTable {
  size_t N, first;  // points per level; first base of the MSM's range
  int c, D;
};

struct GroupOps {
  size_t affine_bytes;
  size_t fq_limbs;  // limbs per coordinate-field base element
  int coords;       // field elements per affine point (2 for G1, 4 for G2)
  // d_pts: device affine points (Montgomery).  d_scalars: device Fr elements.
  int (*msm)(zg_ctx*, const void* d_pts, const uint32_t* d_scalars, size_t n, int scalars_mont, int force_c,
             int out_form, void* out_affine_host, const MsmTable* tab);
  // the same split in two: queue on the context's stream / wait + host tail (contexts run concurrently)
  int (*msm_launch)(zg_ctx*, const void* d_pts, const uint32_t* d_scalars, size_t n, int scalars_mont, int force_c,
                    const MsmTable* tab);
  int (*msm_finish)(zg_ctx*, int out_form, void* out_affine_host);
  int (*to_mont)(zg_ctx*, void* d_pts, size_t n);  // plain -> Montgomery in place
  int (*generate)(zg_ctx*, void* d_pts, size_t n, const uint32_t* t0, const uint32_t* d);
  int (*batch_exp)(zg_ctx*, void* d_pts, size_t n, const uint32_t* d_scalars_plain);
  int (*host_op)(int op, const void* a, const void* b, int form, void* out);
  // host-side conversion of `n` affine points between forms, in place (mont <-> plain)
  void (*host_convert)(void* pts, size_t n, int to_mont);
  // host-side curve-equation check of n affine points; *first_bad = n when all pass
  int (*points_check)(const void* pts, size_t n, int form, size_t* first_bad);
  // ceremony kernels (group_ops.cuh); host arrays in and out
  int (*group_fft)(zg_ctx*, int curve, void* points, unsigned log_n, int inverse, int form);
  int (*group_scale)(zg_ctx*, void* points, size_t n, const uint32_t* k_plain, int form);
  int (*group_combine)(zg_ctx*, const void* points, size_t n_points, const uint64_t* row_ptr, const uint32_t* col,
                       const void* coeff_plain, size_t n_rows, int form, void* out_points);
  int (*host_sum)(const void* pts, size_t n, int form, void* out_affine);
  int (*table_build)(zg_ctx*, void* d_table, size_t N, int c, int D);
  int (*table_params)(size_t N, int window_bits, int* c, int* D);
};

const GroupOps* zg_group_ops(int curve, int group);
int zg_ctx_create_prio(int device, int prio, zg_ctx** out);

struct PkBase {
  virtual ~PkBase() {}
  virtual int prove(const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form,
                    const uint32_t* r_plain, const uint32_t* s_plain, int out_form, void* proof_out) = 0;
  virtual int witness_map_out(const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form,
                              int out_form, void* h_out) = 0;
  virtual size_t domain_size() const = 0;
  virtual int clone_onto(zg_ctx* ctx, PkBase** out) const = 0;
#ifdef ZG_DEBUG_KNOBS
  // which of the proof's streams have drained and how far each got (phase marks); callable from a watchdog thread while
  // zg_prove is blocked (tools/diag_prio.py)
  virtual std::string debug_progress() const = 0;
#endif
};

struct zg_pk {
  zg_ctx* ctx;
  PkBase* impl;
  bool owns_ctx = false;  // clones carry a context of their own
};

// per-curve scalar-field entry points (fr_inst.hip)
struct FrOps {
  int (*ntt_inplace)(zg_ctx*, int curve, void* d_data, unsigned log_m, int inverse, int coset, int form);
  int (*pk_create)(zg_ctx*, int curve, const zg_pk_view* view, PkBase** out);
  // zg_ntt_noalloc: tables and transform buffer inside `scratch`; the tags name the domain whose tables it holds
  int (*ntt_static)(zg_ctx*, int curve, void* d_data, unsigned log_m, int inverse, int coset, int form, void* scratch,
                    size_t scratch_bytes, int* tag_curve, unsigned* tag_log_m);
  size_t (*ntt_static_bytes)();
};
const FrOps* zg_fr_ops(int curve);

struct zg_bases {
  zg_ctx* ctx;
  int curve, group;
  size_t n;
  void* d_pts;  // n points, or table_D levels of n points once zg_bases_precompute has run
  const GroupOps* ops;
  int table_c = 0, table_D = 0;
};
