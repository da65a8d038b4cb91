// Groth16 prover on the device: r1cs_to_qap_witness_map + the four query evaluations + blinding.
//
// Restates libsnark::r1cs_gg_ppzksnark_prover(pk, primary, aux, force_pow_2_domain = true) as Zeth calls it
// (libzeth/snarks/groth16/groth16_snark.tcc:36-46) -- SURVEY.md 3.1 step 5:
//   a. aA_i = <a_i, w>, aB_i = <b_i, w>, aC_i = <c_i, w> (+ the num_inputs + 1 input-consistency rows of A),
//      iFFT, cosetFFT (x3), H = (aA o aB - aC) / Z(g), icosetFFT -> coefficients of H (m + 1, top two zero)
//   c. At = <A_query, 1||w>, Bt = <B_query, 1||w> in G2 and G1, Ht = <H_query, H[0..m-1)>, Lt = <L_query, aux>
//   d. g_A = alpha + At + r delta ; g1_B = beta + Bt.h + s delta ; g_B = beta2 + Bt.g + s delta2 ;
//      g_C = Ht + Lt + s g_A + r g1_B - (r s) delta
// (r, s) are arguments here: libsnark draws them inside the prover, Zeth's API has no hook.
// An unsatisfied assignment is not an error: like the reference's release build it yields a proof that fails
// verification (libzeth/tests/prover/prover_test.cpp:703-714).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>

#include "msm_host.cuh"
#include "ntt_host.cuh"

namespace zg {

// A host thread that lives as long as its prover and runs one job at a time: a proof hands each of its four lanes'
// queueing (and, for large assignments, a slice of the upload) to one.  Creating four threads per proof cost the calling
// thread ~0.1 ms of every proof; when a thread cannot be had the job runs on the caller.
class LaneWorker {
 public:
  LaneWorker() {
    try {
      th_ = std::thread([this] { loop(); });
    } catch (const std::system_error&) {
    }
  }
  ~LaneWorker() {
    if (th_.joinable()) {
      {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
      }
      cv_.notify_all();
      th_.join();
    }
  }
  LaneWorker(const LaneWorker&) = delete;
  void submit(std::function<void()> job) {
    if (!th_.joinable()) {
      job();
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = std::move(job);
      has_job_ = true;
      done_ = false;
    }
    cv_.notify_all();
  }
  void wait() {
    if (!th_.joinable()) return;
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return done_; });
  }

 private:
  void loop() {
    for (;;) {
      std::function<void()> j;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return has_job_ || stop_; });
        if (!has_job_) return;
        j = std::move(job_);
        has_job_ = false;
      }
      j();
      {
        std::lock_guard<std::mutex> lk(m_);
        done_ = true;
      }
      cv_.notify_all();
    }
  }
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool has_job_ = false, done_ = true, stop_ = false;
};

struct CsrDev {
  uint64_t* row_ptr = nullptr;
  uint32_t* col = nullptr;
  void* val = nullptr;  // Fr, Montgomery
  size_t nnz = 0;
};

// w_full[0] = 1, w_full[1 + i] = assignment[i] (converted to Montgomery when the caller passed plain integers)
template <class Fr>
__global__ void __launch_bounds__(256) k_build_w(const Fr* __restrict__ primary, size_t n_primary,
                                                  const Fr* __restrict__ aux, size_t n_aux, int plain,
                                                  Fr* __restrict__ w_full, size_t lo, size_t hi) {  // entries [lo, hi) of 1 || w
  size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hi || i > n_primary + n_aux) return;
  Fr x;
  if (i == 0) {
  ZG_WAVE_PRIO(2);
    x = Fr::one();
  } else {
    x = (i - 1 < n_primary) ? glob_get(primary + (i - 1)) : glob_get(aux + (i - 1 - n_primary));
    if (plain) x = Fr::to_mont(x);
  }
  glob_put(w_full + i, x);
}

// The three constraint matrices of a key, evaluated by ONE launch (blockIdx.y picks the matrix).
struct SpmvMatrices {
  const uint64_t* row_ptr[3];
  const uint32_t* col[3];
  const void* val[3];
};

// Most coefficients of a boolean-heavy circuit are +1 or -1.  Once per key the column indices of those entries are
// tagged in their two top bits (1: the coefficient is one, 2: minus one; indices are below 2^30), so that the constraint
// evaluation neither loads nor multiplies their 32-byte coefficients.
static constexpr uint32_t CSR_COL_MASK = (1u << 30) - 1u;
template <class Fr>
__global__ void __launch_bounds__(256) k_csr_tag_units(uint32_t* __restrict__ col, const Fr* __restrict__ val /* Montgomery */, size_t nnz) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz) return;
  const Fr v = glob_get(val + t);
  if (v == Fr::one()) col[t] |= 1u << 30;
  else if (v == Fr::neg(Fr::one())) col[t] |= 2u << 30;
}

// out[k m + i] = <row_i of matrix k, w> for i < nc; matrix A (k = 0) additionally carries w_0..w_num_inputs in rows
// nc..nc+num_inputs (libsnark r1cs_to_qap: the extra constraints that make the A polynomials linearly independent);
// the rest of each m-sized vector is zero.
template <class Fr>
__global__ void __launch_bounds__(256) k_spmv(SpmvMatrices M, const Fr* __restrict__ w, size_t nc, size_t m, size_t extra_rows,
                                               Fr* __restrict__ out) {
  ZG_WAVE_PRIO(2);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const unsigned k = blockIdx.y;
  const uint64_t* __restrict__ row_ptr = M.row_ptr[k];
  const uint32_t* __restrict__ col = M.col[k];
  const Fr* __restrict__ val = static_cast<const Fr*>(M.val[k]);
  Fr acc = Fr::zero();
  if (i < nc) {
    for (uint64_t t = row_ptr[i]; t < row_ptr[i + 1]; t++) {
      const uint32_t c = col[t], kind = c >> 30;  // k_csr_tag_units
      const Fr x = glob_get(w + (c & CSR_COL_MASK));
      if (kind == 1u) acc = Fr::add(acc, x);
      else if (kind == 2u) acc = Fr::sub(acc, x);
      else acc = Fr::add(acc, Fr::mul(glob_get(val + t), x));
    }
  } else if (k == 0 && i - nc < extra_rows) {
    acc = glob_get(w + (i - nc));
  }
  glob_put(out + (size_t)k * m + i, acc);
}

template <class Fr>
__global__ void __launch_bounds__(256) k_h_pointwise(Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c,
                                                      Fr z_inv, size_t m) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  Fr x = Fr::sub(Fr::mul(glob_get(a + i), glob_get(b + i)), glob_get(c + i));
  glob_put(a + i, Fr::mul(x, z_inv));
}

template <class Fr>
__global__ void __launch_bounds__(256) k_gather(const Fr* __restrict__ w, const uint32_t* __restrict__ idx, size_t n,
                                                 Fr* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  glob_put(out + i, glob_get(w + idx[i]));
}

template <class Fq, class Fq2T, class Fr>
struct ProvingKeyDev : PkBase {
  zg_ctx* ctx = nullptr;
  int curve = 0;
  size_t num_inputs = 0, num_vars = 0, nc = 0, m = 0;
  unsigned log_m = 0;
  CsrDev mat[3];
  Affine<Fq> alpha_g1, beta_g1, delta_g1;  // Montgomery, host
  Affine<Fq2T> beta_g2, delta_g2;
  Affine<Fq>*A = nullptr, *B1 = nullptr, *H = nullptr, *L = nullptr;
  Affine<Fq2T>* B2 = nullptr;
  uint32_t* B_idx = nullptr;
  size_t A_len = 0, B_len = 0, H_len = 0, L_len = 0;
  // H query as a fixed-base table (msm_table.cuh) when it is large: H then holds H_levels levels of H_len points.  The H
  // scalars are the only dense ones of a proof (uniform field elements); A, B and L are mostly 0/1 and stay windowed.
  int H_c = 0, H_levels = 0;
  // The A, B (G1 and G2) and L queries likewise (round 3): their scalars are the witness -- mostly 0 and 1, which cost one
  // entry each in either mode, plus the "dense" ones, which take D = 16 table digits instead of 22 windowed ones -- and
  // one bucket set per evaluation replaces 22 windows' worth of reductions and the host's Horner over them.
  int A_c = 0, A_levels = 0, B1_c = 0, B1_levels = 0, B2_c = 0, B2_levels = 0, L_c = 0, L_levels = 0;
  // per-proof scratch (device): 1||w, three evaluation vectors, one transform buffer, gathered B scalars, inputs
  // (`assign` = the assignment as the caller passed it, behind one more element that holds the plain integer 1: with a
  // plain assignment the witness lanes read their scalars 1 || w straight from `assign_base`)
  Fr *w = nullptr, *ev[3] = {nullptr, nullptr, nullptr}, *tmp = nullptr, *bsc = nullptr, *assign = nullptr, *assign_base = nullptr;
  Fr z_inv;
  // lanes 0-3: the B (G2), A, L and B (G1) evaluations; lanes 4, 5 (large proofs only, `early`): the A and L evaluations
  // over the first half of the assignment, started while the second half is still crossing PCIe
  zg_ctx* lane[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  LaneWorker* worker[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool early = false;
  hipEvent_t w_half = nullptr, bsc_ready = nullptr;
  // page-locked staging of the parallel upload: two halves per lane, and the events that say a half's copy has left it
  static constexpr size_t STAGE_BYTES = (size_t)2 << 20;
  // (debug builds: ZG_PAR_UPLOAD_MIN_KB / ZG_PAR_STAGE_KB shrink the threshold and the staging halves, so that a test-sized
  // proof runs the staged upload with several chunks per lane -- tests/test_gpu_prover.py)
  size_t stage_bytes = zg_knob("ZG_PAR_STAGE_KB") ? (size_t)atoi(zg_knob("ZG_PAR_STAGE_KB")) << 10 : STAGE_BYTES;
  size_t par_upload_min = zg_knob("ZG_PAR_UPLOAD_MIN_KB") ? (size_t)atoi(zg_knob("ZG_PAR_UPLOAD_MIN_KB")) << 10 : PAR_UPLOAD_MIN;
  char* stage[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_ev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  uint32_t dense_hint[4] = {0, 0, 0, 0};
  hipEvent_t w_ready = nullptr;
  bool owns_key = true;  // false for a clone: bases and matrices belong to the key it was cloned from

  ~ProvingKeyDev() override {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& wk : worker) delete wk;  // (joins the thread)
    for (int i = 0; i < 4; i++) {
      if (stage[i]) hipHostFree(stage[i]);
      for (auto& e : stage_ev[i])
        if (e) hipEventDestroy(e);
    }
    for (auto& l : lane)
      if (l) zg_ctx_destroy(l);
    if (w_ready) hipEventDestroy(w_ready);
    if (w_half) hipEventDestroy(w_half);
    if (bsc_ready) hipEventDestroy(bsc_ready);
    if (owns_key) {
      for (auto& M : mat) {
        hipFree(M.row_ptr);
        hipFree(M.col);
        hipFree(M.val);
      }
      hipFree(A); hipFree(B1); hipFree(B2); hipFree(H); hipFree(L); hipFree(B_idx);
    }
    hipFree(w); hipFree(ev[0]); hipFree(tmp); hipFree(bsc); hipFree(assign_base);
  }

  template <class T>
  int dev_copy(T** dst, const void* src, size_t count) {
    *dst = nullptr;
    ZG_HIP(hipMalloc((void**)dst, (count ? count : 1) * sizeof(T)));
    if (count) ZG_HIP(hipMemcpyAsync(*dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return ZG_OK;
  }

  int create(zg_ctx* c, int curve_, const zg_pk_view* v) {
    ctx = c;
    curve = curve_;
    num_inputs = v->num_inputs;
    num_vars = v->num_variables;
    nc = v->num_constraints;
    if (num_inputs > num_vars) return ZG_ERR_INVALID_ARGUMENT;
    // force_pow_2_domain = true (groth16_snark.tcc:43-45): smallest power of two >= nc + num_inputs + 1
    m = 2;
    log_m = 1;
    while (m < nc + num_inputs + 1) {
      m <<= 1;
      log_m++;
    }
    if (v->A_len != num_vars + 1 || v->H_len != m - 1 || v->L_len != num_vars - num_inputs || v->B_len > num_vars + 1)
      return ZG_ERR_INVALID_ARGUMENT;
    if (num_vars >= CSR_COL_MASK) return ZG_ERR_UNSUPPORTED;  // column indices carry two tag bits (k_csr_tag_units)
    const zg_r1cs_matrix* src[3] = {&v->a, &v->b, &v->c};
    const int plain = v->form == ZG_FORM_PLAIN;
    hipStream_t st = ctx->stream;
    for (int k = 0; k < 3; k++) {
      if (nc && (!src[k]->row_ptr || src[k]->row_ptr[0] != 0)) return ZG_ERR_INVALID_ARGUMENT;
      size_t nnz = nc ? (size_t)src[k]->row_ptr[nc] : 0;
      if (nnz && (!src[k]->col || !src[k]->val)) return ZG_ERR_INVALID_ARGUMENT;
      for (size_t i = 0; i < nc; i++)
        if (src[k]->row_ptr[i] > src[k]->row_ptr[i + 1]) return ZG_ERR_INVALID_ARGUMENT;
      for (size_t t = 0; t < nnz; t++)
        if (src[k]->col[t] > num_vars) return ZG_ERR_INVALID_ARGUMENT;
      mat[k].nnz = nnz;
      std::vector<uint64_t> zero_rows;
      const uint64_t* rp = src[k]->row_ptr;
      if (!nc) {
        zero_rows.assign(1, 0);
        rp = zero_rows.data();
      }
      ZG_TRY(dev_copy(&mat[k].row_ptr, rp, nc + 1));
      ZG_TRY(dev_copy(&mat[k].col, src[k]->col, nnz));
      Fr* val = nullptr;
      ZG_TRY(dev_copy(&val, src[k]->val, nnz));
      mat[k].val = val;
      if (plain && nnz)
        hipLaunchKernelGGL((k_fr_convert<Fr>), dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, val, nnz, 1);
      if (nnz)
        hipLaunchKernelGGL((k_csr_tag_units<Fr>), dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, mat[k].col, val, nnz);
    }
    const GroupOps* o1 = zg_group_ops(curve, ZG_GROUP_G1);
    const GroupOps* o2 = zg_group_ops(curve, ZG_GROUP_G2);
    auto up1 = [&](Affine<Fq>** dst, const void* s, size_t n) -> int {
      ZG_TRY(dev_copy(dst, s, n));
      if (plain && n) ZG_TRY(o1->to_mont(ctx, *dst, n));
      return ZG_OK;
    };
    ZG_TRY(up1(&A, v->A_query, v->A_len));
    ZG_TRY(up1(&B1, v->B_g1, v->B_len));
    ZG_TRY(up1(&H, v->H_query, v->H_len));
    ZG_TRY(up1(&L, v->L_query, v->L_len));
    // ZG_PK_TABLE_MIN: smallest domain whose queries get fixed-base tables (default 2^18: a JoinSplit-scale proof, domain
    // 2^19, takes 5.15 ms with the H table and 5.5 ms without; 0 = never)
    const char* e_min = getenv("ZG_PK_TABLE_MIN");
    const size_t table_min = e_min ? strtoull(e_min, nullptr, 0) : ((size_t)1 << 18);
    const bool want_tables = table_min && v->H_len + 1 >= table_min;  // H_len = domain size - 1
    // replaces *pts (len points) by a table of D levels when there is room for it; the plain query proves just as well
    auto make_table = [&](const GroupOps* o, void** pts, size_t len, size_t point_bytes, int want_c, int* c_out, int* D_out) -> int {
      if (!want_tables || len < 1024) return ZG_OK;
      int c = 0, D = 0;
      ZG_TRY(o->table_params(len, want_c, &c, &D));
      if ((uint64_t)D * len >= (1ull << 31)) return ZG_OK;
      void* table = nullptr;
      if (hipMalloc(&table, (size_t)D * len * point_bytes) != hipSuccess) {
        (void)hipGetLastError();  // no room for the table
        return ZG_OK;
      }
      ZG_HIP(hipMemcpyAsync(table, *pts, len * point_bytes, hipMemcpyDeviceToDevice, st));
      const int s = o->table_build(ctx, table, len, c, D);
      if (s != ZG_OK) {
        hipFree(table);
        return s;
      }
      hipFree(*pts);
      *pts = table;
      *c_out = c;
      *D_out = D;
      return ZG_OK;
    };
    {
      static const int h_table_c = zg_knob("ZG_PK_TABLE_C") ? atoi(zg_knob("ZG_PK_TABLE_C")) : 0;  // tuning knob
      // Up to 2^19 points a 17-bit table (16 digits, 2^16 buckets in two groups) instead of the lone evaluation's 20 bits:
      // alone the two take the same time, inside a proof the eight times smaller reduction leaves the device to the
      // other evaluations (JoinSplit shape, four proofs in flight: 375 -> 385-398 proofs/s, latency unchanged).  From
      // 2^20 points on 20 bits win alone and inside a proof (BLS12-377, 2^20 constraints: 11.9 against 12.9 ms).
      const int want_c = h_table_c ? h_table_c : (v->H_len < ((size_t)1 << 19) ? 17 : 0);
      ZG_TRY(make_table(o1, (void**)&H, v->H_len, sizeof(Affine<Fq>), want_c, &H_c, &H_levels));
      // the witness queries: 17 bits (tuning knob ZG_PK_LANE_TABLE_C; 0 there = no tables for them)
      static const int lane_table_c = zg_knob("ZG_PK_LANE_TABLE_C") ? atoi(zg_knob("ZG_PK_LANE_TABLE_C")) : 17;
      if (lane_table_c) {
        ZG_TRY(make_table(o1, (void**)&A, v->A_len, sizeof(Affine<Fq>), lane_table_c, &A_c, &A_levels));
        ZG_TRY(make_table(o1, (void**)&B1, v->B_len, sizeof(Affine<Fq>), lane_table_c, &B1_c, &B1_levels));
        ZG_TRY(make_table(o1, (void**)&L, v->L_len, sizeof(Affine<Fq>), lane_table_c, &L_c, &L_levels));
      }
    }
    ZG_TRY(dev_copy(&B2, v->B_g2, v->B_len));
    if (plain && v->B_len) ZG_TRY(o2->to_mont(ctx, B2, v->B_len));
    {
      static const int lane_table_c = zg_knob("ZG_PK_G2_TABLE_C")     ? atoi(zg_knob("ZG_PK_G2_TABLE_C"))
                                      : zg_knob("ZG_PK_LANE_TABLE_C") ? atoi(zg_knob("ZG_PK_LANE_TABLE_C"))
                                                                      : 17;
      if (lane_table_c) ZG_TRY(make_table(o2, (void**)&B2, v->B_len, sizeof(Affine<Fq2T>), lane_table_c, &B2_c, &B2_levels));
    }
    for (size_t i = 0; i < v->B_len; i++)
      if (v->B_indices[i] > num_vars) return ZG_ERR_INVALID_ARGUMENT;
    ZG_TRY(dev_copy(&B_idx, v->B_indices, v->B_len));
    A_len = v->A_len; B_len = v->B_len; H_len = v->H_len; L_len = v->L_len;
    memcpy(&alpha_g1, v->alpha_g1, sizeof(alpha_g1));
    memcpy(&beta_g1, v->beta_g1, sizeof(beta_g1));
    memcpy(&delta_g1, v->delta_g1, sizeof(delta_g1));
    memcpy(&beta_g2, v->beta_g2, sizeof(beta_g2));
    memcpy(&delta_g2, v->delta_g2, sizeof(delta_g2));
    if (plain) {
      host_affine_convert(&alpha_g1, 1, 1);
      host_affine_convert(&beta_g1, 1, 1);
      host_affine_convert(&delta_g1, 1, 1);
      host_affine_convert(&beta_g2, 1, 1);
      host_affine_convert(&delta_g2, 1, 1);
    }
    ZG_TRY(alloc_scratch());
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipStreamSynchronize(st));
    return ZG_OK;
  }

  // side streams and per-proof device scratch (everything a proof in flight writes)
  int alloc_scratch() {
    // (debug builds: ZG_LANE_PRIO = 1 | 0 | -1 puts the lanes at the highest / default / lowest stream priority and
    // ZG_LANE_PROFILE turns their phase marks on, for tools/diag_prio.py)
    static const int lane_prio = zg_knob("ZG_LANE_PRIO") ? atoi(zg_knob("ZG_LANE_PRIO")) : +1;
    // A proof whose assignment takes the staged upload (>= 32 MB) and whose A and L queries have tables evaluates the
    // first half of both while the second half is uploaded: BLS12-377, 2^22 constraints, 168 MB -- the device no longer
    // sits idle for the second 1.6 ms of the upload (debug builds: ZG_PROVE_NO_EARLY turns it off)
    early = (num_vars * sizeof(Fr)) >= par_upload_min && A_levels > 0 && L_levels > 0 && num_vars - num_inputs >= 8192 &&
            !zg_knob("ZG_PROVE_NO_EARLY");
    const int n_lanes = early ? 6 : 4;
    for (int li = 0; li < n_lanes; li++) {
      zg_ctx*& l = lane[li];
      ZG_TRY(zg_ctx_create_prio(ctx->device, lane_prio, &l));
      l->want_dense = true;
      l->profiling = zg_knob("ZG_LANE_PROFILE") != nullptr;
      if (zg_knob("ZG_LANE_PREALLOC_MB")) ZG_TRY(l->ws_reserve((size_t)atoi(zg_knob("ZG_LANE_PREALLOC_MB")) << 20));
      // The lanes' evaluations are off the proof's critical path (the H chain on the main stream): 128 bucket segments a
      // window instead of 1024 cut their reduction's instruction count 2.6x (profiles/r2_prove_valu_by_kernel.txt: the
      // reductions were 37 % of a proof's vector instructions), worth +10 % proofs/s with four proofs in flight.
      // (set per proof in prove_impl: lane_segments())
    }  // the SAME priority as the main stream.  With mixed priorities and three or more lanes queued at once by their own
       // threads a proof never completes (no fault, killable; every queue pending, none served).  Bisected in round 3
       // (tools/diag_prio.py, profiles/r3_stream_priority_bisect.jsonl): not the cross-stream event, not
       // reallocation, not scratch, not GPU_MAX_HW_QUEUES; a pure-HIP reproducer of the same shape never stalls.
    for (int i = 4; i < n_lanes; i++) worker[i] = new LaneWorker();
    ZG_HIP(hipEventCreateWithFlags(&w_half, hipEventDisableTiming));
    ZG_HIP(hipEventCreateWithFlags(&bsc_ready, hipEventDisableTiming));
    for (int i = 0; i < 4; i++) {
      worker[i] = new LaneWorker();
      if ((num_vars * sizeof(Fr)) >= par_upload_min) {
        ZG_HIP(hipHostMalloc((void**)&stage[i], 2 * stage_bytes, hipHostMallocDefault));
        for (auto& e : stage_ev[i]) ZG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
    }
    ZG_HIP(hipEventCreateWithFlags(&w_ready, hipEventDisableTiming));
    ZG_HIP(hipMalloc((void**)&w, (num_vars + 1) * sizeof(Fr)));
    // the three evaluation vectors are one allocation, m elements apart, and so is the transform buffer: the transforms
    // of all three run as one batched launch
    ZG_HIP(hipMalloc((void**)&ev[0], 3 * m * sizeof(Fr)));
    ev[1] = ev[0] + m;
    ev[2] = ev[1] + m;
    ZG_HIP(hipMalloc((void**)&tmp, 3 * m * sizeof(Fr)));
    ZG_HIP(hipMalloc((void**)&bsc, (B_len ? B_len : 1) * sizeof(Fr)));
    ZG_HIP(hipMalloc((void**)&assign_base, (num_vars + 1) * sizeof(Fr)));
    assign = assign_base + 1;
    {
      Fr plain_one = Fr::zero();
      plain_one.v[0] = 1;
      ZG_HIP(hipMemcpy(assign_base, &plain_one, sizeof(Fr), hipMemcpyHostToDevice));
    }
    // 1 / Z(g) = 1 / (g^m - 1)   (divide_by_Z_on_coset)
    using P = typename Fr::Params;
    Fr g = fr_from_limbs<Fr>(P::GEN_MONT);
    z_inv = Fr::inv(Fr::sub(host_pow_u64(g, m), Fr::one()));
    NttDomain<Fr>* D = nullptr;
    ZG_TRY(ntt_get_domain<Fr>(ctx, curve, log_m, &D));
    return ZG_OK;
  }

  // a second prover over the same resident key: shares bases and matrices, owns its scratch and streams
  int clone_onto(zg_ctx* c, PkBase** out) const override {
    auto* p = new ProvingKeyDev<Fq, Fq2T, Fr>();
    p->owns_key = false;
    p->ctx = c;
    p->curve = curve;
    p->num_inputs = num_inputs; p->num_vars = num_vars; p->nc = nc; p->m = m; p->log_m = log_m;
    for (int k = 0; k < 3; k++) p->mat[k] = mat[k];
    p->alpha_g1 = alpha_g1; p->beta_g1 = beta_g1; p->delta_g1 = delta_g1; p->beta_g2 = beta_g2; p->delta_g2 = delta_g2;
    p->A = A; p->B1 = B1; p->H = H; p->L = L; p->B2 = B2; p->B_idx = B_idx;
    p->A_len = A_len; p->B_len = B_len; p->H_len = H_len; p->L_len = L_len;
    p->H_c = H_c; p->H_levels = H_levels;
    p->A_c = A_c; p->A_levels = A_levels; p->B1_c = B1_c; p->B1_levels = B1_levels;
    p->B2_c = B2_c; p->B2_levels = B2_levels; p->L_c = L_c; p->L_levels = L_levels;
    int s = p->alloc_scratch();
    if (s == ZG_OK && hipStreamSynchronize(c->stream) != hipSuccess) s = ZG_ERR_HIP;
    if (s != ZG_OK) {
      delete p;
      return s;
    }
    *out = p;
    return ZG_OK;
  }

  // leaves the m + 1 coefficients of H (Montgomery; entry m is zero by construction and not stored) in `tmp`
  // and 1||w in `w`
  int witness_map(const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form) {
    ZG_TRY(load_assignment(primary, n_primary, aux, n_aux, form));
    return qap_h();
  }

  // Assignments of this size or more are uploaded by the four lane threads, a quarter each, through page-locked staging:
  // the runtime's own path for pageable memory reaches ~29 GB/s (BLS12-377, 2^22 constraints: 168 MB, 5.8 of 33 ms with
  // nothing else able to start), four copying threads with their transfers in flight reach 53 GB/s (3.2 ms).  Not below
  // 32 MB: a JoinSplit-sized assignment (17.5 MB) gains 0.1 ms alone and loses 8 % of the throughput with four proofs in
  // flight (sixteen copying threads on the box's cores).
  static constexpr size_t PAR_UPLOAD_MIN = (size_t)32 << 20;

  // lane i's share of the upload: src -> staging halves -> dst, the copy of one half in flight while the other is filled
  int upload_slice(int i, char* dst, const char* src, size_t bytes) {
    if (hipSetDevice(ctx->device) != hipSuccess) return ZG_ERR_HIP;
    hipStream_t ls = lane[i]->stream;
    size_t k = 0;
    for (size_t off = 0; off < bytes; off += stage_bytes, k++) {
      const size_t len = bytes - off < stage_bytes ? bytes - off : stage_bytes;
      const int h = (int)(k & 1);
      if (k >= 2) ZG_HIP(hipEventSynchronize(stage_ev[i][h]));
      memcpy(stage[i] + h * stage_bytes, src + off, len);
      ZG_HIP(hipMemcpyAsync(dst + off, stage[i] + h * stage_bytes, len, hipMemcpyHostToDevice, ls));
      ZG_HIP(hipEventRecord(stage_ev[i][h], ls));
    }
    ZG_HIP(hipStreamSynchronize(ls));
    return ZG_OK;
  }

  // 1 || w in Montgomery form on the device.  `on_half` (may be empty; used when `early` is set and the upload is staged):
  // called with h once aux[0, h) is on the device and entries [0, 1 + n_primary + h) of 1 || w are queued on the main
  // stream with `w_half` recorded behind them -- the rest of aux is uploaded after it returns.
  int load_assignment(const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form,
                      const std::function<int(size_t)>& on_half = nullptr) {
    if (n_primary != num_inputs || n_primary + n_aux != num_vars) return ZG_ERR_INVALID_ARGUMENT;
    hipStream_t st = ctx->stream;
    if (n_primary) ZG_HIP(hipMemcpyAsync(assign, primary, n_primary * sizeof(Fr), hipMemcpyHostToDevice, st));
    const size_t aux_bytes = n_aux * sizeof(Fr);
    const int plain = form == ZG_FORM_PLAIN;
    auto build_w = [&](size_t lo, size_t hi) {  // entries [lo, hi) of 1 || w
      if (hi > lo)
        hipLaunchKernelGGL((k_build_w<Fr>), dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, st, assign, n_primary,
                           assign + n_primary, n_aux, plain, w, lo, hi);
    };
    static const bool no_par_upload = zg_knob("ZG_PROVE_NO_PAR_UPLOAD") != nullptr;  // measurement knob
    if (aux_bytes >= par_upload_min && stage[0] && !no_par_upload) {
      auto upload_range = [&](size_t b0, size_t b1) -> int {  // bytes [b0, b1) of aux, a quarter per lane thread
        int status[4] = {ZG_OK, ZG_OK, ZG_OK, ZG_OK};
        const size_t bytes = b1 - b0;
        const size_t per = ((bytes / 4) + 255) & ~(size_t)255;
        for (int i = 0; i < 4; i++) {
          const size_t lo = per * i < bytes ? per * i : bytes, hi = lo + per < bytes ? lo + per : bytes;
          char* dst = reinterpret_cast<char*>(assign + n_primary) + b0 + lo;
          const char* src = static_cast<const char*>(aux) + b0 + lo;
          worker[i]->submit([this, i, dst, src, lo, hi, &status] { status[i] = upload_slice(i, dst, src, hi - lo); });
        }
        for (int i = 0; i < 4; i++) worker[i]->wait();
        for (int i = 0; i < 4; i++) ZG_TRY(status[i]);
        return ZG_OK;
      };
      // (debug builds: ZG_PROVE_EARLY_PCT = the first part's share of the assignment, in percent)
      static const size_t early_pct = zg_knob("ZG_PROVE_EARLY_PCT") ? (size_t)atoi(zg_knob("ZG_PROVE_EARLY_PCT")) : 50;
      const size_t half = (early && on_half) ? n_aux * early_pct / 100 : 0;
      if (half) {
        ZG_TRY(upload_range(0, half * sizeof(Fr)));
        build_w(0, 1 + n_primary + half);
        ZG_HIP(hipEventRecord(w_half, st));
        ZG_TRY(on_half(half));
        ZG_TRY(upload_range(half * sizeof(Fr), aux_bytes));
        build_w(1 + n_primary + half, num_vars + 1);
      } else {
        ZG_TRY(upload_range(0, aux_bytes));
        build_w(0, num_vars + 1);
      }
    } else {
      if (n_aux) ZG_HIP(hipMemcpyAsync(assign + n_primary, aux, aux_bytes, hipMemcpyHostToDevice, st));
      build_w(0, num_vars + 1);
    }
    return ZG_OK;
  }

  // constraint evaluation, the seven transforms and the pointwise pass: H's coefficients end up in `tmp`.  Three parts
  // (0: constraint evaluation + iFFTs, 1: cosetFFTs, 2: pointwise + icosetFFT) so that the prover can place an event
  // between them.
  int qap_h(int first_part = 0, int last_part = 2) {
    hipStream_t st = ctx->stream;
    const unsigned mb = (unsigned)((m + 255) / 256);
    NttDomain<Fr>* D = nullptr;
    ZG_TRY(ntt_get_domain<Fr>(ctx, curve, log_m, &D));
    // With the domain's full-size tables (ntt_host.cuh) the chain is: iFFT whose output scaling also carries g^j, a PLAIN
    // FFT (that is the coset FFT then), and an icosetFFT whose load builds (A B - C) / Z(g) from the three vectors -- two
    // products per element fewer in each of the three coset FFTs, and no pointwise kernel.  Without them: the separate steps.
    const bool fused = D->full != nullptr;
    if (first_part <= 0 && last_part >= 0) {
      SpmvMatrices M;
      for (int k = 0; k < 3; k++) {
        M.row_ptr[k] = mat[k].row_ptr;
        M.col[k] = mat[k].col;
        M.val[k] = mat[k].val;
      }
      hipLaunchKernelGGL((k_spmv<Fr>), dim3(mb, 3), dim3(256), 0, st, M, (const Fr*)w, nc, m, num_inputs + 1, ev[0]);
      ZG_TRY(ntt_launch<Fr>(ctx, D, ev[0], tmp, 1, 0, 3, m, fused ? NTT_FUSE_COSET_NEXT : 0));  // iFFT of A, B, C
    }
    if (first_part <= 1 && last_part >= 1) ZG_TRY(ntt_launch<Fr>(ctx, D, tmp, ev[0], 0, fused ? 0 : 1, 3, m));  // cosetFFT of A, B, C
    if (first_part <= 2 && last_part >= 2) {
      if (fused) {
        ZG_TRY(ntt_launch<Fr>(ctx, D, ev[0], tmp, 1, 1, 1, m, NTT_QAP_IN));  // (A B - C) / Z, icosetFFT
      } else {
        hipLaunchKernelGGL((k_h_pointwise<Fr>), dim3(mb), dim3(256), 0, st, ev[0], ev[1], ev[2], z_inv, m);
        ZG_TRY(ntt_launch<Fr>(ctx, D, ev[0], tmp, 1, 1));  // icosetFFT
      }
    }
    ZG_HIP(hipGetLastError());
    return ZG_OK;
  }

  int prove_impl(const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form,
                 const uint32_t* r_plain, const uint32_t* s_plain, int out_form, void* proof_out) {
    const bool trace = zg_knob("ZG_PROVE_TRACE") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
      if (trace) fprintf(stderr, "  [prove] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    };
    ZG_CRUMB_SLOT(4);
    ZG_CRUMB("prove: load_assignment");
    const GroupOps* o1 = zg_group_ops(curve, ZG_GROUP_G1);
    const GroupOps* o2 = zg_group_ops(curve, ZG_GROUP_G2);
    // Large proofs (`early`): the A and L evaluations are split at the middle of the assignment.  The first halves are
    // queued on lanes 4 and 5 as soon as the first half of the assignment is on the device, by their own threads, while
    // this thread goes on to upload the second half; lanes 1 and 2 then take the second halves (MsmTable::first).
    // The witness lanes' scalars: a PLAIN assignment is read as the caller passed it (1 || w with the plain one in front,
    // `assign_base`) -- the evaluations recode plain integers, so the Montgomery copy that k_build_w makes for the
    // constraint evaluation would only be converted back, twice per scalar and evaluation
    const Fr* const lane_w = form == ZG_FORM_PLAIN ? assign_base : w;
    const int lane_mont = form == ZG_FORM_PLAIN ? 0 : 1;
    size_t a_split = 0, l_split = 0;  // bases [0, split) go to the early lanes
    int early_status[2] = {ZG_OK, ZG_OK};
    struct JoinEarly {  // the early lanes' threads use this frame: they are joined on every way out, and after an error
      ProvingKeyDev* pk;  // their streams are drained like the others' (Drain below)
      bool armed = true;
      ~JoinEarly() {
        for (int i = 4; i < 6; i++) {
          if (pk->worker[i]) pk->worker[i]->wait();
          if (armed && pk->lane[i]) hipStreamSynchronize(pk->lane[i]->stream);
        }
      }
    } join_early{this};
    auto on_half = [&](size_t h) -> int {
      a_split = 1 + n_primary + h;
      l_split = h;
      for (int i = 4; i < 6; i++) ZG_HIP(hipStreamWaitEvent(lane[i]->stream, w_half, 0));
      worker[4]->submit([this, o1, a_split, lane_w, lane_mont, &early_status] {
        if (hipSetDevice(ctx->device) != hipSuccess) {
          early_status[0] = ZG_ERR_HIP;
          return;
        }
        const MsmTable t{A_len, 0, A_c, A_levels};
        early_status[0] = o1->msm_launch(lane[4], A, (const uint32_t*)lane_w, a_split, lane_mont, 0, &t);
      });
      worker[5]->submit([this, o1, l_split, lane_w, lane_mont, &early_status] {
        if (hipSetDevice(ctx->device) != hipSuccess) {
          early_status[1] = ZG_ERR_HIP;
          return;
        }
        const MsmTable t{L_len, 0, L_c, L_levels};
        early_status[1] = o1->msm_launch(lane[5], L, (const uint32_t*)(lane_w + num_inputs + 1), l_split, lane_mont, 0, &t);
      });
      return ZG_OK;
    };
    if (early)
      ZG_TRY(load_assignment(primary, n_primary, aux, n_aux, form, on_half));
    else
      ZG_TRY(load_assignment(primary, n_primary, aux, n_aux, form));
    stamp("assignment queued");
    hipStream_t st = ctx->stream;
    static const int hold = zg_knob("ZG_PROVE_HOLD") ? atoi(zg_knob("ZG_PROVE_HOLD")) : 0;
    static const int lane_wait = zg_knob("ZG_LANE_WAIT") ? atoi(zg_knob("ZG_LANE_WAIT")) : 0;
    // The B query's scalars are a gather over 1||w that only the two B lanes read: it runs on lane 0's stream (lane 3
    // waits for it through an event), not ahead of the constraint evaluation on the main stream, whose chain is the
    // proof's longest (35 us of it at JoinSplit scale).  (The debug builds' hold / host-wait modes keep it on the main stream.)
    const bool gather_on_lane = B_len && hold == 0 && lane_wait == 0;
    if (B_len && !gather_on_lane)
      hipLaunchKernelGGL((k_gather<Fr>), dim3((unsigned)((B_len + 255) / 256)), dim3(256), 0, st, lane_w, B_idx, B_len, bsc);
    // The A, B (G2 and G1) and L evaluations depend only on 1||w: they run on four side streams ("lanes", each a
    // context with its own workspace) while the main stream computes H's coefficients and then the H evaluation.
    // ZG_PROVE_HOLD = 1 | 2 | 3 (measurement knob): the lanes start only after that many parts of the H chain's
    // transforms have run (0: as soon as 1||w is there)
    int h_status = ZG_OK;
    if (hold >= 1 && hold <= 3) h_status = qap_h(0, hold - 1);
    ZG_HIP(hipEventRecord(w_ready, st));
    // (debug builds: ZG_LANE_WAIT = 1 waits for 1||w on the host instead of in the lanes' queues)
    if (lane_wait == 1) {
      ZG_HIP(hipEventSynchronize(w_ready));
    } else {
      for (int i = 0; i < 4; i++) ZG_HIP(hipStreamWaitEvent(lane[i]->stream, w_ready, 0));
    }
    if (gather_on_lane) {
      hipLaunchKernelGGL((k_gather<Fr>), dim3((unsigned)((B_len + 255) / 256)), dim3(256), 0, lane[0]->stream, lane_w, B_idx, B_len, bsc);
      ZG_HIP(hipEventRecord(bsc_ready, lane[0]->stream));
      ZG_HIP(hipStreamWaitEvent(lane[3]->stream, bsc_ready, 0));
    }
    XYZZ<Fq> At, Bt1, Ht, Lt;
    XYZZ<Fq2T> Bt2;
    // scalars are Montgomery residues in device memory
    // Witness scalars are mostly 0/1: the windows of these four evaluations are sized by the number
    // of other ("dense") scalars the same evaluation saw in the previous proof with this key -- the role libff's
    // 0/1 filter plays in multi_exp_with_mixed_addition.  Any width gives the same group element.
    // the longest dependent chain (constraint evaluation, seven transforms, H evaluation) is queued by this thread on
    // the main stream; the side lanes run at the SAME stream priority (with the lowest priority they starved behind
    // the main stream's kernels: a proof did not finish in 100 s)
    static const int h_c = zg_knob("ZG_PROVE_H_C") ? atoi(zg_knob("ZG_PROVE_H_C")) : 0;        // tuning knobs
    static const int lane_c = zg_knob("ZG_PROVE_LANE_C") ? atoi(zg_knob("ZG_PROVE_LANE_C")) : 0;
    // One bit narrower than a lone evaluation of that size would take (msm_pick_c): with five evaluations sharing the
    // device the reduction's chains count for more (JoinSplit shape, 44 K dense scalars: width 12 against 13 is 0.15-0.2 ms
    // of a proof; 11 and 14 are slower than either)
    auto width = [](uint32_t dense) {
      if (lane_c) return lane_c;
      if (!dense) return 0;
      const int c = msm_pick_c(dense < 256 ? 256 : dense);
      return c == 13 ? 12 : c;
    };
    // A proof is ~100 kernel launches and the host needs 20-35 us for each: queued from one thread, the last lane's
    // first kernel started 3.3 ms after the proof began (rocprofv3 timeline, JoinSplit scale) although its input was
    // ready after 0.05 ms.  Every lane is therefore queued by a thread of its own while this thread queues the H chain.
    // Few, long segments only while the evaluation is small: with 2^16 or more dense scalars the accumulation dwarfs the
    // reduction's instructions and its longer chain is what shows (BLS12-377, 2^20 constraints, 105 K dense scalars: 13.1 -> 12.2 ms with the
    // plan's own segment count).
    static const int lane_t = zg_knob("ZG_PROVE_LANE_SEGMENTS") ? atoi(zg_knob("ZG_PROVE_LANE_SEGMENTS")) : 128;  // tuning knobs
    static const int lane_t_g2 = zg_knob("ZG_PROVE_G2_SEGMENTS") ? atoi(zg_knob("ZG_PROVE_G2_SEGMENTS")) : 256;
    for (int i = 0; i < 4; i++) {
      const int t = i == 0 ? lane_t_g2 : lane_t;  // lane 0 runs the G2 evaluation, whose additions cost 3x
      const bool small = dense_hint[i] < (1u << 16);
      lane[i]->reduce_segments = (small && t > 0 && (t & (t - 1)) == 0) ? (uint32_t)t : 0u;
      static const int lane_ul = zg_knob("ZG_PROVE_LANE_UNIT_LEN") ? atoi(zg_knob("ZG_PROVE_LANE_UNIT_LEN")) : 0;  // tuning knob
      lane[i]->table_unit_len = lane_ul > 0 ? (uint32_t)lane_ul : 0u;
      static const int lane_tt = zg_knob("ZG_PROVE_LANE_TABLE_SEGMENTS") ? atoi(zg_knob("ZG_PROVE_LANE_TABLE_SEGMENTS")) : 0;  // tuning knob
      lane[i]->table_segments = (lane_tt > 0 && (lane_tt & (lane_tt - 1)) == 0) ? (uint32_t)lane_tt : 0u;
    }
    int lane_status[4] = {ZG_OK, ZG_OK, ZG_OK, ZG_OK};
    // (debug builds: ZG_LANE_MASK, bit i clear = lane i's evaluation is skipped -- wrong proofs, for tools/diag_prio.py)
    static const int lane_mask = zg_knob("ZG_LANE_MASK") ? atoi(zg_knob("ZG_LANE_MASK")) : 15;
    auto queue_lane = [&](int i) {
      if (hipSetDevice(ctx->device) != hipSuccess) {
        lane_status[i] = ZG_ERR_HIP;
        return;
      }
      ZG_CRUMB_SLOT(i);
      ZG_CRUMB("lane thread started");
      if (!((lane_mask >> i) & 1)) {
        lane[i]->job_n = 0;
        return;
      }
      // (a query that has a table is evaluated over it: the window width is the table's, and the hint plays no part)
      const MsmTable tabs[4] = {{B_len, 0, B2_c, B2_levels}, {A_len, a_split, A_c, A_levels}, {L_len, l_split, L_c, L_levels}, {B_len, 0, B1_c, B1_levels}};
      const MsmTable* tab = tabs[i].D ? &tabs[i] : nullptr;
      const int c = tab ? 0 : width(dense_hint[i]);
      switch (i) {
        case 0: lane_status[0] = o2->msm_launch(lane[0], B2, (const uint32_t*)bsc, B_len, lane_mont, c, tab); break;
        case 1: lane_status[1] = o1->msm_launch(lane[1], A, (const uint32_t*)(lane_w + a_split), A_len - a_split, lane_mont, c, tab); break;
        case 2: lane_status[2] = o1->msm_launch(lane[2], L, (const uint32_t*)(lane_w + num_inputs + 1 + l_split), L_len - l_split, lane_mont, c, tab); break;
        default: lane_status[3] = o1->msm_launch(lane[3], B1, (const uint32_t*)bsc, B_len, lane_mont, c, tab); break;
      }
    };
    static const bool serial_queue = zg_knob("ZG_PROVE_SERIAL_QUEUE") != nullptr;  // measurement knob: one queueing thread
    // the head of the longest chain is queued before the lanes' jobs are handed out
    const bool head_first = h_status == ZG_OK && hold == 0;
    ZG_CRUMB("prove: qap_h head");
    if (head_first) h_status = qap_h(0, 0);
    ZG_CRUMB("prove: starting lane threads");
    for (int i = 0; i < 4; i++) {
      if (serial_queue)
        queue_lane(i);
      else
        worker[i]->submit([&queue_lane, i] { queue_lane(i); });
    }
    ZG_CRUMB("prove: qap_h rest");
    if (h_status == ZG_OK && !(hold >= 3)) h_status = qap_h(hold >= 1 && hold <= 3 ? hold : (head_first ? 1 : 0), 2);
    ZG_CRUMB("prove: H msm_launch");
    const MsmTable h_tab{H_len, 0, H_c, H_levels};
    if (h_status == ZG_OK) h_status = o1->msm_launch(ctx, H, (const uint32_t*)tmp, H_len, 1, h_c, H_levels ? &h_tab : nullptr);
    ZG_CRUMB("prove: joining lane threads");
    if (!serial_queue)
      for (int i = 0; i < 4; i++) worker[i]->wait();
    for (int i = 4; i < 6; i++)
      if (worker[i]) worker[i]->wait();
    // From here on kernels of this proof are queued on five streams; they read w, bsc and tmp and write the lanes' pinned
    // buffers.  Whatever fails below, every stream is drained before the error is returned, so that the next proof on
    // this key starts on idle lanes.
    struct Drain {
      ProvingKeyDev* pk;
      bool armed = true;
      ~Drain() {
        if (!armed) return;
        hipStreamSynchronize(pk->ctx->stream);
        for (auto* l : pk->lane)
          if (l) hipStreamSynchronize(l->stream);
      }
    } drain{this};
    if (const char* inj = zg_knob("ZG_TEST_FAIL_LANE")) {  // test hook: pretend that lane's queueing failed
      const int which = atoi(inj);
      if (which >= 0 && which < 4) lane_status[which] = ZG_ERR_HIP;
      if (which == 4) h_status = ZG_ERR_HIP;
    }
    ZG_TRY(h_status);
    for (int i = 0; i < 4; i++) ZG_TRY(lane_status[i]);
    for (int i = 0; i < 2; i++) ZG_TRY(early_status[i]);
    stamp("all work queued");
    // Host tail, ordered so that everything that does not need a device result runs while the GPU is busy:
    // the four blinding multiples of delta first, then each evaluation as its lane completes.
    auto X1 = [](const Affine<Fq>& p) { return XYZZ<Fq>::from_affine(p); };
    auto X2 = [](const Affine<Fq2T>& p) { return XYZZ<Fq2T>::from_affine(p); };
    Fr rf, sf;
    memcpy(&rf, r_plain, sizeof(rf));
    memcpy(&sf, s_plain, sizeof(sf));
    const Fr rs = Fr::from_mont(Fr::mul(Fr::to_mont(rf), Fr::to_mont(sf)));
    const XYZZ<Fq> r_delta1 = xyzz_mul_limbs(X1(delta_g1), r_plain, Fr::N);
    const XYZZ<Fq> s_delta1 = xyzz_mul_limbs(X1(delta_g1), s_plain, Fr::N);
    const XYZZ<Fq> rs_delta1 = xyzz_mul_limbs(X1(delta_g1), rs.v, Fr::N);
    const XYZZ<Fq2T> s_delta2 = xyzz_mul_limbs(X2(delta_g2), s_plain, Fr::N);
    stamp("delta multiples (host)");
    ZG_TRY(o1->msm_finish(lane[1], ZG_FORM_XYZZ_INTERNAL, &At));
    if (a_split) {
      XYZZ<Fq> lo;
      ZG_TRY(o1->msm_finish(lane[4], ZG_FORM_XYZZ_INTERNAL, &lo));
      xyzz_add(At, lo);
    }
    stamp("A finished");
    XYZZ<Fq> gA = X1(alpha_g1);
    xyzz_add(gA, At);
    xyzz_add(gA, r_delta1);
    const XYZZ<Fq> s_gA = xyzz_mul_limbs(gA, s_plain, Fr::N);
    stamp("s*gA (host)");
    ZG_TRY(o1->msm_finish(lane[3], ZG_FORM_XYZZ_INTERNAL, &Bt1));
    stamp("B1 finished");
    XYZZ<Fq> g1B = X1(beta_g1);
    xyzz_add(g1B, Bt1);
    xyzz_add(g1B, s_delta1);
    const XYZZ<Fq> r_g1B = xyzz_mul_limbs(g1B, r_plain, Fr::N);
    stamp("r*g1B (host)");
    ZG_TRY(o2->msm_finish(lane[0], ZG_FORM_XYZZ_INTERNAL, &Bt2));
    stamp("B2 finished");
    XYZZ<Fq2T> gB = X2(beta_g2);
    xyzz_add(gB, Bt2);
    xyzz_add(gB, s_delta2);
    ZG_TRY(o1->msm_finish(lane[2], ZG_FORM_XYZZ_INTERNAL, &Lt));
    if (l_split) {
      XYZZ<Fq> lo;
      ZG_TRY(o1->msm_finish(lane[5], ZG_FORM_XYZZ_INTERNAL, &lo));
      xyzz_add(Lt, lo);
    }
    stamp("L finished");
    ZG_TRY(o1->msm_finish(ctx, ZG_FORM_XYZZ_INTERNAL, &Ht));
    for (int i = 0; i < 4; i++) dense_hint[i] = lane[i]->job_n ? lane[i]->job_dense + 1 : 0;
    stamp("H finished");
    XYZZ<Fq> gC = Ht;
    xyzz_add(gC, Lt);
    xyzz_add(gC, s_gA);
    xyzz_add(gC, r_g1B);
    xyzz_add(gC, xyzz_neg(rs_delta1));
    struct {
      Affine<Fq> a;
      Affine<Fq2T> b;
      Affine<Fq> c;
    } proof;
    proof.a = xyzz_to_affine(gA);
    proof.b = xyzz_to_affine(gB);
    proof.c = xyzz_to_affine(gC);
    if (out_form == ZG_FORM_PLAIN) {
      host_affine_convert(&proof.a, 1, 0);
      host_affine_convert(&proof.b, 1, 0);
      host_affine_convert(&proof.c, 1, 0);
    }
    memcpy(proof_out, &proof, sizeof(proof));
    stamp("proof normalised");
    drain.armed = false;  // every msm_finish above has synchronised its stream
    join_early.armed = false;
    return ZG_OK;
  }

  int prove(const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form, const uint32_t* r_plain,
            const uint32_t* s_plain, int out_form, void* proof_out) override {
    return prove_impl(primary, n_primary, aux, n_aux, form, r_plain, s_plain, out_form, proof_out);
  }
  int witness_map_out(const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form, int out_form,
                      void* h_out) override {
    ZG_TRY(witness_map(primary, n_primary, aux, n_aux, form));
    return read_h(out_form, h_out);
  }
  size_t domain_size() const override { return m; }
#ifdef ZG_DEBUG_KNOBS
  std::string debug_progress() const override {
    auto state = [](hipError_t e) { return e == hipSuccess ? "done" : (e == hipErrorNotReady ? "PENDING" : "error"); };
    std::string out = std::string("w_ready=") + state(hipEventQuery(w_ready)) + " main=" + state(hipStreamQuery(ctx->stream));
    auto marks = [&](const zg_ctx* c) {
      std::string m;
      for (size_t i = 0; i < c->ev_used && i < c->ev_names.size(); i++)
        m += std::string(" ") + c->ev_names[i] + ":" + state(hipEventQuery(c->ev_pool[i]));
      return m;
    };
    out += " [main marks:" + marks(ctx) + "]";
    for (int i = 0; i < 4; i++)
      out += " lane" + std::to_string(i) + "=" + state(hipStreamQuery(lane[i]->stream)) + " [" + marks(lane[i]) + "]";
    return out;
  }
#endif

  int read_h(int out_form, void* out) {
    hipStream_t st = ctx->stream;
    if (out_form == ZG_FORM_PLAIN)
      hipLaunchKernelGGL((k_fr_convert<Fr>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, tmp, m, 0);
    ZG_HIP(hipMemcpyAsync(out, tmp, m * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ZG_HIP(zg_stream_wait(st));
    memset((char*)out + m * sizeof(Fr), 0, sizeof(Fr));
    return ZG_OK;
  }
};

template <class Fq, class Fq2T, class Fr>
int pk_create_impl(zg_ctx* ctx, int curve, const zg_pk_view* view, PkBase** out) {
  auto* pk = new ProvingKeyDev<Fq, Fq2T, Fr>();
  int s = pk->create(ctx, curve, view);
  if (s != ZG_OK) {
    delete pk;
    return s;
  }
  *out = pk;
  return ZG_OK;
}

}  // namespace zg
