/* zeth_gpu.h -- C ABI of the MI355X (gfx950) Groth16 proving backend for Zeth's prover hot path.
 *
 * This is the boundary a `libzeth/gpu` adapter binds.  Plain pointers and sizes only.
 * Each entry point cites the reference interface (path:line under the clearmatics/zeth tree) it replaces.
 *
 * Element formats (all little-endian 32-bit limbs, i.e. byte-identical to libff's bigint<n> of 64-bit limbs
 * on a little-endian host):
 *   Fr element   8 limbs (32 B) on both curves
 *   Fq element   8 limbs (32 B) alt_bn128, 12 limbs (48 B) BLS12-377
 *   G1 affine    X || Y                       (64 B / 96 B)
 *   G2 affine    X.c0 || X.c1 || Y.c0 || Y.c1 (128 B / 192 B)
 *   identity     X = Y = 0 (all limbs zero)
 *   proof        g_A (G1) || g_B (G2) || g_C (G1), affine
 * `form` selects plain integers (ZG_FORM_PLAIN) or Montgomery residues x*R mod p with R = 2^(32*limbs)
 * (ZG_FORM_MONT -- libff's in-memory mont_repr).
 *
 * Every function returns 0 (ZG_OK) or a negative zg_status; zg_strerror() names it.  No function throws.
 * A context is bound to one HIP device; calls on one context must not overlap (use one context per thread).
 */
#ifndef ZETH_GPU_H
#define ZETH_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ZG_OK = 0,
  ZG_ERR_INVALID_ARGUMENT = -1, /* adapter rethrows std::invalid_argument (groth16_snark.tcc:83,98,126,146) */
  ZG_ERR_LENGTH = -2,           /* std::length_error (r1cs_variable_assignment_serialization.tcc:34-37) */
  ZG_ERR_NO_DEVICE = -3,        /* no usable HIP device: the library never falls back to the CPU */
  ZG_ERR_HIP = -4,              /* a HIP runtime call failed (std::runtime_error) */
  ZG_ERR_OUT_OF_MEMORY = -5,
  ZG_ERR_UNSUPPORTED = -6
} zg_status;

typedef enum { ZG_CURVE_ALT_BN128 = 0, ZG_CURVE_BLS12_377 = 1 } zg_curve; /* zeth_tool/tool_common.hpp:44-67 */
typedef enum { ZG_GROUP_G1 = 1, ZG_GROUP_G2 = 2 } zg_group;
typedef enum { ZG_FORM_PLAIN = 0, ZG_FORM_MONT = 1 } zg_form;
typedef enum { ZG_MEM_HOST = 0, ZG_MEM_DEVICE = 1 } zg_mem;

typedef struct zg_ctx zg_ctx;
typedef struct zg_bases zg_bases;
typedef struct zg_pk zg_pk;

const char* zg_strerror(int status);
/* sizes in bytes for a curve: Fr element, Fq element, G1 affine, G2 affine */
int zg_sizes(int curve, size_t* fr, size_t* fq, size_t* g1, size_t* g2);

/* Replaces ppT::init_public_params() + process-wide state (zeth_tool/prove_cmd.cpp:32): one context per
 * HIP device; owns a stream and a grow-only device workspace. */
int zg_ctx_create(int device, zg_ctx** out);
void zg_ctx_destroy(zg_ctx* ctx);
/* waits for everything queued on the context's stream (every entry point below already returns only when its
 * host-visible results are complete; this is for callers that time a region) */
int zg_ctx_synchronize(zg_ctx* ctx);
/* per-kernel timing of the last MSM/NTT call (HIP events on the context's stream); names[i] are static strings */
int zg_ctx_set_profiling(zg_ctx* ctx, int enabled);
int zg_ctx_last_timings(zg_ctx* ctx, const char** names, float* ms, int max_entries, int* n_entries);

/* Device buffers owned by the library (so that callers need no HIP runtime of their own to keep scalars resident
 * in HBM; no counterpart in the reference, whose vectors live in host memory). */
int zg_dev_malloc(zg_ctx* ctx, size_t bytes, void** dptr);
int zg_dev_free(zg_ctx* ctx, void* dptr);
int zg_dev_upload(zg_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int zg_dev_download(zg_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);
/* benchmark input: n Fr scalars, plain form, uniform in [0, r): element i is the first value < r of
 * the stream w(i, t) = 4 x splitmix64 words of counter (seed + i) * 2^8 + t, masked to the bit length of r,
 * t = 0, 1, ... (rejection sampling; reproducible on the host, see zeth_amd/synth.py) */
int zg_scalars_random(zg_ctx* ctx, int curve, void* dptr, size_t n, uint64_t seed);

/* Resident base points: the device-side analogue of the keypair cache (prover_server/prover_server.cpp:436-493)
 * -- the `gs` argument of libzeth::multi_exp (libzeth/core/multi_exp.hpp:13-22) uploaded once. */
int zg_bases_create(zg_ctx* ctx, int curve, int group, const void* points, size_t n, int form, int mem,
                    zg_bases** out);
/* benchmark/test input: P_i = (t0 + i*d) * G for the curve's generator G (t0, d: Fr, plain, 32 B) -- SURVEY 8d */
int zg_bases_generate(zg_ctx* ctx, int curve, int group, size_t n, const void* t0, const void* d, zg_bases** out);
/* P_i = s_i * G for n plain Fr scalars in host memory: fixed-base batch exponentiation, the operation
 * r1cs_gg_ppzksnark_generator uses to build the A/B/H/L queries (libff::batch_exp; Zeth calls the generator at
 * libzeth/snarks/groth16/groth16_snark.tcc:18-25).  Used here to fabricate real keys for tests and benchmarks. */
int zg_bases_batch_exp(zg_ctx* ctx, int curve, int group, const void* scalars, size_t n, zg_bases** out);
int zg_bases_count(const zg_bases* b, size_t* n);
/* copy `count` points starting at `first` back to the host (affine, `form`) */
int zg_bases_read(zg_bases* b, size_t first, size_t count, int form, void* out);
/* Fixed-base table.  No counterpart in the reference (libff::multi_exp recomputes from the bare bases every time);
 * it exists because the bases of a proving key never change (libsnark r1cs_gg_ppzksnark_proving_key's A/B/H/L queries,
 * loaded once at prover_server.cpp:449-493) and HBM is large: the library keeps the multiples 2^(j*c) P_i, j < D =
 * ceil((bits + 2) / c), of every base, so that an MSM sorts all n*D digit terms into ONE set of 2^(c-1) buckets and
 * needs D = 12 mixed additions per 254-bit scalar at c = 22 (13 at c = 20) instead of 16-17.  Costs D times the base memory (12.9 GB
 * for 2^24 BN254 G1 points) and one pass of D*c doublings per base.  window_bits: 0 = chosen from the number of bases,
 * else 10..22.  Results of zg_msm are unchanged (same group element); zg_bases_read still returns the bases.
 * Calling it again is a no-op (ZG_ERR_INVALID_ARGUMENT if a different width is asked for). */
int zg_bases_precompute(zg_bases* b, int window_bits);
/* *window_bits = *levels = 0 when the set has no table */
int zg_bases_table_info(const zg_bases* b, int* window_bits, int* levels);
void zg_bases_free(zg_bases* b);

/* sum_{i<n} scalars[i] * bases[first + i]  ->  one affine point.
 * Replaces libzeth::multi_exp<ppT, GroupT>(gs, fs) (libzeth/core/multi_exp.tcc:13-35, i.e.
 * libff::multi_exp_filter_one_zero<GroupT, FieldT, multi_exp_method_BDLO12>) and, inside the libsnark prover,
 * libff::multi_exp / multi_exp_with_mixed_addition.  Like the reference it requires
 * first + n <= number of bases (multi_exp.tcc:28-31 asserts gs.size() >= fs.size()) -> ZG_ERR_INVALID_ARGUMENT.
 * `scalars` are n Fr elements (host or device memory per `scalars_mem`); `out` is host memory. */
int zg_msm(zg_bases* bases, size_t first, const void* scalars, size_t n, int scalar_form, int scalars_mem,
           int out_form, void* out_affine);
/* same, with an explicit window width (0 = automatic); test and tuning hook */
int zg_msm_ex(zg_bases* bases, size_t first, const void* scalars, size_t n, int scalar_form, int scalars_mem,
              int out_form, void* out_affine, int window_bits);

/* One multi-exponentiation sharded by point range over several devices (what libff::multi_exp does with its
 * OpenMP chunks, here one chunk per GPU): shard i is a set of resident bases created on its own context (normally one
 * context per HIP device; two contexts on one device also work), scalars[i] are its counts[i] Fr scalars (host memory, or
 * memory of shard i's device, per scalars_mem) for its bases [0, counts[i]).  Every shard is driven by a host thread of
 * its own; the n_shards partial points are added on the host.  One process, N devices, no collective library needed.
 * The shards must share curve and group and sit on distinct contexts -> else ZG_ERR_INVALID_ARGUMENT. */
int zg_msm_sharded(zg_bases* const* shards, size_t n_shards, const void* const* scalars, const size_t* counts,
                   int scalar_form, int scalars_mem, int out_form, void* out_affine);

/* Self-test of the reduced-radix field arithmetic the accumulation and transform kernels use (field29.cuh), run on
 * the HOST build of the same templates against the 32-bit-limb arithmetic; needs no device.  field: 0 alt_bn128 Fq
 * (+ the mixed addition on G1), 1 BLS12-377 Fq (+ G1), 2 alt_bn128 Fr, 3 BLS12-377 Fr, 4 / 5 the Fq2 product, square
 * and G2 mixed addition of alt_bn128 / BLS12-377.  No counterpart in the
 * reference (its field arithmetic is libff's, tested there); it exists so that the CPU-only test tier covers the
 * arithmetic the GPU kernels run. */
int zg_selftest_reduced_radix(int field, int iters, uint64_t seed, int* mismatches);
/* The quad-lane point arithmetic of the combine and reduction kernels (curve_quad.cuh, curve_quad29.cuh) against the
 * one-lane formulas: curve_group 0 alt_bn128 G1, 1 BLS12-377 G1, 2 alt_bn128 G2, 3 BLS12-377 G2.  The first runs the
 * kernels' templates on the host (four threads stand in for the four lanes; no GPU needed); the second runs them on the
 * current device (G1: 32-bit limbs and the reduced radix; G2: the Fq2 instantiations).  No counterpart in the reference. */
int zg_selftest_quad_lanes(int curve_group, int* mismatches);
int zg_selftest_quad_lanes_device(int curve_group, int* mismatches);

/* Which ROCm runtime serves this process, as a JSON object in buf: the libamdhip64 this library is bound to, its version,
 * and every libamdhip64 / libhsa-runtime64 image mapped into the process.  *mixed = 1 when more than one copy of either
 * is loaded or HIP and HSA come from different installations (a host that imports another ROCm copy, e.g. a Python
 * wheel's, next to the system one): results are then only as good as that mixture -- see DESIGN.md section 7. */
int zg_runtime_info(char* buf, size_t cap, int* mixed);

/* In-place radix-2 evaluation-domain transform of 2^log_n Fr elements (natural order in and out):
 *   inverse = 0, coset = 0   domain.FFT(a)           inverse = 1, coset = 0   domain.iFFT(a)
 *   inverse = 0, coset = 1   domain.cosetFFT(a, g)    inverse = 1, coset = 1   domain.icosetFFT(a, g)
 * with g = Fr::multiplicative_generator, the only shift the prover uses.  Replaces
 * libfqfft::basic_radix2_domain<Fr>(2^log_n) as called at libzeth/core/evaluator_from_lagrange.tcc:43
 * (domain.iFFT(coefficients)) and inside libsnark's r1cs_to_qap_witness_map.
 * log_n beyond the field's two-adicity (28 / 47) or this build's limit (2^24) -> ZG_ERR_UNSUPPORTED. */
int zg_ntt(zg_ctx* ctx, int curve, void* data, unsigned log_n, int inverse, int coset, int form, int mem);

/* zg_ntt on device memory (d_data: 2^log_n Fr elements in HBM, transformed in place) for callers that cannot allocate:
 * tight serving loops and harnesses that forbid hipMalloc.  No counterpart in the reference.
 * It calls no hipMalloc / hipFree / allocating routine: the domain's twiddle and coset tables and the transform buffer
 * live in a static device array of 4,254,720 bytes that every GPU the library runs on holds until the process ends.
 * Limit: log_n <= 16 (ZG_ERR_UNSUPPORTED beyond).  Like zg_ntt it runs on the context's own stream and returns when the
 * result is in d_data (it ends in a stream synchronisation), so it is NOT a call to place inside a HIP graph capture.
 * All calls on one GPU share that array (whatever their context): the library makes them take turns (a per-GPU lock
 * held for the whole call), keeps the last call's tables and rebuilds them, with kernels only, when another
 * (curve, log_n) is asked for.  Results are those of zg_ntt. */
int zg_ntt_noalloc(zg_ctx* ctx, int curve, void* d_data, unsigned log_n, int inverse, int coset, int form);

/* One R1CS matrix in CSR form: row i (constraint i) holds the terms of its linear combination, `col` = variable
 * index with 0 = the constant ONE (libsnark linear_term.index), `val` = Fr coefficient.  The on-disk equivalent is
 * libzeth/serialization/r1cs_serialization.tcc:199-275. */
typedef struct {
  const uint64_t* row_ptr; /* num_constraints + 1 */
  const uint32_t* col;     /* nnz */
  const void* val;         /* nnz Fr elements */
} zg_r1cs_matrix;

/* Host view of r1cs_gg_ppzksnark_proving_key (libzeth/snarks/groth16/groth16_snark.hpp:18-21; fields enumerated by
 * groth16_snark.tcc:101-110): every point affine, every element in `form`. */
typedef struct {
  size_t num_inputs;      /* primary input size */
  size_t num_variables;   /* primary + auxiliary, without the constant */
  size_t num_constraints;
  zg_r1cs_matrix a, b, c; /* pk.constraint_system */
  const void *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2;
  const void* A_query; size_t A_len; /* num_variables + 1 G1 points */
  const uint32_t* B_indices;         /* B_query is a sparse knowledge_commitment_vector<G2, G1> */
  const void* B_g2; const void* B_g1; size_t B_len;
  const void* H_query; size_t H_len; /* domain size - 1 */
  const void* L_query; size_t L_len; /* num_variables - num_inputs */
  int form;
} zg_pk_view;

/* Uploads the proving key once (bases to HBM in Montgomery affine form, constraint matrices as CSR) and allocates
 * the per-proof device scratch: the analogue of loading the keypair before proving (zeth_tool/prove_cmd.cpp:39-43,
 * prover_server/prover_server.cpp:449-493).  Size mismatches -> ZG_ERR_INVALID_ARGUMENT.
 * The H query of a key whose domain has 2^18 points or more is kept as a fixed-base table (see zg_bases_precompute;
 * 12-13 times its memory; when that memory cannot be had the plain query is used); the environment variable
 * ZG_PK_TABLE_MIN sets another domain-size threshold (0 = never). */
int zg_pk_create(zg_ctx* ctx, int curve, const zg_pk_view* view, zg_pk** out);
void zg_pk_free(zg_pk* pk);
/* A second prover over the same resident key, for another calling thread: it shares the key's bases and matrices
 * in HBM and owns its per-proof scratch, its streams and a context of its own, so zg_prove on the clone may overlap
 * zg_prove on the original (and on other clones).  This is how several proofs are kept in flight on one device
 * (BASELINE configs[4]; prover_server shares one key between its handlers, prover_server.cpp:51,451).  Free every
 * clone before the key it was cloned from. */
int zg_pk_clone(zg_pk* pk, zg_pk** out);
int zg_pk_domain_size(const zg_pk* pk, size_t* m);

/* Replaces groth16_snark<ppT>::generate_proof(pk, primary_input, auxiliary_input)
 * (libzeth/snarks/groth16/groth16_snark.hpp:37-40, .tcc:36-46) = libsnark::r1cs_gg_ppzksnark_prover(pk, primary,
 * aux, force_pow_2_domain = true), with the prover randomness (r, s) passed in (Fr, plain, 32 B each) so that the
 * proof is reproducible.  proof_out receives g_A || g_B || g_C affine in out_form (256 B alt_bn128, 384 B
 * BLS12-377).  n_primary != pk.num_inputs or n_primary + n_aux != pk.num_variables -> ZG_ERR_INVALID_ARGUMENT;
 * an assignment that does not satisfy the constraints is NOT an error (it yields a proof that fails to verify,
 * as in the reference: libzeth/tests/prover/prover_test.cpp:703-714). */
int zg_prove(zg_pk* pk, const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form, const void* r,
             const void* s, int out_form, void* proof_out);
/* libsnark::r1cs_to_qap_witness_map(cs, primary, aux, 0, 0, 0, true).coefficients_for_H: writes domain size + 1
 * Fr elements to h_out (host). */
int zg_witness_map(zg_pk* pk, const void* primary, size_t n_primary, const void* aux, size_t n_aux, int form,
                   int out_form, void* h_out);

/* group-law helpers on the host tail (same code the library uses to finish an MSM / assemble a proof);
 * exported so tests can pin it without a GPU.  op: 0 = a + b, 1 = k * a (k: Fr plain 32 B in `b`) */
int zg_host_group_op(int curve, int group, int op, const void* a_affine, const void* b, int form, void* out_affine);

/* ---- group-element kernels of the SRS ceremony (host arrays in and out; set-up time, not the proving path) ---- */

/* In-place radix-2 transform over 2^log_n GROUP elements (affine, natural order): out_j = sum_i w^(i j) P_i with
 * w = omega (inverse = 0), or w = omega^-1 followed by the scaling by n^-1 (inverse = 1).  inverse = 1 is
 * compute_lagrange_from_powers = libfqfft::_basic_radix2_FFT<Fr, Gr>(powers, omega_inv) then n_inv * a
 * (libzeth/mpc/groth16/powersoftau_utils.tcc:125-134), the body of powersoftau_compute_lagrange_evaluations
 * (:830-902): from [tau^i] to [L_j(tau)]. */
int zg_group_fft(zg_ctx* ctx, int curve, int group, void* points, unsigned log_n, int inverse, int form);

/* P_i <- k * P_i for one scalar k (Fr, plain, 32 B): the delta_j^-1 update of L_g1 and H_g1 in
 * srs_mpc_phase2_update_accumulator (libzeth/mpc/groth16/phase2.tcc:447-470). */
int zg_group_scale(zg_ctx* ctx, int curve, int group, void* points, size_t n, const void* k, int form);

/* out_j = sum over the sparse row j of coeff * points[col] (CSR; coeff: Fr, plain): the per-variable evaluations
 * A_j, B_j, C_j at x from the Lagrange evaluations in mpc_compute_linearcombination
 * (libzeth/mpc/groth16/mpc_utils.tcc:187-242; rows = qap.{A,B,C}_in_Lagrange_basis[j]).  col >= n_points ->
 * ZG_ERR_INVALID_ARGUMENT. */
int zg_group_sparse_combine(zg_ctx* ctx, int curve, int group, const void* points, size_t n_points,
                            const uint64_t* row_ptr, const uint32_t* col, const void* coeff, size_t n_rows, int form,
                            void* out_points);

/* Sum of n affine points on the host: the combine step of an MSM sharded by point range, after the ranks' partial
 * results (one point each) have been all-gathered -- what libff::multi_exp does with the partial sums of its OpenMP
 * chunks. */
int zg_host_group_sum(int curve, int group, const void* points, size_t n, int form, void* out_affine);

/* libff's G::is_well_formed() over n affine points (the identity, or Y^2 = X^3 + b), on the host: what
 * groth16_snark<ppT>::proving_key_read_bytes runs on a key it has just read
 * (libzeth/snarks/groth16/groth16_snark.tcc:145-147, 193-213).  *first_bad = n when every point passes, else the
 * index of the first point off the curve. */
int zg_host_points_check(int curve, int group, const void* points, size_t n, int form, size_t* first_bad);

#ifdef __cplusplus
}
#endif
#endif /* ZETH_GPU_H */
