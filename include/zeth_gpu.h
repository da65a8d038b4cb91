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
/* per-kernel timing of the last MSM/NTT call (HIP events on the context's stream); names[i] are static strings */
int zg_ctx_set_profiling(zg_ctx* ctx, int enabled);
int zg_ctx_last_timings(zg_ctx* ctx, const char** names, float* ms, int max_entries, int* n_entries);

/* Resident base points: the device-side analogue of the keypair cache (prover_server/prover_server.cpp:436-493)
 * -- the `gs` argument of libzeth::multi_exp (libzeth/core/multi_exp.hpp:13-22) uploaded once. */
int zg_bases_create(zg_ctx* ctx, int curve, int group, const void* points, size_t n, int form, int mem,
                    zg_bases** out);
/* benchmark/test input: P_i = (t0 + i*d) * G for the curve's generator G (t0, d: Fr, plain, 32 B) -- SURVEY 8d */
int zg_bases_generate(zg_ctx* ctx, int curve, int group, size_t n, const void* t0, const void* d, zg_bases** out);
int zg_bases_count(const zg_bases* b, size_t* n);
/* copy `count` points starting at `first` back to the host (affine, `form`) */
int zg_bases_read(zg_bases* b, size_t first, size_t count, int form, void* out);
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

/* group-law helpers on the host tail (same code the library uses to finish an MSM / assemble a proof);
 * exported so tests can pin it without a GPU.  op: 0 = a + b, 1 = k * a (k: Fr plain 32 B in `b`) */
int zg_host_group_op(int curve, int group, int op, const void* a_affine, const void* b, int form, void* out_affine);

#ifdef __cplusplus
}
#endif
#endif /* ZETH_GPU_H */
