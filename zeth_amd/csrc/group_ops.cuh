// Group-element kernels of the SRS ceremony, generic over the group:
//   group_fft      libfqfft::_basic_radix2_FFT<Fr, Gr>(powers, omega_inv) followed by the n^-1 scaling =
//                  compute_lagrange_from_powers (libzeth/mpc/groth16/powersoftau_utils.tcc:125-134), the kernel of
//                  powersoftau_compute_lagrange_evaluations (:830-902)
//   group_scale    P_i <- k P_i, the delta^-1 update of L and H in srs_mpc_phase2_update_accumulator
//                  (libzeth/mpc/groth16/phase2.tcc:447-470)
//   group_combine  out_j = sum_t coeff_t * P[col_t] over the sparse row j: the A_j / B_j / C_j evaluations of
//                  mpc_compute_linearcombination (libzeth/mpc/groth16/mpc_utils.tcc:187-242)
// Every output is normalised to its unique affine point, so the results do not depend on the order of operations.
// These are set-up operations (once per ceremony step): one lane per butterfly / point / row with double-and-add
// scalar multiplications; the points stay in XYZZ form in HBM between the stages of the transform.
#pragma once
#include "msm_host.cuh"
#include "ntt_host.cuh"

namespace zg {

template <class F>
__global__ void __launch_bounds__(64) k_gfft_load(const Affine<F>* __restrict__ in, XYZZ<F>* __restrict__ a, unsigned log_n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  // bit-reversed placement (decimation in time)
  size_t r = 0;
  for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
  a[r] = XYZZ<F>::from_affine(load_affine(in, i));
}

// stage s (butterfly span m = 2^s): a[k+j], a[k+j+m] <- a[k+j] +- w^(j n / 2m) a[k+j+m]
template <class F, class Fr>
__global__ void __launch_bounds__(64) k_gfft_stage(XYZZ<F>* __restrict__ a, const Fr* __restrict__ tw_mont, unsigned log_n,
                                                    unsigned s) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> (log_n - 1)) return;
  const size_t m = (size_t)1 << s;
  const size_t j = i & (m - 1);
  const size_t i0 = ((i >> s) << (s + 1)) + j, i1 = i0 + m;
  XYZZ<F> t = a[i1];
  if (j != 0) {
    const Fr w = Fr::from_mont(tw_mont[j << (log_n - 1 - s)]);
    uint32_t k[Fr::N];
#pragma unroll
    for (int q = 0; q < Fr::N; q++) k[q] = w.v[q];
    t = xyzz_mul_limbs(t, k, Fr::N);
  }
  XYZZ<F> lo = a[i0], hi = lo;
  xyzz_add(lo, t);
  xyzz_add(hi, xyzz_neg(t));
  a[i0] = lo;
  a[i1] = hi;
}

template <class F, class Fr>
__global__ void __launch_bounds__(64) k_group_scale(XYZZ<F>* __restrict__ a, size_t n, Fr k_plain) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[Fr::N];
#pragma unroll
  for (int q = 0; q < Fr::N; q++) k[q] = k_plain.v[q];
  a[i] = xyzz_mul_limbs(a[i], k, Fr::N);
}

template <class F>
__global__ void __launch_bounds__(64) k_group_lift(const Affine<F>* __restrict__ in, XYZZ<F>* __restrict__ a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  a[i] = XYZZ<F>::from_affine(load_affine(in, i));
}

// one lane per sparse row; coefficients are plain Fr; +1 and -1 (the bulk of an R1CS) skip the multiplication
template <class F, class Fr>
__global__ void __launch_bounds__(64) k_group_combine(const Affine<F>* __restrict__ pts, const uint64_t* __restrict__ row_ptr,
                                                       const uint32_t* __restrict__ col, const Fr* __restrict__ coeff_plain,
                                                       size_t n_rows, Fr minus_one_plain, XYZZ<F>* __restrict__ out) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint64_t t = row_ptr[r]; t < row_ptr[r + 1]; t++) {
    const Fr c = coeff_plain[t];
    Affine<F> p = load_affine(pts, col[t]);
    if (c == Fr::raw_one()) {
      xyzz_madd(acc, p);
    } else if (c == minus_one_plain) {
      p.y = F::neg(p.y);
      xyzz_madd(acc, p);
    } else {
      uint32_t k[Fr::N];
#pragma unroll
      for (int q = 0; q < Fr::N; q++) k[q] = c.v[q];
      xyzz_add(acc, xyzz_mul_limbs(XYZZ<F>::from_affine(p), k, Fr::N));
    }
  }
  out[r] = acc;
}

// XYZZ array on the device -> affine array (device), through the batched normalisation of msm_host.cuh
template <class F>
int group_normalize(zg_ctx* ctx, const XYZZ<F>* d_xyzz, size_t n, Affine<F>* d_out) {
  F* pref = nullptr;
  ZG_HIP(hipMalloc((void**)&pref, (n ? n : 1) * sizeof(F)));
  const size_t lanes = (n + GEN_CHUNK - 1) / GEN_CHUNK;
  hipLaunchKernelGGL((k_normalize<F>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, ctx->stream, d_xyzz, pref, n, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(pref);
  return e == hipSuccess ? ZG_OK : ZG_ERR_HIP;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) hipFree(p);
  }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e == hipSuccess) return ZG_OK;
    p = nullptr;
    return e == hipErrorOutOfMemory ? ZG_ERR_OUT_OF_MEMORY : ZG_ERR_HIP;
  }
};

// points: n = 2^log_n affine points in HOST memory, overwritten with the transform
template <class F, class Fr>
int group_fft(zg_ctx* ctx, int curve, void* points, unsigned log_n, int inverse, int form) {
  using P = typename Fr::Params;
  if (log_n > (unsigned)P::TWO_ADICITY || log_n > 26) return ZG_ERR_UNSUPPORTED;
  const size_t n = (size_t)1 << log_n;
  hipStream_t st = ctx->stream;
  DevBuf aff, xy;
  ZG_TRY(aff.alloc(n * sizeof(Affine<F>)));
  ZG_TRY(xy.alloc(n * sizeof(XYZZ<F>)));
  Affine<F>* d_aff = reinterpret_cast<Affine<F>*>(aff.p);
  XYZZ<F>* d_xy = reinterpret_cast<XYZZ<F>*>(xy.p);
  ZG_HIP(hipMemcpyAsync(d_aff, points, n * sizeof(Affine<F>), hipMemcpyHostToDevice, st));
  if (form == ZG_FORM_PLAIN) ZG_TRY((bases_to_mont<F>(ctx, d_aff, n)));
  const unsigned blocks = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL((k_gfft_load<F>), dim3(blocks), dim3(64), 0, st, d_aff, d_xy, log_n);
  if (log_n > 0) {
    NttDomain<Fr>* D = nullptr;
    ZG_TRY(ntt_get_domain<Fr>(ctx, curve, log_n, &D));
    const Fr* tw = inverse ? D->T.tw_inv : D->T.tw;
    const unsigned hb = (unsigned)((n / 2 + 63) / 64);
    for (unsigned s = 0; s < log_n; s++)
      hipLaunchKernelGGL((k_gfft_stage<F, Fr>), dim3(hb), dim3(64), 0, st, d_xy, tw, log_n, s);
    if (inverse) hipLaunchKernelGGL((k_group_scale<F, Fr>), dim3(blocks), dim3(64), 0, st, d_xy, n, Fr::from_mont(D->T.m_inv));
  }
  ZG_HIP(hipGetLastError());
  ZG_TRY(group_normalize<F>(ctx, d_xy, n, d_aff));
  ZG_HIP(hipMemcpy(points, d_aff, n * sizeof(Affine<F>), hipMemcpyDeviceToHost));
  if (form == ZG_FORM_PLAIN) host_affine_convert(reinterpret_cast<Affine<F>*>(points), n, 0);
  return ZG_OK;
}

template <class F, class Fr>
int group_scale(zg_ctx* ctx, void* points, size_t n, const uint32_t* k_plain, int form) {
  if (n == 0) return ZG_OK;
  hipStream_t st = ctx->stream;
  DevBuf aff, xy;
  ZG_TRY(aff.alloc(n * sizeof(Affine<F>)));
  ZG_TRY(xy.alloc(n * sizeof(XYZZ<F>)));
  Affine<F>* d_aff = reinterpret_cast<Affine<F>*>(aff.p);
  XYZZ<F>* d_xy = reinterpret_cast<XYZZ<F>*>(xy.p);
  ZG_HIP(hipMemcpyAsync(d_aff, points, n * sizeof(Affine<F>), hipMemcpyHostToDevice, st));
  if (form == ZG_FORM_PLAIN) ZG_TRY((bases_to_mont<F>(ctx, d_aff, n)));
  Fr k;
  memcpy(&k, k_plain, sizeof(k));
  const unsigned blocks = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL((k_group_lift<F>), dim3(blocks), dim3(64), 0, st, d_aff, d_xy, n);
  hipLaunchKernelGGL((k_group_scale<F, Fr>), dim3(blocks), dim3(64), 0, st, d_xy, n, k);
  ZG_HIP(hipGetLastError());
  ZG_TRY(group_normalize<F>(ctx, d_xy, n, d_aff));
  ZG_HIP(hipMemcpy(points, d_aff, n * sizeof(Affine<F>), hipMemcpyDeviceToHost));
  if (form == ZG_FORM_PLAIN) host_affine_convert(reinterpret_cast<Affine<F>*>(points), n, 0);
  return ZG_OK;
}

template <class F, class Fr>
int group_combine(zg_ctx* ctx, const void* points, size_t n_points, const uint64_t* row_ptr, const uint32_t* col,
                  const void* coeff_plain, size_t n_rows, int form, void* out_points) {
  if (n_rows == 0) return ZG_OK;
  if (!row_ptr || row_ptr[0] != 0) return ZG_ERR_INVALID_ARGUMENT;
  const size_t nnz = (size_t)row_ptr[n_rows];
  for (size_t r = 0; r < n_rows; r++)
    if (row_ptr[r] > row_ptr[r + 1]) return ZG_ERR_INVALID_ARGUMENT;
  for (size_t t = 0; t < nnz; t++)
    if (col[t] >= n_points) return ZG_ERR_INVALID_ARGUMENT;
  hipStream_t st = ctx->stream;
  DevBuf aff, xy, rp, cl, cf, oaff;
  ZG_TRY(aff.alloc(n_points * sizeof(Affine<F>)));
  ZG_TRY(xy.alloc(n_rows * sizeof(XYZZ<F>)));
  ZG_TRY(oaff.alloc(n_rows * sizeof(Affine<F>)));
  ZG_TRY(rp.alloc((n_rows + 1) * 8));
  ZG_TRY(cl.alloc(nnz * 4));
  ZG_TRY(cf.alloc(nnz * sizeof(Fr)));
  Affine<F>* d_aff = reinterpret_cast<Affine<F>*>(aff.p);
  ZG_HIP(hipMemcpyAsync(d_aff, points, n_points * sizeof(Affine<F>), hipMemcpyHostToDevice, st));
  if (form == ZG_FORM_PLAIN && n_points) ZG_TRY((bases_to_mont<F>(ctx, d_aff, n_points)));
  ZG_HIP(hipMemcpyAsync(rp.p, row_ptr, (n_rows + 1) * 8, hipMemcpyHostToDevice, st));
  if (nnz) {
    ZG_HIP(hipMemcpyAsync(cl.p, col, nnz * 4, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(cf.p, coeff_plain, nnz * sizeof(Fr), hipMemcpyHostToDevice, st));
  }
  Fr minus_one;  // r - 1, plain
  for (int i = 0; i < Fr::N; i++) minus_one.v[i] = Fr::Params::P[i];
  minus_one.v[0] -= 1;  // r is odd: no borrow
  hipLaunchKernelGGL((k_group_combine<F, Fr>), dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, st, d_aff,
                     (const uint64_t*)rp.p, (const uint32_t*)cl.p, (const Fr*)cf.p, n_rows, minus_one,
                     reinterpret_cast<XYZZ<F>*>(xy.p));
  ZG_HIP(hipGetLastError());
  ZG_TRY(group_normalize<F>(ctx, reinterpret_cast<XYZZ<F>*>(xy.p), n_rows, reinterpret_cast<Affine<F>*>(oaff.p)));
  ZG_HIP(hipMemcpy(out_points, oaff.p, n_rows * sizeof(Affine<F>), hipMemcpyDeviceToHost));
  if (form == ZG_FORM_PLAIN) host_affine_convert(reinterpret_cast<Affine<F>*>(out_points), n_rows, 0);
  return ZG_OK;
}

}  // namespace zg
