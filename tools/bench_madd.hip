// Register-only issue-rate benchmark for the field multiplier and the XYZZ mixed addition (no memory traffic in the
// loops): separates "what the VALU can issue" from what k_accum achieves with its gathers.  Prints one JSON line per
// case with cycles per wave-level operation (at the shader clock measured alongside with clock64/wall_clock64).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I zeth_amd/csrc tools/bench_madd.hip -o build/bench_madd
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "curve.cuh"

using namespace zg;
using F = Bn254Fq;

struct Clk {
  unsigned long long core, wall;
};

template <int CHAINS>
__global__ void __launch_bounds__(256) k_mul(const F* in, F* out, int iters, Clk* clk) {
  F a[CHAINS], b = in[threadIdx.x & 7];
  for (int c = 0; c < CHAINS; c++) a[c] = in[(threadIdx.x + c) & 7];
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = F::mul(a[c], b);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  F r = a[0];
  for (int c = 1; c < CHAINS; c++) r = F::add(r, a[c]);
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
}

template <int OP>
__global__ void __launch_bounds__(256) k_lin(const F* in, F* out, int iters, Clk* clk) {
  F a = in[threadIdx.x & 7], b = in[(threadIdx.x + 3) & 7];
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (OP == 0) a = F::sub(a, b);
      if (OP == 1) a = F::add(a, b);
      if (OP == 2) a = F::cneg(a, ((i + k) & 1) != 0);
      if (OP == 3) a = F::sqr(a);
    }
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
}

// the mixed addition without its exceptional-case branches (straight-line general case)
__global__ void __launch_bounds__(256) k_madd_straight(const Affine<F>* in, XYZZ<F>* out, int iters, Clk* clk) {
  Affine<F> p = in[threadIdx.x & 7];
  XYZZ<F> acc = XYZZ<F>::from_affine(in[(threadIdx.x + 1) & 7]);
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    F U2 = F::mul(p.x, acc.zz);
    F S2 = F::mul(p.y, acc.zzz);
    F Pd = F::sub(U2, acc.x);
    F R = F::sub(S2, acc.y);
    F PP = F::sqr(Pd);
    F PPP = F::mul(Pd, PP);
    F Q = F::mul(acc.x, PP);
    F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    F Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
}

__global__ void __launch_bounds__(256) k_madd(const Affine<F>* in, XYZZ<F>* out, int iters, Clk* clk) {
  Affine<F> p = in[threadIdx.x & 7];
  XYZZ<F> acc = XYZZ<F>::from_affine(in[(threadIdx.x + 1) & 7]);
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    p.y = F::cneg(p.y, (i & 1) != 0);
    xyzz_madd(acc, p);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
}

template <class K, class... A>
static void run(const char* name, K kern, int blocks, int iters, double ops_per_iter, Clk* d_clk, A... args) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args..., iters, d_clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  Clk c;
  hipMemcpy(&c, d_clk, sizeof(c), hipMemcpyDeviceToHost);
  double mhz = (double)c.core / ((double)c.wall / 100e6) / 1e6;  // wall_clock64 ticks at 100 MHz
  double waves_per_simd = blocks * 4.0 / 1024.0;
  double wave_ops = (double)blocks * 4 * iters * ops_per_iter;
  double cyc = best * 1e-3 * mhz * 1e6 * 1024.0 / wave_ops;  // SIMD-cycles per wave-level operation
  printf("{\"case\": \"%s\", \"blocks\": %d, \"waves_per_simd\": %.2f, \"ms\": %.3f, \"shader_mhz\": %.0f, "
         "\"simd_cycles_per_wave_op\": %.0f}\n", name, blocks, waves_per_simd, best, mhz, cyc);
}

int main() {
  F hin[8];
  for (int i = 0; i < 8; i++) {
    hin[i] = F::one();
    hin[i].v[0] += 977u * (i + 1);
  }
  Affine<F> hp[8];
  for (int i = 0; i < 8; i++) {
    for (int k = 0; k < F::N; k++) {
      hp[i].x.v[k] = Bn254G1Params::GX_MONT[k];
      hp[i].y.v[k] = Bn254G1Params::GY_MONT[k];
    }
  }
  // distinct points: k * G, so that the addition never degenerates into the doubling branch
  for (int i = 1; i < 8; i++) {
    XYZZ<F> t = XYZZ<F>::from_affine(hp[i - 1]);
    xyzz_madd(t, hp[0]);
    if (i == 1) t = xyzz_dbl(XYZZ<F>::from_affine(hp[0]));
    hp[i] = xyzz_to_affine(t);
  }
  F* din;
  Affine<F>* dp;
  void* dout;
  Clk* dclk;
  const int maxblocks = 768 * 4;
  hipMalloc(&din, sizeof(hin));
  hipMalloc(&dp, sizeof(hp));
  hipMalloc(&dout, (size_t)maxblocks * 256 * sizeof(XYZZ<F>));
  hipMalloc(&dclk, sizeof(Clk));
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice);
  for (int wps : {1, 2, 3, 4, 6, 8}) run("mul_chain1", k_mul<1>, 256 * wps, 4096, 1.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 2, 3, 4}) run("mul_chain2", k_mul<2>, 256 * wps, 2048, 2.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 2, 3, 6}) run("madd", k_madd, 256 * wps, 512, 1.0, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  for (int wps : {1, 3}) run("madd_straight", k_madd_straight, 256 * wps, 512, 1.0, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  for (int wps : {1, 3}) run("sub", k_lin<0>, 256 * wps, 2048, 8.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 3}) run("add", k_lin<1>, 256 * wps, 2048, 8.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 3}) run("cneg", k_lin<2>, 256 * wps, 2048, 8.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 3}) run("sqr", k_lin<3>, 256 * wps, 512, 8.0, dclk, (const F*)din, (F*)dout);
  return 0;
}
