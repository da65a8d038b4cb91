// Register-only issue-rate benchmark for the field multiplier and the XYZZ mixed addition (no memory traffic in the
// loops): separates "what the VALU can issue" from what k_accum achieves with its gathers.  Prints one JSON line per
// case with cycles per wave-level operation (at the shader clock measured alongside with clock64/wall_clock64).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I zeth_amd/csrc tools/bench_madd.hip -o build/bench_madd
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include <algorithm>

#include "msm.cuh"

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

// the mixed addition without its exceptional-case branches (straight-line general case).  DESYNC: every wave first
// runs a different number of products so that the waves of a CU sit at different places of the loop body (as they do
// in k_accum, where memory latency spreads them out) instead of fetching the same instructions together.
template <int DESYNC>
__global__ void __launch_bounds__(256) k_madd_straight(const Affine<F>* in, XYZZ<F>* out, int iters, Clk* clk) {
  Affine<F> p = in[threadIdx.x & 7];
  XYZZ<F> acc = XYZZ<F>::from_affine(in[(threadIdx.x + 1) & 7]);
  if (DESYNC) {
    const int skew = (int)((blockIdx.x * 4 + threadIdx.x / 64) * 2654435761u >> 24) % 53;
    for (int i = 0; i < skew; i++) p.x = F::mul(p.x, p.y);
  }
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

// k_accum itself on fabricated inputs: every unit `len` entries long, entries uniformly random over `nbases` points
// mode 0: every unit `len` long.  mode 1: lengths len +- sqrt(len) (Gaussian, like bucket loads), in arrival order.
// mode 2: the same lengths handed out longest first (what k_units' order does).
static void run_accum(const char* name, int blocks, uint32_t len, uint32_t nbases, const Affine<F>* dp8, int mode = 0, int extra_blocks = 0) {
  const size_t lanes = (size_t)blocks * 256;
  std::vector<UnitDesc> hu(lanes);
  std::vector<uint32_t> hord(lanes), hw(2);
  uint64_t x = 88172645463325252ull;
  auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  uint64_t total = 0;
  for (size_t i = 0; i < lanes; i++) {
    uint32_t l = len;
    if (mode) {
      double g = 0;
      for (int k = 0; k < 12; k++) g += (double)(rnd() & 0xFFFFFF) / 16777216.0;
      l = (uint32_t)((double)len + (g - 6.0) * sqrt((double)len));
      if (l < 1) l = 1;
    }
    hu[i] = UnitDesc{total, l, (uint32_t)i};
    total += l;
    hord[i] = (uint32_t)i;
  }
  if (mode == 2) std::sort(hord.begin(), hord.end(), [&](uint32_t a, uint32_t b) { return hu[a].len > hu[b].len; });
  std::vector<uint32_t> hsorted(total);
  for (auto& e : hsorted) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    e = (uint32_t)(x % nbases) | ((x >> 40) & 1 ? 0x80000000u : 0u);
  }
  hw[0] = (uint32_t)lanes;
  std::vector<Affine<F>> hb(nbases);
  Affine<F> h8[8];
  hipMemcpy(h8, dp8, sizeof(h8), hipMemcpyDeviceToHost);
  {
    // distinct points (i + 1) G for the first 4096, repeated with a stride that keeps neighbours distinct
    const uint32_t distinct = nbases < 4096 ? nbases : 4096;
    XYZZ<F> t = XYZZ<F>::from_affine(h8[0]);
    for (uint32_t i = 0; i < distinct; i++) {
      hb[i] = xyzz_to_affine(t);
      xyzz_madd(t, h8[0]);
    }
    for (uint32_t i = distinct; i < nbases; i++) hb[i] = hb[(i * 2654435761u) % distinct];
  }
  UnitDesc* du; uint32_t *dord, *dsorted, *dw; Affine<F>* db; XYZZ<F>* dpart;
  hipMalloc(&du, hu.size() * sizeof(UnitDesc)); hipMalloc(&dord, lanes * 4); hipMalloc(&dsorted, hsorted.size() * 4);
  hipMalloc(&dw, 8); hipMalloc(&db, hb.size() * sizeof(Affine<F>)); hipMalloc(&dpart, lanes * sizeof(XYZZ<F>));
  hipMemcpy(du, hu.data(), hu.size() * sizeof(UnitDesc), hipMemcpyHostToDevice);
  hipMemcpy(dord, hord.data(), lanes * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsorted, hsorted.data(), hsorted.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), 8, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * sizeof(Affine<F>), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_accum<F>), dim3((unsigned)(lanes / ZG_ACCUM_BLOCK) + extra_blocks, 1), dim3(ZG_ACCUM_BLOCK), 0, 0, du, dw, (uint32_t)lanes, dord, dsorted, db, dpart, 0x7FFFFFFFu, 1u, 0u);
    if (hipGetLastError() != hipSuccess) printf("{\"error\": \"k_accum launch failed\"}\n");
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double wave_ops = (double)total / 64.0;
  printf("{\"case\": \"%s\", \"mode\": %d, \"blocks\": %d, \"len\": %u, \"nbases\": %u, \"ms\": %.3f, \"simd_cycles_per_wave_op_at_2.4GHz\": %.0f}\n",
         name, mode, blocks, len, nbases, best, best * 1e-3 * 2.4e9 * 1024.0 / wave_ops);
  hipFree(du); hipFree(dord); hipFree(dsorted); hipFree(dw); hipFree(db); hipFree(dpart);
}

int main(int argc, char** argv) {
  const bool accum_only = argc > 1;
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
  if (!accum_only) {
  for (int wps : {1, 2, 3, 4, 6, 8}) run("mul_chain1", k_mul<1>, 256 * wps, 4096, 1.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 2, 3, 4}) run("mul_chain2", k_mul<2>, 256 * wps, 2048, 2.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 2, 3, 6}) run("madd", k_madd, 256 * wps, 512, 1.0, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  for (int wps : {1, 3}) run("madd_straight", k_madd_straight<0>, 256 * wps, 512, 1.0, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  // sustained load (~100 ms per launch): does the shader clock hold?
  run("madd_straight_desync", k_madd_straight<1>, 512, 512 * 16, 1.0, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  run("madd_straight_long", k_madd_straight<0>, 512, 512 * 8, 1.0, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  }
  run_accum("k_accum_synth", 512, 512, 1u << 12, dp);
  run_accum("k_accum_synth", 2048, 512, 1u << 12, dp);
  run_accum("k_accum_synth", 2048, 512, 1u << 22, dp);
  run_accum("k_accum_synth", 2048, 512, 1u << 22, dp, 1);
  run_accum("k_accum_synth", 2048, 512, 1u << 22, dp, 2);
  run_accum("k_accum_synth", 2048, 512, 1u << 24, dp, 2);
  run_accum("k_accum_synth_1040_empty_blocks", 2048, 512, 1u << 22, dp, 2, 1040);
  run_accum("k_accum_synth", 2048, 64, 1u << 21, dp, 1);
  run_accum("k_accum_synth", 2048, 64, 1u << 21, dp, 2);
  if (!accum_only) {
  for (int wps : {1, 3}) run("sub", k_lin<0>, 256 * wps, 2048, 8.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 3}) run("add", k_lin<1>, 256 * wps, 2048, 8.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 3}) run("cneg", k_lin<2>, 256 * wps, 2048, 8.0, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 3}) run("sqr", k_lin<3>, 256 * wps, 512, 8.0, dclk, (const F*)din, (F*)dout);
  }
  return 0;
}
