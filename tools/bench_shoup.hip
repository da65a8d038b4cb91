// Gate for the quotient-based product by a constant (f29::mul_shoup, field29.cuh) beside the Montgomery product on the
// scalar fields the transforms run in: host check of the same templates against the 32-bit-limb arithmetic, then the
// register-only issue rate of both and a device-side equality check.  One JSON line per case.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I zeth_amd/csrc tools/bench_shoup.hip -o build/bench_shoup
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "field29.cuh"

using namespace zg;

struct Clk {
  unsigned long long core, wall;
};

template <class P, int MODE>
__global__ void __launch_bounds__(256) k_chain(const Fp<P>* in, Fp<P>* out, int iters, Clk* clk) {
  using F = Fp<P>;
  const F a = in[threadIdx.x & 7], b = in[(threadIdx.x + 3) & 7];
  auto x = f29::widen<3, K29<P>::LNORM>(f29::mul(f29::unpack(a), f29::one<P>()));  // a R'
  unsigned long long c0, c1, w0, w1;
  if (MODE == 0) {
    const auto y = f29::mul(f29::unpack(b), f29::one<P>());  // b R'
    c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) x = f29::widen<3, K29<P>::LNORM>(f29::mul(x, y));
    c1 = clock64(), w1 = wall_clock64();
  } else {
    const f29::Shoup29<P> c = f29::shoup_from_plain<P>(F::from_mont(b));
    c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) x = f29::mul_shoup(x, c);
    c1 = clock64(), w1 = wall_clock64();
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = f29::pack(x);
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
}

template <class P>
static int check_host(const char* tag) {
  using F = Fp<P>;
  uint64_t x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
  int bad = 0;
  for (int t = 0; t < 20000; t++) {
    F a, b;
    for (int i = 0; i < F::N; i++) { a.v[i] = rnd(); b.v[i] = rnd(); }
    a.v[F::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;  // below p
    b.v[F::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;
    if (t == 0) a = F::zero();
    if (t == 1) b = F::zero();
    if (t == 2 || t == 3) { for (int i = 0; i < F::N; i++) a.v[i] = P::P[i]; a.v[0] -= 1; }  // p - 1
    if (t == 3 || t == 4) { for (int i = 0; i < F::N; i++) b.v[i] = P::P[i]; b.v[0] -= 1; }
    // a, b are stored (Montgomery R) forms of values A, B.  The table entry holds the PLAIN B.
    const f29::Shoup29<P> c = f29::shoup_from_plain<P>(F::from_mont(b));
    const auto ua = f29::unpack(a);                                     // A R', lazy (< 32 p or more)
    const F want = F::mul(a, b);                                       // stored form of A B
    if (!(f29::pack(f29::mul_shoup(ua, c)) == want)) bad++;
    // lazy operands, as the butterflies produce them: sums and differences of earlier results
    const auto r1 = f29::mul_shoup(ua, c);                              // A B R' < 3 p
    const auto s = f29::carry(f29::add(f29::sub(ua, r1), f29::dbl(ua)));  // (3 A - A B) R', value < 100 p
    if (!(f29::pack(f29::mul_shoup(s, c)) == F::mul(F::sub(F::add(F::dbl(a), a), want), b))) bad++;
    if (!(f29::pack(f29::mul_shoup(r1, c)) == F::mul(want, b))) bad++;
  }
  printf("{\"check\": \"%s host mul_shoup vs 32-bit limbs (canonical, p - 1, zero and lazy operands)\", \"mismatches\": %d}\n", tag, bad);
  return bad;
}

template <class P>
static int bench(const char* tag, int iters) {
  using F = Fp<P>;
  F hin[8];
  for (int i = 0; i < 8; i++) {
    hin[i] = F::one();
    hin[i].v[0] += 977u * (i + 1);
    hin[i] = F::mul(hin[i], hin[i]);
  }
  F* din;
  F *dout0, *dout1;
  Clk* dclk;
  const size_t nout = (size_t)1024 * 256;
  hipMalloc(&din, sizeof(hin));
  hipMalloc(&dout0, nout * sizeof(F));
  hipMalloc(&dout1, nout * sizeof(F));
  hipMalloc(&dclk, sizeof(Clk));
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  int bad = 0;
  for (int mode = 0; mode < 2; mode++)
    for (int wps : {1, 2, 3, 4}) {
      const int blocks = 256 * wps;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float best = 1e9f;
      for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        if (mode == 0)
          hipLaunchKernelGGL((k_chain<P, 0>), dim3(blocks), dim3(256), 0, 0, (const F*)din, dout0, iters, dclk);
        else
          hipLaunchKernelGGL((k_chain<P, 1>), dim3(blocks), dim3(256), 0, 0, (const F*)din, dout1, iters, dclk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      Clk c;
      hipMemcpy(&c, dclk, sizeof(c), hipMemcpyDeviceToHost);
      const double mhz = (double)c.core / ((double)c.wall / 100e6) / 1e6;
      const double cyc = best * 1e-3 * mhz * 1e6 * 1024.0 / ((double)blocks * 4 * iters);
      printf("{\"case\": \"%s_%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"shader_mhz\": %.0f, \"simd_cycles_per_wave_product\": %.0f}\n",
             mode ? "mul_shoup" : "mul_montgomery", tag, wps, best, mhz, cyc);
    }
  // both chains computed a b^iters: the stored forms must agree word for word
  static F h0[256], h1[256];
  hipMemcpy(h0, dout0, sizeof(h0), hipMemcpyDeviceToHost);
  hipMemcpy(h1, dout1, sizeof(h1), hipMemcpyDeviceToHost);
  for (int i = 0; i < 256; i++)
    if (!(h0[i] == h1[i])) bad++;
  printf("{\"check\": \"%s device: %d-product chains, mul_shoup vs Montgomery\", \"mismatches\": %d}\n", tag, iters, bad);
  return bad;
}

int main(int argc, char** argv) {
  int bad = check_host<Bn254FrParams>("bn254_fr") + check_host<Bls377FrParams>("bls377_fr");
  if (argc > 1 && !strcmp(argv[1], "--host-only")) return bad != 0;
  bad += bench<Bn254FrParams>("bn254_fr", 4096);
  bad += bench<Bls377FrParams>("bls377_fr", 4096);
  return bad != 0;
}
