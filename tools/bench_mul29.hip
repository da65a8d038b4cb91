// Register-only issue-rate benchmark of the reduced-radix multiplier (field29.cuh) beside the 32-bit-limb one
// (mont_asm.h), plus a device-side check that both give the same field element.  One JSON line per case.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I zeth_amd/csrc tools/bench_mul29.hip -o build/bench_mul29
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "accum29.cuh"
#include "field29.cuh"

using namespace zg;

struct Clk {
  unsigned long long core, wall;
};

template <class P, int MODE>
__global__ void __launch_bounds__(256) k_chain(const Fp<P>* in, Fp<P>* out, int iters, Clk* clk) {
  using F = Fp<P>;
  F a = in[threadIdx.x & 7], b = in[(threadIdx.x + 3) & 7];
  unsigned long long c0 = 0, c1 = 0, w0 = 0, w1 = 0;
  if (MODE == 0) {  // 32-bit limbs
    c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) a = F::mul(a, b);
    c1 = clock64(), w1 = wall_clock64();
  } else {
    auto x = f29::mul(f29::unpack(a), f29::one<P>());
    const auto y = f29::mul(f29::unpack(b), f29::one<P>());
    c0 = clock64(), w0 = wall_clock64();
    if (MODE == 1)
      for (int i = 0; i < iters; i++) x = f29::mul(x, y);
    if (MODE == 2)
      for (int i = 0; i < iters; i++) x = f29::sqr(x);
    if (MODE == 3)
      for (int i = 0; i < iters; i++) x = f29::mul(f29::carry(f29::sub(x, y)), y);
    c1 = clock64(), w1 = wall_clock64();
    a = f29::pack(x);
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
}

// the straight-line mixed addition in both arithmetics
template <class P, int MODE>
__global__ void __launch_bounds__(256) k_madd_chain(const Affine<Fp<P>>* in, XYZZ<Fp<P>>* out, int iters, Clk* clk) {
  using F = Fp<P>;
  Affine<F> p = in[threadIdx.x & 7];
  XYZZ<F> acc = XYZZ<F>::from_affine(in[(threadIdx.x + 1) & 7]);
  unsigned long long c0, c1, w0, w1;
  if (MODE == 0) {
    c0 = clock64(), w0 = wall_clock64();
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
    c1 = clock64(), w1 = wall_clock64();
  } else {
    Acc29<P> a29 = acc29_from_xyzz<P>(acc);
    c0 = clock64(), w0 = wall_clock64();
    if (MODE == 1)
      for (int i = 0; i < iters; i++) {
        bool bad = false;
        a29 = madd29_fast<P>(a29, p, (i & 1) != 0, &bad);
      }
    else
      for (int i = 0; i < iters; i++) a29 = madd29<P>(a29, p, (i & 1) != 0);
    c1 = clock64(), w1 = wall_clock64();
    acc = acc29_to_xyzz<P>(a29);
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = Clk{c1 - c0, w1 - w0};
}

template <class K, class... A>
static void run(const char* name, K kern, int blocks, int iters, Clk* d_clk, A... args) {
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
  double mhz = (double)c.core / ((double)c.wall / 100e6) / 1e6;
  double wave_ops = (double)blocks * 4 * iters;
  double cyc = best * 1e-3 * mhz * 1e6 * 1024.0 / wave_ops;
  printf("{\"case\": \"%s\", \"blocks\": %d, \"waves_per_simd\": %.2f, \"ms\": %.3f, \"shader_mhz\": %.0f, \"simd_cycles_per_wave_op\": %.0f}\n",
         name, blocks, blocks * 4.0 / 1024.0, best, mhz, cyc);
}

template <class P>
static int check_mul(const char* tag) {
  using F = Fp<P>;
  // host: the same code paths compiled for the CPU, against the 32-bit-limb arithmetic.  unpack(x R) = x R' (lazy),
  // products stay in the R' form, pack(x R') = x R.
  uint64_t x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
  int bad = 0;
  for (int t = 0; t < 20000; t++) {
    F a, b;
    for (int i = 0; i < F::N; i++) { a.v[i] = rnd(); b.v[i] = rnd(); }
    a.v[F::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;  // below p
    b.v[F::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;
    if (t == 0) a = F::zero();
    if (t == 1) { for (int i = 0; i < F::N; i++) a.v[i] = P::P[i]; a.v[0] -= 1; }  // p - 1
    if (t == 2) { for (int i = 0; i < F::N; i++) a.v[i] = b.v[i] = P::P[i]; a.v[0] -= 1; b.v[0] -= 1; }
    const auto ua = f29::unpack(a), ub = f29::unpack(b);
    if (!(f29::pack(ua) == a)) bad++;
    const auto ra = f29::mul(ua, f29::one<P>()), rb = f29::mul(ub, f29::one<P>());  // below 2 p
    if (!(f29::pack(ra) == a)) bad++;
    if (!(f29::pack(f29::mul(ra, ub)) == F::mul(a, b))) bad++;
    if (!(f29::pack(f29::sqr(ra)) == F::mul(a, a))) bad++;
    const auto ab = f29::mul(ra, ub), bb = f29::sqr(rb);
    if (!(f29::pack(f29::carry(f29::sub(ab, bb))) == F::sub(F::mul(a, b), F::mul(b, b)))) bad++;
    if (!(f29::pack(f29::add(ab, f29::dbl(bb))) == F::add(F::mul(a, b), F::dbl(F::mul(b, b))))) bad++;
    if (!(f29::pack(f29::mul(f29::carry(f29::sub(ab, bb)), f29::sub(bb, ab))) ==
          F::mul(F::sub(F::mul(a, b), F::mul(b, b)), F::sub(F::mul(b, b), F::mul(a, b)))))
      bad++;
    if (!(f29::pack(f29::neg(ab)) == F::neg(F::mul(a, b)))) bad++;
    if (f29::is_zero_mod_p(f29::mul(ua, f29::one<P>())) != a.is_zero()) bad++;
  }
  printf("{\"check\": \"%s host unpack/pack/mul/sqr/add/sub/neg vs 32-bit limbs\", \"mismatches\": %d}\n", tag, bad);
  return bad;
}

template <class P, class GP>
static int check_madd(const char* tag) {
  using F = Fp<P>;
  Affine<F> g;
  for (int k = 0; k < F::N; k++) {
    g.x.v[k] = GP::GX_MONT[k];
    g.y.v[k] = GP::GY_MONT[k];
  }
  int bad = 0;
  XYZZ<F> ref = XYZZ<F>::inf();
  Acc29<P> a = acc29_inf<P>();
  Affine<F> pts[6];
  XYZZ<F> t = XYZZ<F>::from_affine(g);
  for (int i = 0; i < 6; i++) {
    pts[i] = xyzz_to_affine(t);
    t = xyzz_dbl(t);
    xyzz_madd(t, g);
  }
  // sequence with every exceptional case: identity base, first point, doubling, cancellation, restart
  const int seq[] = {-1, 0, 1, 2, -1, 3, 4, 5, 0, 1};
  const int neg[] = {0, 0, 0, 1, 0, 0, 1, 0, 1, 1};
  for (int s = 0; s < 10; s++) {
    Affine<F> p = seq[s] < 0 ? Affine<F>::inf() : pts[seq[s]];
    Affine<F> ps = p;
    ps.y = F::cneg(p.y, neg[s] != 0);
    xyzz_madd(ref, ps);
    a = madd29<P>(a, p, neg[s] != 0);
    if (!(xyzz_to_affine(acc29_to_xyzz<P>(a)).x == xyzz_to_affine(ref).x) || !(xyzz_to_affine(acc29_to_xyzz<P>(a)).y == xyzz_to_affine(ref).y)) bad++;
  }
  // doubling: P + P, cancellation: P - P, then on from the identity
  for (int rep = 0; rep < 2; rep++) {
    XYZZ<F> r2 = XYZZ<F>::from_affine(pts[2]);
    Acc29<P> a2 = acc29_from_affine<P>(pts[2], false);
    Affine<F> q = pts[2];
    q.y = F::cneg(q.y, rep != 0);
    xyzz_madd(r2, q);
    a2 = madd29<P>(a2, pts[2], rep != 0);
    if (acc29_to_xyzz<P>(a2).is_inf() != r2.is_inf()) bad++;
    xyzz_madd(r2, pts[4]);
    a2 = madd29<P>(a2, pts[4], false);
    if (!(xyzz_to_affine(acc29_to_xyzz<P>(a2)).x == xyzz_to_affine(r2).x) || !(xyzz_to_affine(acc29_to_xyzz<P>(a2)).y == xyzz_to_affine(r2).y)) bad++;
  }
  printf("{\"check\": \"%s host madd29 vs xyzz_madd incl. exceptional cases\", \"mismatches\": %d}\n", tag, bad);
  return bad;
}

int main(int argc, char** argv) {
  int bad = check_mul<Bn254FqParams>("bn254") + check_mul<Bls377FqParams>("bls377");
  bad += check_madd<Bn254FqParams, Bn254G1Params>("bn254") + check_madd<Bls377FqParams, Bls377G1Params>("bls377");
  if (argc > 1 && !strcmp(argv[1], "--host-only")) return bad != 0;
  using F = Bn254Fq;
  F hin[8];
  for (int i = 0; i < 8; i++) {
    hin[i] = F::one();
    hin[i].v[0] += 977u * (i + 1);
  }
  Affine<F> hp[8];
  for (int k = 0; k < F::N; k++) {
    hp[0].x.v[k] = Bn254G1Params::GX_MONT[k];
    hp[0].y.v[k] = Bn254G1Params::GY_MONT[k];
  }
  for (int i = 1; i < 8; i++) {
    XYZZ<F> t = i == 1 ? xyzz_dbl(XYZZ<F>::from_affine(hp[0])) : XYZZ<F>::from_affine(hp[i - 1]);
    if (i != 1) xyzz_madd(t, hp[0]);
    hp[i] = xyzz_to_affine(t);
  }
  F* din;
  Affine<F>* dp;
  void* dout;
  Clk* dclk;
  hipMalloc(&din, sizeof(hin));
  hipMalloc(&dp, sizeof(hp));
  hipMalloc(&dout, (size_t)2048 * 256 * sizeof(XYZZ<F>));
  hipMalloc(&dclk, sizeof(Clk));
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice);
  for (int wps : {1, 2, 3, 4}) run("mul32_bn254", k_chain<Bn254FqParams, 0>, 256 * wps, 4096, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 2, 3, 4}) run("mul29_bn254", k_chain<Bn254FqParams, 1>, 256 * wps, 4096, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 2, 3, 4}) run("sqr29_bn254", k_chain<Bn254FqParams, 2>, 256 * wps, 4096, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 3}) run("sub_carry_mul29_bn254", k_chain<Bn254FqParams, 3>, 256 * wps, 4096, dclk, (const F*)din, (F*)dout);
  for (int wps : {1, 2, 3, 4}) run("madd32_bn254", k_madd_chain<Bn254FqParams, 0>, 256 * wps, 512, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  for (int wps : {1, 2, 3, 4}) run("madd29_fast_bn254", k_madd_chain<Bn254FqParams, 1>, 256 * wps, 512, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  for (int wps : {1, 2, 3, 4}) run("madd29_bn254", k_madd_chain<Bn254FqParams, 2>, 256 * wps, 512, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  // sustained load (100+ ms per launch): does the shader clock hold under the denser multiply mix?
  run("madd32_bn254_long", k_madd_chain<Bn254FqParams, 0>, 768, 512 * 16, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  run("madd29_bn254_long", k_madd_chain<Bn254FqParams, 2>, 768, 512 * 16, dclk, (const Affine<F>*)dp, (XYZZ<F>*)dout);
  run("mul29_bn254_long", k_chain<Bn254FqParams, 1>, 768, 4096 * 32, dclk, (const F*)din, (F*)dout);
  run("mul32_bn254_long", k_chain<Bn254FqParams, 0>, 768, 4096 * 32, dclk, (const F*)din, (F*)dout);
  {
    using F2 = Bls377Fq;
    F2 hin2[8];
    for (int i = 0; i < 8; i++) {
      hin2[i] = F2::one();
      hin2[i].v[0] += 977u * (i + 1);
    }
    F2* din2;
    hipMalloc(&din2, sizeof(hin2));
    hipMemcpy(din2, hin2, sizeof(hin2), hipMemcpyHostToDevice);
    for (int wps : {1, 3}) run("mul32_bls377", k_chain<Bls377FqParams, 0>, 256 * wps, 2048, dclk, (const F2*)din2, (F2*)dout);
    for (int wps : {1, 3}) run("mul29_bls377", k_chain<Bls377FqParams, 1>, 256 * wps, 2048, dclk, (const F2*)din2, (F2*)dout);
  }
  return bad != 0;
}
