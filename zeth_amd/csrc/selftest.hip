// Host-side self-test of the reduced-radix arithmetic (field29.cuh, accum29.cuh): the same templates the kernels
// instantiate, compiled for the CPU and compared with the 32-bit-limb arithmetic of field.cuh / curve.cuh.  Runs without
// a GPU, so the CPU test suite (tests/test_field29.py) covers the kernels' arithmetic in every field that uses it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/zeth_gpu.h"
#include "accum29.cuh"
#include "accum29x2.cuh"
#include "curve_constants.h"
#include "curve_quad29.cuh"
#include "field29.cuh"
#include "knobs.h"

using namespace zg;

namespace {

struct Rng {
  uint64_t x;
  uint32_t next() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return (uint32_t)(x >> 16);
  }
};

template <class P>
Fp<P> random_element(Rng& g, int t) {
  using F = Fp<P>;
  F a;
  for (int i = 0; i < F::N; i++) a.v[i] = g.next();
  const int top_bits = (P::BITS - 1) % 32;  // one bit below the modulus' length: always < p
  a.v[F::N - 1] &= top_bits ? ((1u << top_bits) - 1u) : 0u;
  if (t % 97 == 0) a = F::zero();
  if (t % 97 == 1) {  // p - 1
    for (int i = 0; i < F::N; i++) a.v[i] = P::P[i];
    a.v[0] -= 1;
  }
  if (t % 97 == 2) a = F::one();
  return a;
}

template <class P>
int check_field(int iters, uint64_t seed) {
  using F = Fp<P>;
  Rng g{seed | 1};
  int bad = 0;
  for (int t = 0; t < iters; t++) {
    const F a = random_element<P>(g, t), b = random_element<P>(g, t + 50);
    const auto ua = f29::unpack(a), ub = f29::unpack(b);
    bad += !(f29::pack(ua) == a);
    const auto ra = f29::mul(ua, f29::one<P>()), rb = f29::mul(ub, f29::one<P>());  // below 2 p
    bad += !(f29::pack(ra) == a);
    bad += !(f29::pack(f29::mul(ra, ub)) == F::mul(a, b));
    bad += !(f29::pack(f29::sqr(ra)) == F::mul(a, a));
    const auto ab = f29::mul(ra, ub), bb = f29::sqr(rb);
    bad += !(f29::pack(f29::carry(f29::sub(ab, bb))) == F::sub(F::mul(a, b), F::mul(b, b)));
    bad += !(f29::pack(f29::add(ab, f29::dbl(bb))) == F::add(F::mul(a, b), F::dbl(F::mul(b, b))));
    bad += !(f29::pack(f29::mul(f29::carry(f29::sub(ab, bb)), f29::sub(bb, ab))) ==
             F::mul(F::sub(F::mul(a, b), F::mul(b, b)), F::sub(F::mul(b, b), F::mul(a, b))));
    bad += !(f29::pack(f29::neg(ab)) == F::neg(F::mul(a, b)));
    bad += f29::is_zero_mod_p(ra) != a.is_zero();
    // two products, one reduction: a b + (k p - b) a = 0 and a a + b b
    bad += !f29::is_zero_mod_p(f29::mul_add(ra, rb, f29::carry(f29::neg(rb)), ra));
    bad += !(f29::pack(f29::mul_add(ra, ra, rb, rb)) == F::add(F::mul(a, a), F::mul(b, b)));
    // the stored form read as it is: x R' against y R gives the stored form of x y; canonical forms
    const auto s = f29::canon(f29::mul(ra, f29::unpack_raw(b)));
    bad += !(f29::to_words(s) == F::mul(a, b));
    bad += !(f29::to_words(f29::neg_canon(s)) == F::neg(F::mul(a, b)));
    bad += !(f29::to_words(f29::canon(f29::mul(ra, f29::r_conv<P>()))) == a);
  }
  return bad;
}

template <class P, class GP>
int check_mixed_addition() {
  using F = Fp<P>;
  Affine<F> gen;
  for (int k = 0; k < F::N; k++) {
    gen.x.v[k] = GP::GX_MONT[k];
    gen.y.v[k] = GP::GY_MONT[k];
  }
  int bad = 0;
  Affine<F> pts[6];
  XYZZ<F> t = XYZZ<F>::from_affine(gen);
  for (int i = 0; i < 6; i++) {
    pts[i] = xyzz_to_affine(t);
    t = xyzz_dbl(t);
    xyzz_madd(t, gen);
  }
  auto same = [](const Acc29<P>& a, const XYZZ<F>& r) {
    const XYZZ<F> x = acc29_to_xyzz<P>(a);
    if (x.is_inf() || r.is_inf()) return x.is_inf() == r.is_inf();
    const Affine<F> p = xyzz_to_affine(x), q = xyzz_to_affine(r);
    return p.x == q.x && p.y == q.y;
  };
  // every exceptional case in one sequence: identity base, first point, doubling, cancellation, restart
  const int seq[] = {-1, 0, 0, 1, 2, -1, 3, 4, 5, 0, 1};
  const int neg[] = {0, 0, 0, 0, 1, 0, 0, 1, 0, 1, 1};
  XYZZ<F> ref = XYZZ<F>::inf();
  Acc29<P> acc = acc29_inf<P>();
  for (int s = 0; s < 11; s++) {
    const Affine<F> p = seq[s] < 0 ? Affine<F>::inf() : pts[seq[s]];
    Affine<F> ps = p;
    ps.y = F::cneg(p.y, neg[s] != 0);
    xyzz_madd(ref, ps);
    acc = madd29<P>(acc, p, neg[s] != 0);
    bad += !same(acc, ref);
  }
  for (int rep = 0; rep < 2; rep++) {  // P + P and P - P, then on
    XYZZ<F> r2 = XYZZ<F>::from_affine(pts[2]);
    Acc29<P> a2 = acc29_from_affine<P>(pts[2], false);
    Affine<F> q = pts[2];
    q.y = F::cneg(q.y, rep != 0);
    xyzz_madd(r2, q);
    a2 = madd29<P>(a2, pts[2], rep != 0);
    bad += !same(a2, r2);
    xyzz_madd(r2, pts[4]);
    a2 = madd29<P>(a2, pts[4], false);
    bad += !same(a2, r2);
  }
  // a long walk: 200 additions of alternating sign over the six points
  for (int i = 0; i < 200; i++) {
    const Affine<F> p = pts[(i * 7 + 3) % 6];
    const bool n = (i % 3) == 1;
    Affine<F> ps = p;
    ps.y = F::cneg(p.y, n);
    xyzz_madd(ref, ps);
    acc = madd29<P>(acc, p, n);
  }
  bad += !same(acc, ref);
  return bad;
}

// the G2 mixed addition (accum29x2.cuh) against xyzz_madd over Fp2, and the Fq2 product and square
template <class P, int NR, class GP>
int check_mixed_addition_g2(int iters, uint64_t seed) {
  using B = Fp<P>;
  using F = Fp2<B, NR>;
  using Q = F29x2<P, NR>;
  using A = AccumOf<F>;
  Rng g{seed | 1};
  int bad = 0;
  for (int t = 0; t < iters; t++) {
    const F a{random_element<P>(g, t), random_element<P>(g, t + 7)}, b{random_element<P>(g, t + 50), random_element<P>(g, t + 3)};
    const auto ra = Q::normalise(Q::unpack(a));
    bad += !(Q::pack(Q::mul(ra, Q::unpack(b))) == F::mul(a, b));
    bad += !(Q::pack(Q::sqr(ra)) == F::sqr(a));
    bad += !(Q::pack(Q::sqr(f29::carry(f29::sub(ra, Q::normalise(Q::unpack(b)))))) == F::sqr(F::sub(a, b)));
  }
  Affine<F> gen;
  for (int k = 0; k < B::N; k++) {
    gen.x.c0.v[k] = GP::GX_C0_MONT[k];
    gen.x.c1.v[k] = GP::GX_C1_MONT[k];
    gen.y.c0.v[k] = GP::GY_C0_MONT[k];
    gen.y.c1.v[k] = GP::GY_C1_MONT[k];
  }
  Affine<F> pts[6];
  XYZZ<F> t = XYZZ<F>::from_affine(gen);
  for (int i = 0; i < 6; i++) {
    pts[i] = xyzz_to_affine(t);
    t = xyzz_dbl(t);
    xyzz_madd(t, gen);
  }
  auto same = [](const typename A::Acc& a, const XYZZ<F>& r) {
    const XYZZ<F> x = A::to_xyzz(a);
    if (x.is_inf() || r.is_inf()) return x.is_inf() == r.is_inf();
    const Affine<F> p = xyzz_to_affine(x), q = xyzz_to_affine(r);
    return p.x == q.x && p.y == q.y;
  };
  const int seq[] = {-1, 0, 0, 1, 2, -1, 3, 4, 5, 0, 1};
  const int neg[] = {0, 0, 0, 0, 1, 0, 0, 1, 0, 1, 1};
  XYZZ<F> ref = XYZZ<F>::inf();
  typename A::Acc acc = A::from_affine(Affine<F>::inf(), false);
  for (int s = 0; s < 11; s++) {
    const Affine<F> p = seq[s] < 0 ? Affine<F>::inf() : pts[seq[s]];
    Affine<F> ps = p;
    ps.y = F::cneg(p.y, neg[s] != 0);
    xyzz_madd(ref, ps);
    acc = A::madd(acc, p, neg[s] != 0);
    bad += !same(acc, ref);
  }
  for (int rep = 0; rep < 2; rep++) {  // P + P and P - P, then on
    XYZZ<F> r2 = XYZZ<F>::from_affine(pts[2]);
    typename A::Acc a2 = A::from_affine(pts[2], false);
    Affine<F> q = pts[2];
    q.y = F::cneg(q.y, rep != 0);
    xyzz_madd(r2, q);
    a2 = A::madd(a2, pts[2], rep != 0);
    bad += !same(a2, r2);
    xyzz_madd(r2, pts[4]);
    a2 = A::madd(a2, pts[4], false);
    bad += !same(a2, r2);
  }
  for (int i = 0; i < 200; i++) {
    const Affine<F> p = pts[(i * 7 + 3) % 6];
    const bool n = (i % 3) == 1;
    Affine<F> ps = p;
    ps.y = F::cneg(p.y, n);
    xyzz_madd(ref, ps);
    acc = A::madd(acc, p, n);
  }
  bad += !same(acc, ref);
  return bad;
}

// the same operations on the device: out[i] = a[i] + b[i] (i < n), dbl_out[i] = 2 a[i]; one quad per pair
template <class Q, class F>
__global__ void k_selftest_quad(const XYZZ<F>* a, const XYZZ<F>* b, int n, XYZZ<F>* out, XYZZ<F>* dbl_out, XYZZ<F>* mid_out) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const unsigned q = threadIdx.x & 3u;
  if (i >= n) return;
  const typename Q::Coord ca = Q::load(a + i, q), cb = Q::load(b + i, q);
  Q::store_final(out + i, q, Q::add(ca, cb, q));
  Q::store_final(dbl_out + i, q, Q::dbl(ca, q));
  // through the buffer form and back, then on
  Q::store(mid_out + i, q, Q::add(ca, cb, q));
  const typename Q::Coord cm = Q::load(mid_out + i, q);
  Q::store_final(mid_out + i, q, Q::add(cm, ca, q));
}

// (launched from host-only code below: instantiated here for the device pass)
#define ZG_INSTANTIATE_SELFTEST_QUAD(Q, F) \
  template __global__ void k_selftest_quad<Q, F>(const XYZZ<F>*, const XYZZ<F>*, int, XYZZ<F>*, XYZZ<F>*, XYZZ<F>*)
ZG_INSTANTIATE_SELFTEST_QUAD(QuadOps32<Fp<Bn254FqParams>>, Fp<Bn254FqParams>);
ZG_INSTANTIATE_SELFTEST_QUAD(QuadOps29<Bn254FqParams>, Fp<Bn254FqParams>);
ZG_INSTANTIATE_SELFTEST_QUAD(QuadOps32<Fp<Bls377FqParams>>, Fp<Bls377FqParams>);
ZG_INSTANTIATE_SELFTEST_QUAD(QuadOps29<Bls377FqParams>, Fp<Bls377FqParams>);
using ZgSelftestFq2Bn = Fp2<Fp<Bn254FqParams>, 1>;
using ZgSelftestFq2Bls = Fp2<Fp<Bls377FqParams>, 5>;
ZG_INSTANTIATE_SELFTEST_QUAD(QuadOps32<ZgSelftestFq2Bn>, ZgSelftestFq2Bn);
ZG_INSTANTIATE_SELFTEST_QUAD(QuadOps32<ZgSelftestFq2Bls>, ZgSelftestFq2Bls);

#if !defined(__HIP_DEVICE_COMPILE__)  // (the host stand-ins for the lane exchanges exist in the host pass only)
// The quad-lane point arithmetic (curve_quad.cuh, curve_quad29.cuh) with the four lanes as four host threads, against
// the one-lane formulas of curve.cuh: additions of distinct points, of a point to itself, to its negative and to the
// identity, doublings, and the round trip through the kernels' buffer forms.
template <class Fn>
void run_quad(Fn fn) {
  QuadHost h;
  quad_host() = &h;
  std::thread t[4];
  for (int q = 0; q < 4; q++)
    t[q] = std::thread([&fn, q] {
      quad_host_lane() = q;
      fn((unsigned)q);
    });
  for (auto& x : t) x.join();
  quad_host() = nullptr;
}

template <class F>
bool same_point(const XYZZ<F>& x, const XYZZ<F>& r) {
  if (x.is_inf() || r.is_inf()) return x.is_inf() == r.is_inf();
  const Affine<F> p = xyzz_to_affine(x), q = xyzz_to_affine(r);
  return p.x == q.x && p.y == q.y;
}

// a point in the form the quad kernels' buffers hold for Q
template <class Q, class F>
XYZZ<F> to_buffer_form(const XYZZ<F>& p) {
  if constexpr (Q::rprime_buffers) {
    using P = typename F::Params;
    const auto o = f29::one<P>();
    XYZZ<F> r;
    r.x = f29::to_words(f29::canon(f29::mul(f29::unpack(p.x), o)));
    r.y = f29::to_words(f29::canon(f29::mul(f29::unpack(p.y), o)));
    r.zz = f29::to_words(f29::canon(f29::mul(f29::unpack(p.zz), o)));
    r.zzz = f29::to_words(f29::canon(f29::mul(f29::unpack(p.zzz), o)));
    return r;
  } else {
    return p;
  }
}

template <class Q, class F, class GX>
int check_quad_ops(const GX& set_generator) {
  Affine<F> gen;
  set_generator(gen);
  int bad = 0;
  XYZZ<F> pts[8];  // multiples of the generator with ZZ != 1
  XYZZ<F> t = XYZZ<F>::from_affine(gen);
  for (int i = 0; i < 8; i++) {
    t = xyzz_dbl(t);
    xyzz_madd(t, gen);
    pts[i] = t;
  }
  auto quad_add_points = [&](const XYZZ<F>& a, const XYZZ<F>& b, bool final_form) {
    const XYZZ<F> ba = to_buffer_form<Q>(a), bb = to_buffer_form<Q>(b);
    XYZZ<F> out;
    run_quad([&](unsigned q) {
      const typename Q::Coord r = Q::add(Q::load(&ba, q), Q::load(&bb, q), q);
      if (final_form)
        Q::store_final(&out, q, r);
      else
        Q::store(&out, q, r);
    });
    return out;
  };
  auto one_lane_add = [](XYZZ<F> a, const XYZZ<F>& b) {
    xyzz_add(a, b);
    return a;
  };
  const XYZZ<F> inf = XYZZ<F>::inf();
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      XYZZ<F> b = pts[j];
      if ((i + j) % 3 == 1) b.y = F::neg(b.y);  // j == i: P + P or P - P
      bad += !same_point(quad_add_points(pts[i], b, true), one_lane_add(pts[i], b));
    }
  // the same point under two different (zz, zzz): still a doubling; and the identity on either side
  bad += !same_point(quad_add_points(pts[3], XYZZ<F>::from_affine(xyzz_to_affine(pts[3])), true), xyzz_dbl(pts[3]));
  bad += !same_point(quad_add_points(inf, pts[2], true), pts[2]);
  bad += !same_point(quad_add_points(pts[2], inf, true), pts[2]);
  bad += !same_point(quad_add_points(inf, inf, true), inf);
  // a chain held in the lanes: running sums and doublings as k_reduce1 does them, stored in the buffer form, loaded again
  XYZZ<F> ref_run = inf, ref_tot = inf, got_tot, got_run;
  for (int i = 0; i < 8; i++) {
    xyzz_add(ref_run, pts[i]);
    xyzz_add(ref_tot, ref_run);
  }
  for (int i = 0; i < 5; i++) ref_tot = xyzz_dbl(ref_tot);
  XYZZ<F> bp[8];
  for (int i = 0; i < 8; i++) bp[i] = to_buffer_form<Q>(pts[i]);
  XYZZ<F> mid;
  run_quad([&](unsigned q) {
    typename Q::Coord run = Q::zero(), tot = Q::zero();
    for (int i = 0; i < 8; i++) {
      run = Q::add(run, Q::load(&bp[i], q), q);
      tot = Q::add(tot, run, q);
    }
    Q::store(&mid, q, tot);
    Q::store_final(&got_run, q, run);
  });
  run_quad([&](unsigned q) {
    typename Q::Coord tot = Q::load(&mid, q);
    for (int i = 0; i < 5; i++) tot = Q::dbl(tot, q);
    Q::store_final(&got_tot, q, tot);
  });
  bad += !same_point(got_tot, ref_tot);
  bad += !same_point(got_run, ref_run);
  // doubling the identity
  XYZZ<F> di;
  run_quad([&](unsigned q) { Q::store_final(&di, q, Q::dbl(Q::zero(), q)); });
  bad += !di.is_inf();
  return bad;
}

template <class Q, class F, class GX>
int check_quad_device(const GX& set_generator, int* bad_out) {
  Affine<F> gen;
  set_generator(gen);
  constexpr int NP = 8, N = NP * NP + 4;
  XYZZ<F> pts[NP], ha[N], hb[N], ba[N], bb[N], got[N], got_dbl[N], got_mid[N];
  XYZZ<F> t = XYZZ<F>::from_affine(gen);
  for (int i = 0; i < NP; i++) {
    t = xyzz_dbl(t);
    xyzz_madd(t, gen);
    pts[i] = t;
  }
  int n = 0;
  for (int i = 0; i < NP; i++)
    for (int j = 0; j < NP; j++) {
      ha[n] = pts[i];
      hb[n] = pts[j];
      if ((i + j) % 3 == 1) hb[n].y = F::neg(hb[n].y);
      n++;
    }
  ha[n] = XYZZ<F>::inf(), hb[n++] = pts[1];
  ha[n] = pts[1], hb[n++] = XYZZ<F>::inf();
  ha[n] = XYZZ<F>::inf(), hb[n++] = XYZZ<F>::inf();
  ha[n] = pts[3], hb[n++] = XYZZ<F>::from_affine(xyzz_to_affine(pts[3]));
  for (int i = 0; i < N; i++) {
    ba[i] = to_buffer_form<Q>(ha[i]);
    bb[i] = to_buffer_form<Q>(hb[i]);
  }
  XYZZ<F>* d = nullptr;
  if (hipMalloc(&d, 5 * sizeof(ha)) != hipSuccess) return ZG_ERR_HIP;
  int rc = ZG_OK;
  if (hipMemcpy(d, ba, sizeof(ba), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + N, bb, sizeof(bb), hipMemcpyHostToDevice) != hipSuccess)
    rc = ZG_ERR_HIP;
  if (rc == ZG_OK) {
    hipLaunchKernelGGL((k_selftest_quad<Q, F>), dim3((4 * N + 63) / 64), dim3(64), 0, 0, d, d + N, N, d + 2 * N, d + 3 * N, d + 4 * N);
    if (hipMemcpy(got, d + 2 * N, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(got_dbl, d + 3 * N, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(got_mid, d + 4 * N, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess)
      rc = ZG_ERR_HIP;
  }
  (void)hipFree(d);
  if (rc != ZG_OK) return rc;
  int bad = 0;
  for (int i = 0; i < N; i++) {
    XYZZ<F> r = ha[i];
    xyzz_add(r, hb[i]);
    const bool ok_add = same_point(got[i], r);
    const bool ok_dbl = same_point(got_dbl[i], xyzz_dbl(ha[i]));
    xyzz_add(r, ha[i]);
    const bool ok_mid = same_point(got_mid[i], r);
    if (!(ok_add && ok_dbl && ok_mid) && zg_knob("ZG_SELFTEST_VERBOSE")) fprintf(stderr, "quad device case %d: add %d dbl %d mid %d\n", i, ok_add, ok_dbl, ok_mid);
    bad += !ok_add + !ok_dbl + !ok_mid;
  }
  *bad_out += bad;
  return ZG_OK;
}

template <class P, class GP>
int check_quad_g1() {
  using F = Fp<P>;
  auto set_gen = [](Affine<F>& gen) {
    for (int k = 0; k < F::N; k++) {
      gen.x.v[k] = GP::GX_MONT[k];
      gen.y.v[k] = GP::GY_MONT[k];
    }
  };
  return check_quad_ops<QuadOps32<F>, F>(set_gen) + check_quad_ops<QuadOps29<P>, F>(set_gen);
}
template <class P, int NR, class GP>
int check_quad_g2() {
  using B = Fp<P>;
  using F = Fp2<B, NR>;
  auto set_gen = [](Affine<F>& gen) {
    for (int k = 0; k < B::N; k++) {
      gen.x.c0.v[k] = GP::GX_C0_MONT[k];
      gen.x.c1.v[k] = GP::GX_C1_MONT[k];
      gen.y.c0.v[k] = GP::GY_C0_MONT[k];
      gen.y.c1.v[k] = GP::GY_C1_MONT[k];
    }
  };
  return check_quad_ops<QuadOps32<F>, F>(set_gen);
}
#endif

}  // namespace

extern "C" int zg_selftest_quad_lanes_device(int curve_group, int* mismatches) {
  if (!mismatches) return ZG_ERR_INVALID_ARGUMENT;
  *mismatches = 0;
#if !defined(__HIP_DEVICE_COMPILE__)
  auto g1 = [](auto params, auto gparams) {
    using P = decltype(params);
    using GP = decltype(gparams);
    using F = Fp<P>;
    return [](Affine<F>& gen) {
      for (int k = 0; k < F::N; k++) {
        gen.x.v[k] = GP::GX_MONT[k];
        gen.y.v[k] = GP::GY_MONT[k];
      }
    };
  };
  switch (curve_group) {
    case 0: {
      using P = Bn254FqParams;
      const auto sg = g1(P{}, Bn254G1Params{});
      int rc = check_quad_device<QuadOps32<Fp<P>>, Fp<P>>(sg, mismatches);
      if (rc == ZG_OK) rc = check_quad_device<QuadOps29<P>, Fp<P>>(sg, mismatches);
      return rc;
    }
    case 1: {
      using P = Bls377FqParams;
      const auto sg = g1(P{}, Bls377G1Params{});
      int rc = check_quad_device<QuadOps32<Fp<P>>, Fp<P>>(sg, mismatches);
      if (rc == ZG_OK) rc = check_quad_device<QuadOps29<P>, Fp<P>>(sg, mismatches);
      return rc;
    }
    case 2: {  // the Fq2 instantiations of the same templates (32-bit limbs): the G2 combine and reduction kernels
      using B = Fp<Bn254FqParams>;
      using F = Fp2<B, 1>;
      using GP = Bn254G2Params;
      return check_quad_device<QuadOps32<F>, F>(
          [](Affine<F>& gen) {
            for (int k = 0; k < B::N; k++) {
              gen.x.c0.v[k] = GP::GX_C0_MONT[k], gen.x.c1.v[k] = GP::GX_C1_MONT[k];
              gen.y.c0.v[k] = GP::GY_C0_MONT[k], gen.y.c1.v[k] = GP::GY_C1_MONT[k];
            }
          },
          mismatches);
    }
    case 3: {
      using B = Fp<Bls377FqParams>;
      using F = Fp2<B, 5>;
      using GP = Bls377G2Params;
      return check_quad_device<QuadOps32<F>, F>(
          [](Affine<F>& gen) {
            for (int k = 0; k < B::N; k++) {
              gen.x.c0.v[k] = GP::GX_C0_MONT[k], gen.x.c1.v[k] = GP::GX_C1_MONT[k];
              gen.y.c0.v[k] = GP::GY_C0_MONT[k], gen.y.c1.v[k] = GP::GY_C1_MONT[k];
            }
          },
          mismatches);
    }
    default: return ZG_ERR_INVALID_ARGUMENT;
  }
#endif
  return ZG_OK;
}

extern "C" int zg_selftest_quad_lanes(int curve_group, int* mismatches) {
  if (!mismatches) return ZG_ERR_INVALID_ARGUMENT;
#if !defined(__HIP_DEVICE_COMPILE__)
  switch (curve_group) {
    case 0: *mismatches = check_quad_g1<Bn254FqParams, Bn254G1Params>(); break;
    case 1: *mismatches = check_quad_g1<Bls377FqParams, Bls377G1Params>(); break;
    case 2: *mismatches = check_quad_g2<Bn254FqParams, 1, Bn254G2Params>(); break;
    case 3: *mismatches = check_quad_g2<Bls377FqParams, 5, Bls377G2Params>(); break;
    default: return ZG_ERR_INVALID_ARGUMENT;
  }
#endif
  return ZG_OK;
}

extern "C" int zg_selftest_reduced_radix(int field, int iters, uint64_t seed, int* mismatches) {
  if (!mismatches || iters < 0) return ZG_ERR_INVALID_ARGUMENT;
  switch (field) {
    case 0: *mismatches = check_field<Bn254FqParams>(iters, seed) + check_mixed_addition<Bn254FqParams, Bn254G1Params>(); break;
    case 1: *mismatches = check_field<Bls377FqParams>(iters, seed) + check_mixed_addition<Bls377FqParams, Bls377G1Params>(); break;
    case 2: *mismatches = check_field<Bn254FrParams>(iters, seed); break;
    case 3: *mismatches = check_field<Bls377FrParams>(iters, seed); break;
    case 4: *mismatches = check_mixed_addition_g2<Bn254FqParams, 1, Bn254G2Params>(iters, seed); break;
    case 5: *mismatches = check_mixed_addition_g2<Bls377FqParams, 5, Bls377G2Params>(iters, seed); break;
    default: return ZG_ERR_INVALID_ARGUMENT;
  }
  return ZG_OK;
}
