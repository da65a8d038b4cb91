// Host-side self-test of the reduced-radix arithmetic (field29.cuh, accum29.cuh): the same templates the kernels
// instantiate, compiled for the CPU and compared with the 32-bit-limb arithmetic of field.cuh / curve.cuh.  Runs without
// a GPU, so the CPU test suite (tests/test_field29.py) covers the kernels' arithmetic in every field that uses it.
#include <hip/hip_runtime.h>

#include "../../include/zeth_gpu.h"
#include "accum29.cuh"
#include "accum29x2.cuh"
#include "curve_constants.h"
#include "field29.cuh"

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

}  // namespace

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
