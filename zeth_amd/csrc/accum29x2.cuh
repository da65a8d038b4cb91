// The bucket accumulator of k_accum over Fq2 (G2) in the reduced-radix arithmetic of field29.cuh.
//
// An Fq2 element is a pair of lazy E29 values with common bounds.  Its product is written on the limb products directly,
// not through Karatsuba: (a0 + a1 u)(b0 + b1 u) with u^2 = -k is
//     c0 = a0 b0 + (m p - k a1) b1        c1 = a0 b1 + a1 b0
// -- two products and ONE Montgomery reduction per component (f29::mul_add), 6 NL^2 limb multiplies against the 6 NL^2
// of three separate products, but none of Karatsuba's five field additions and no carry chains; the square is
// (a0 + a1)(a0 - a1), 2 a0 a1 for k = 1 (4 NL^2).  The mixed addition (EFD madd-2008-s, as accum29.cuh) goes further:
//     X3 = R^2 - (PPP + 2 Q)              every component one reduction over its products and the subtrahend times one
//     Y3 = R (Q - X3) - Y1 PPP            four products per component, one reduction
// so that X3 and Y3 come out below 2 p and the accumulator's bounds reproduce themselves at 2 p in every coordinate
// (with wider X the square of U2 - X1 would leave the multiplier's range of alt_bn128: R'/p = 169).  57 NL^2 limb
// multiplies per addition against 3 x 10 x 2 NL^2 = 60 NL^2 plus ~40 Fq2 additions on the 32-bit-limb path.
//
// Same group law and the same results as curve.cuh's xyzz_madd over Fp2 (libff's mixed addition on G2, SURVEY.md 3.1
// step 5c); exceptional inputs are found after the fact as in accum29.cuh (ZZ3 congruent to zero in both components).
#pragma once
#include "accum29.cuh"
#include "curve.cuh"
#include "field29.cuh"

namespace zg {

template <class P, uint64_t VB, uint64_t LMAX>
struct E29x2 {
  static constexpr uint64_t vb = VB, lmax = LMAX;
  E29<P, VB, LMAX> c0, c1;
};

namespace f29 {

template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
ZG_HD auto add(const E29x2<P, VA, LA>& a, const E29x2<P, VB, LB>& b) {
  const auto c0 = add(a.c0, b.c0);
  using T = decltype(c0);
  return E29x2<P, T::vb, T::lmax>{c0, add(a.c1, b.c1)};
}
template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
ZG_HD auto sub(const E29x2<P, VA, LA>& a, const E29x2<P, VB, LB>& b) {
  const auto c0 = sub(a.c0, b.c0);
  using T = decltype(c0);
  return E29x2<P, T::vb, T::lmax>{c0, sub(a.c1, b.c1)};
}
template <class P, uint64_t VA, uint64_t LA>
ZG_HD auto dbl(const E29x2<P, VA, LA>& a) {
  const auto c0 = dbl(a.c0);
  using T = decltype(c0);
  return E29x2<P, T::vb, T::lmax>{c0, dbl(a.c1)};
}
template <class P, uint64_t VA, uint64_t LA>
ZG_HD auto neg(const E29x2<P, VA, LA>& a) {
  const auto c0 = neg(a.c0);
  using T = decltype(c0);
  return E29x2<P, T::vb, T::lmax>{c0, neg(a.c1)};
}
template <class P, uint64_t VA, uint64_t LA>
ZG_HD auto carry(const E29x2<P, VA, LA>& a) {
  const auto c0 = carry(a.c0);
  using T = decltype(c0);
  return E29x2<P, T::vb, T::lmax>{c0, carry(a.c1)};
}
template <uint64_t VB, uint64_t LB, class P, uint64_t VA, uint64_t LA>
ZG_HD E29x2<P, VB, LB> widen2(const E29x2<P, VA, LA>& a) {
  return E29x2<P, VB, LB>{widen<VB, LB>(a.c0), widen<VB, LB>(a.c1)};
}
template <class P, uint64_t VA, uint64_t LA>
ZG_HD E29x2<P, VA, LA> select(const E29x2<P, VA, LA>& a, const E29x2<P, VA, LA>& b, bool flag) {
  return E29x2<P, VA, LA>{select(a.c0, b.c0, flag), select(a.c1, b.c1, flag)};
}

// k x for the non-residue's k = 1 or 5, lazily
template <int NR, class P, uint64_t VA, uint64_t LA>
ZG_HD auto times_nr(const E29<P, VA, LA>& x) {
  static_assert(NR == 1 || NR == 5, "u^2 = -1 or -5");
  if constexpr (NR == 1) {
    return x;
  } else {
    return add(dbl(dbl(x)), x);
  }
}
// m p - k x with normalised limbs: the operand that turns a subtraction of products into a sum
template <int NR, class P, uint64_t VA, uint64_t LA>
ZG_HD auto neg_nr(const E29<P, VA, LA>& x) {
  return carry(neg(times_nr<NR>(x)));
}

}  // namespace f29

// Fq2 = Fq[u] / (u^2 + NR) over lazy values
template <class P, int NR>
struct F29x2 {
  using K = K29<P>;
  using N2 = E29x2<P, 2, K::LNORM>;  // a product: both components below 2 p, limbs normalised

  // a b; the first operand is the one negated inside, so the caller passes the operand with the smaller bound first
  template <uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
  ZG_HD static N2 mul(const E29x2<P, VA, LA>& a, const E29x2<P, VB, LB>& b) {
    N2 r;
    r.c0 = f29::mul_add_chains(a.c0, b.c0, f29::neg_nr<NR>(a.c1), b.c1);
    r.c1 = f29::mul_add_chains(a.c0, b.c1, a.c1, b.c0);
    return r;
  }
  template <uint64_t VA, uint64_t LA>
  ZG_HD static N2 sqr(const E29x2<P, VA, LA>& a) {
    N2 r;
    if constexpr (NR == 1) {
      r.c0 = f29::mul_chains(f29::add(a.c0, a.c1), f29::add(a.c0, f29::neg_nr<1>(a.c1)));
    } else {
      r.c0 = f29::mul_add_chains(a.c0, a.c0, f29::neg_nr<NR>(a.c1), a.c1);
    }
    r.c1 = f29::mul_chains(f29::dbl(a.c0), a.c1);
    return r;
  }
  // componentwise product with a base-field value (used to bring a lazy value below 2 p: times one)
  template <uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
  ZG_HD static N2 scale(const E29x2<P, VA, LA>& a, const E29<P, VB, LB>& k) {
    return N2{f29::mul_chains(a.c0, k), f29::mul_chains(a.c1, k)};
  }
  template <uint64_t VA, uint64_t LA>
  ZG_HD static N2 normalise(const E29x2<P, VA, LA>& a) {
    return scale(a, f29::one<P>());
  }
  ZG_HD static auto unpack(const Fp2<Fp<P>, NR>& s) {
    const auto c0 = f29::unpack(s.c0);
    using T = decltype(c0);
    return E29x2<P, T::vb, T::lmax>{c0, f29::unpack(s.c1)};
  }
  template <uint64_t VA, uint64_t LA>
  ZG_HD static Fp2<Fp<P>, NR> pack(const E29x2<P, VA, LA>& a) {
    return Fp2<Fp<P>, NR>{f29::pack(a.c0), f29::pack(a.c1)};
  }
  ZG_HD static bool is_zero_mod_p(const N2& a) { return f29::is_zero_mod_p(a.c0) && f29::is_zero_mod_p(a.c1); }
  ZG_HD static N2 zero() {
    N2 r;
#pragma unroll
    for (int i = 0; i < K::NL; i++) r.c0.v[i] = r.c1.v[i] = 0;
    return r;
  }
  ZG_HD static N2 one() {
    N2 r = zero();
    r.c0 = f29::widen<2, K::LNORM>(f29::one<P>());
    return r;
  }
};

template <class P, int NR>
struct Acc29x2 {
  using N2 = typename F29x2<P, NR>::N2;
  N2 x, y, zz, zzz;  // every coordinate a product: below 2 p
};

template <class B, int NR>
ZG_HD bool affine_words_zero(const Affine<Fp2<B, NR>>& p) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < B::N; i++) o |= p.x.c0.v[i] | p.x.c1.v[i] | p.y.c0.v[i] | p.y.c1.v[i];
  return o == 0;
}

template <class P, int NR>
ZG_HD Acc29x2<P, NR> acc29x2_inf() {
  using Q = F29x2<P, NR>;
  return Acc29x2<P, NR>{Q::zero(), Q::zero(), Q::zero(), Q::zero()};
}

// (+-)y of a base from the stored form
template <class P, int NR>
ZG_HD auto base_y29x2(const Fp2<Fp<P>, NR>& y, bool negate) {
  const auto c0 = base_y29<P>(y.c0, negate);
  using T = decltype(c0);
  return E29x2<P, T::vb, T::lmax>{c0, base_y29<P>(y.c1, negate)};
}

template <class P, int NR>
ZG_HD Acc29x2<P, NR> acc29x2_from_affine(const Affine<Fp2<Fp<P>, NR>>& p, bool negate) {
  using Q = F29x2<P, NR>;
  if (affine_words_zero(p)) return acc29x2_inf<P, NR>();
  Acc29x2<P, NR> a;
  a.x = Q::normalise(Q::unpack(p.x));
  a.y = Q::normalise(base_y29x2<P, NR>(p.y, negate));
  a.zz = Q::one();
  a.zzz = Q::one();
  return a;
}

template <class P, int NR>
ZG_HD XYZZ<Fp2<Fp<P>, NR>> acc29x2_to_xyzz(const Acc29x2<P, NR>& a) {
  using Q = F29x2<P, NR>;
  XYZZ<Fp2<Fp<P>, NR>> r;
  r.x = Q::pack(a.x);
  r.y = Q::pack(a.y);
  r.zz = Q::pack(a.zz);
  r.zzz = Q::pack(a.zzz);
  return r;
}

// 2 (+-)P for an affine P != identity (EFD mdbl-2008-s-1, a = 0); rare path
template <class P, int NR>
ZG_HD Acc29x2<P, NR> mdbl29x2(const Affine<Fp2<Fp<P>, NR>>& p, bool negate) {
  using Q = F29x2<P, NR>;
  const auto x = Q::normalise(Q::unpack(p.x));
  const auto y = Q::normalise(base_y29x2<P, NR>(p.y, negate));
  const auto U = f29::carry(f29::dbl(y));
  const auto V = Q::sqr(U);
  const auto W = Q::mul(V, U);
  const auto S = Q::mul(V, x);
  const auto XX = Q::sqr(x);
  const auto M = Q::normalise(f29::add(f29::dbl(XX), XX));
  const auto X3 = Q::normalise(f29::carry(f29::sub(Q::sqr(M), f29::dbl(S))));
  const auto Y3 = Q::normalise(f29::carry(f29::sub(Q::mul(M, f29::carry(f29::sub(S, X3))), Q::mul(W, y))));
  return Acc29x2<P, NR>{X3, Y3, V, W};
}

// acc + (+-)P by the straight-line formula; *suspect as in madd29_fast
template <class P, int NR>
ZG_HD Acc29x2<P, NR> madd29x2_fast(const Acc29x2<P, NR>& a, const Affine<Fp2<Fp<P>, NR>>& p, bool negate, bool* suspect) {
  using Q = F29x2<P, NR>;
  using K = K29<P>;
  const auto U2 = Q::mul(a.zz, Q::unpack(p.x));
  const auto S2 = Q::mul(a.zzz, base_y29x2<P, NR>(p.y, negate));
  const auto Pd = f29::carry(f29::sub(U2, a.x));
  const auto R = f29::carry(f29::sub(S2, a.y));
  const auto PP = Q::sqr(Pd);
  const auto PPP = Q::mul(PP, Pd);
  const auto Qv = Q::mul(PP, a.x);
  // X3 = R^2 - (PPP + 2 Q): the subtrahend enters each component's reduction as (m p - T) times one
  const auto nT = f29::carry(f29::neg(f29::add(PPP, f29::dbl(Qv))));
  const auto one = f29::one<P>();
  typename Q::N2 X3;
  X3.c0 = f29::mul_add3(R.c0, R.c0, f29::neg_nr<NR>(R.c1), R.c1, nT.c0, one);
  X3.c1 = f29::mul_add_chains(f29::dbl(R.c0), R.c1, nT.c1, one);
  // Y3 = R (Q - X3) - Y1 PPP
  const auto D = f29::carry(f29::sub(Qv, X3));
  const auto nY0 = f29::carry(f29::neg(a.y.c0)), nY1 = f29::carry(f29::neg(a.y.c1));
  typename Q::N2 Y3;
  Y3.c0 = f29::mul_add4(R.c0, D.c0, f29::neg_nr<NR>(R.c1), D.c1, nY0, PPP.c0, f29::carry(f29::times_nr<NR>(a.y.c1)), PPP.c1);
  Y3.c1 = f29::mul_add4(R.c0, D.c1, R.c1, D.c0, nY0, PPP.c1, nY1, PPP.c0);
  Acc29x2<P, NR> r;
  r.x = X3;
  r.y = Y3;
  r.zz = Q::mul(a.zz, PP);
  r.zzz = Q::mul(a.zzz, PPP);
  const uint32_t l0 = r.zz.c0.v[0], l1 = r.zz.c1.v[0];
  *suspect = affine_words_zero(p) || ((l0 == 0 || l0 == K::PL.v[0]) && (l1 == 0 || l1 == K::PL.v[0]));
  return r;
}

// the exceptional cases, inlined like madd29_fixup (see there)
template <class P, int NR>
ZG_HD Acc29x2<P, NR> madd29x2_fixup(const Acc29x2<P, NR>& a, const Affine<Fp2<Fp<P>, NR>>& p, bool negate,
                                    const Acc29x2<P, NR>& fast) {
  using Q = F29x2<P, NR>;
  if (affine_words_zero(p)) return a;                                       // + identity
  if (Q::is_zero_mod_p(a.zz)) return acc29x2_from_affine<P, NR>(p, negate);  // identity + P
  if (!Q::is_zero_mod_p(fast.zz)) return fast;                              // false alarm
  const auto S2 = Q::mul(a.zzz, base_y29x2<P, NR>(p.y, negate));
  const auto R = Q::normalise(f29::carry(f29::sub(S2, a.y)));
  if (Q::is_zero_mod_p(R)) return mdbl29x2<P, NR>(p, negate);
  return acc29x2_inf<P, NR>();
}

template <class P, int NR>
ZG_HD Acc29x2<P, NR> madd29x2(const Acc29x2<P, NR>& a, const Affine<Fp2<Fp<P>, NR>>& p, bool negate) {
  bool suspect;
  Acc29x2<P, NR> r = madd29x2_fast<P, NR>(a, p, negate, &suspect);
#if defined(__HIP_DEVICE_COMPILE__)
  if (__ballot(suspect) != 0ull) {
    if (suspect) r = madd29x2_fixup<P, NR>(a, p, negate, r);
  }
#else
  if (suspect) r = madd29x2_fixup<P, NR>(a, p, negate, r);
#endif
  return r;
}

// The accumulator of k_accum for a coordinate field F: reduced radix for Fq (accum29.cuh) and Fq2 (here)
template <class F>
struct AccumOf;
template <class P>
struct AccumOf<Fp<P>> {
  using Acc = Acc29<P>;
  ZG_HD static Acc from_affine(const Affine<Fp<P>>& p, bool n) { return acc29_from_affine<P>(p, n); }
  ZG_HD static Acc madd(const Acc& a, const Affine<Fp<P>>& p, bool n) { return madd29<P>(a, p, n); }
  ZG_HD static XYZZ<Fp<P>> to_xyzz(const Acc& a) { return acc29_to_xyzz<P>(a); }
  ZG_HD static XYZZ<Fp<P>> to_part(const Acc& a) { return acc29_to_part<P>(a); }
};
template <class P, int NR>
struct AccumOf<Fp2<Fp<P>, NR>> {
  using Acc = Acc29x2<P, NR>;
  using F = Fp2<Fp<P>, NR>;
  ZG_HD static Acc from_affine(const Affine<F>& p, bool n) { return acc29x2_from_affine<P, NR>(p, n); }
  ZG_HD static Acc madd(const Acc& a, const Affine<F>& p, bool n) { return madd29x2<P, NR>(a, p, n); }
  ZG_HD static XYZZ<F> to_xyzz(const Acc& a) { return acc29x2_to_xyzz<P, NR>(a); }
};

}  // namespace zg
