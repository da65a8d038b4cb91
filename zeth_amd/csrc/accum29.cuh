// The bucket accumulator of k_accum in the reduced-radix arithmetic of field29.cuh: an XYZZ point whose coordinates
// are lazy E29 values, and the mixed addition acc += (+-)P for an affine base P read in the stored form.
//
// Same group law as curve.cuh (EFD madd-2008-s, 8M + 2S; libff's mixed addition inside multi_exp_with_mixed_addition,
// SURVEY.md 3.1 step 5c) -- only the representation of the field elements differs, and the result is converted back
// to the stored form (acc29_to_xyzz) before anything else reads it.
//
// The straight-line formula runs for every lane.  Its exceptional inputs -- an identity base, an identity accumulator,
// equal or opposite points -- all leave ZZ3 = ZZ1 * (U2 - X1)^2 congruent to zero (or are flagged from the base's
// words), so ONE test after the fact finds them and a cold fix-up, entered by the whole wave on a ballot, redoes those
// lanes from the old accumulator.
#pragma once
#include "curve.cuh"
#include "field29.cuh"

namespace zg {

template <class P>
struct Acc29 {
  using K = K29<P>;
  static constexpr uint64_t LN = K::LNORM, TWO_B = 1ull << K::B;
  // bounds that the mixed addition reproduces on its own output (checked by the types below)
  using X = E29<P, 9, LN + 5>;               // a carried difference of products
  using Y = E29<P, 5, LN + LN + TWO_B>;      // an uncarried difference of two products
  using Z = E29<P, 2, LN>;                   // products
  X x;
  Y y;
  Z zz, zzz;
};

template <class P>
ZG_HD Acc29<P> acc29_inf() {
  Acc29<P> a;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) a.x.v[i] = a.y.v[i] = a.zz.v[i] = a.zzz.v[i] = 0;
  return a;
}

template <class F>
ZG_HD bool affine_words_zero(const Affine<F>& p) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) o |= p.x.v[i] | p.y.v[i];
  return o == 0;
}

// (+-)y of a base, from the stored form: value < (2^SHIFT + 1) p
template <class P>
ZG_HD auto base_y29(const Fp<P>& y, bool negate) {
  const auto yp = f29::unpack(y);
  const auto yn = f29::neg(yp);
  using T = decltype(yn);
  return f29::select(f29::widen<T::vb, T::lmax>(yp), yn, negate);
}

// the accumulator holding the single point (+-)P; P = identity gives the identity
template <class P>
ZG_HD Acc29<P> acc29_from_affine(const Affine<Fp<P>>& p, bool negate) {
  using A = Acc29<P>;
  if (affine_words_zero(p)) return acc29_inf<P>();
  const auto o = f29::one<P>();
  A a;
  a.x = f29::widen<A::X::vb, A::X::lmax>(f29::mul(f29::unpack(p.x), o));
  a.y = f29::widen<A::Y::vb, A::Y::lmax>(f29::mul(base_y29<P>(p.y, negate), o));
  a.zz = f29::widen<2, A::LN>(o);
  a.zzz = a.zz;
  return a;
}

template <class P>
ZG_HD Acc29<P> acc29_from_xyzz(const XYZZ<Fp<P>>& s) {
  using A = Acc29<P>;
  const auto o = f29::one<P>();
  A a;
  a.x = f29::widen<A::X::vb, A::X::lmax>(f29::mul(f29::unpack(s.x), o));
  a.y = f29::widen<A::Y::vb, A::Y::lmax>(f29::mul(f29::unpack(s.y), o));
  a.zz = f29::mul(f29::unpack(s.zz), o);
  a.zzz = f29::mul(f29::unpack(s.zzz), o);
  if (s.is_inf()) return acc29_inf<P>();
  return a;
}

// back to the stored form, every coordinate in [0, p)
template <class P>
ZG_HD XYZZ<Fp<P>> acc29_to_xyzz(const Acc29<P>& a) {
  XYZZ<Fp<P>> r;
  r.x = f29::pack(a.x);
  r.y = f29::pack(a.y);
  r.zz = f29::pack(a.zz);
  r.zzz = f29::pack(a.zzz);
  return r;
}

// x R' mod p in [0, p) in the words of a field element: the form of the buffers the quad kernels exchange
// (curve_quad29.cuh), where a load is then free of any product
template <class P>
ZG_HD XYZZ<Fp<P>> acc29_to_part(const Acc29<P>& a) {
  const auto o = f29::one<P>();
  XYZZ<Fp<P>> r;
  r.x = f29::to_words(f29::canon(f29::mul(a.x, o)));
  r.y = f29::to_words(f29::canon(f29::mul(a.y, o)));
  r.zz = f29::to_words(f29::canon(f29::mul(a.zz, o)));
  r.zzz = f29::to_words(f29::canon(f29::mul(a.zzz, o)));
  return r;
}

// 2 (+-)P for an affine P != identity (EFD mdbl-2008-s-1, a = 0)
template <class P>
ZG_HD Acc29<P> mdbl29(const Affine<Fp<P>>& p, bool negate) {
  using A = Acc29<P>;
  const auto o = f29::one<P>();
  const auto x = f29::mul(f29::unpack(p.x), o);
  const auto y = f29::mul(base_y29<P>(p.y, negate), o);
  const auto U = f29::dbl(y);
  const auto V = f29::sqr(U);
  const auto W = f29::mul(U, V);
  const auto S = f29::mul(x, V);
  const auto XX = f29::sqr(x);
  const auto M = f29::carry(f29::add(f29::dbl(XX), XX));
  const auto X3 = f29::carry(f29::sub(f29::sqr(M), f29::dbl(S)));
  const auto Y3 = f29::sub(f29::mul(M, f29::sub(S, X3)), f29::mul(W, y));
  A r;
  r.x = f29::widen<A::X::vb, A::X::lmax>(X3);
  r.y = f29::widen<A::Y::vb, A::Y::lmax>(Y3);
  r.zz = V;
  r.zzz = W;
  return r;
}

// acc + (+-)P by the straight-line formula.  *suspect is set when the result cannot be trusted: P is the identity, or
// the low limb of ZZ3 is that of a value congruent to zero (ZZ3 < 2 p: 0 or p) -- which every exceptional input
// produces, and an ordinary one with probability 2^(1-B).
template <class P>
ZG_HD Acc29<P> madd29_fast(const Acc29<P>& a, const Affine<Fp<P>>& p, bool negate, bool* suspect) {
  using A = Acc29<P>;
  using K = K29<P>;
  const auto U2 = f29::mul(f29::unpack(p.x), a.zz);
  const auto S2 = f29::mul(base_y29<P>(p.y, negate), a.zzz);
  const auto Pd = f29::carry(f29::sub(U2, a.x));
  const auto R = f29::carry(f29::sub(S2, a.y));
  const auto PP = f29::sqr(Pd);
  const auto PPP = f29::mul(Pd, PP);
  const auto Q = f29::mul(a.x, PP);
  const auto X3 = f29::carry(f29::sub(f29::sqr(R), f29::add(PPP, f29::dbl(Q))));
  // Y3 = R (Q - X3) - Y1 PPP as R (Q - X3) + (k p - Y1) PPP: two products, one reduction
  const auto Y3 = f29::mul_add(R, f29::sub(Q, X3), f29::carry(f29::neg(a.y)), PPP);
  A r;
  r.x = f29::widen<A::X::vb, A::X::lmax>(X3);
  r.y = f29::widen<A::Y::vb, A::Y::lmax>(Y3);
  r.zz = f29::mul(a.zz, PP);
  r.zzz = f29::mul(a.zzz, PPP);
  *suspect = affine_words_zero(p) || r.zz.v[0] == 0 || r.zz.v[0] == K::PL.v[0];
  return r;
}

// the exceptional cases, for the lanes madd29_fast flagged: `fast` is kept when the flag was a false alarm.
// Inlined although it runs for a wave once in thousands of additions: as a call it takes the accumulator's address (or
// returns through memory), and the compiler then keeps the loop-carried accumulator in scratch memory -- a store of all
// of it per addition, measured at 12,100 cycles per addition against 10,000 without.
template <class P>
ZG_HD Acc29<P> madd29_fixup(const Acc29<P>& a, const Affine<Fp<P>>& p, bool negate, const Acc29<P>& fast) {
  if (affine_words_zero(p)) return a;                                    // + identity
  if (f29::is_zero_mod_p(a.zz)) return acc29_from_affine<P>(p, negate);  // identity + P
  if (!f29::is_zero_mod_p(fast.zz)) return fast;                         // false alarm
  // U2 == X1: the same point (R == 0: double it) or opposite points (the identity)
  const auto S2 = f29::mul(base_y29<P>(p.y, negate), a.zzz);
  const auto R = f29::mul(f29::carry(f29::sub(S2, a.y)), f29::one<P>());
  if (f29::is_zero_mod_p(R)) return mdbl29<P>(p, negate);
  return acc29_inf<P>();
}

// acc += (+-)P, every case.  On the device the fix-up is entered by the whole wave when any lane needs it.
template <class P>
ZG_HD Acc29<P> madd29(const Acc29<P>& a, const Affine<Fp<P>>& p, bool negate) {
  bool suspect;
  Acc29<P> r = madd29_fast<P>(a, p, negate, &suspect);
#if defined(__HIP_DEVICE_COMPILE__)
  if (__ballot(suspect) != 0ull) {
    if (suspect) r = madd29_fixup<P>(a, p, negate, r);
  }
#else
  if (suspect) r = madd29_fixup<P>(a, p, negate, r);
#endif
  return r;
}

}  // namespace zg
