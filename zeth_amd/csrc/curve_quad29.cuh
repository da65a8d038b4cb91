// The quad-lane point arithmetic of curve_quad.cuh in the reduced-radix field arithmetic of field29.cuh, for the
// prime-field curves (G1): the products of k_combine_*, k_reduce1 and k_reduce2 take 2 NL^2 multiply instructions and no
// carry instruction (field29.cuh) instead of the 32-bit-limb product's two instructions per limb product, and the
// differences between them are limb-wise with no conditional subtraction.
//
// Same formulas, same lane layout and the same exceptional cases as curve_quad.cuh (EFD add-2008-s / dbl-2008-s-1), so
// every sum is the same group element.  What differs:
//   * a coordinate is a lazy value of ONE static type, Quad29<P>::C (value < 10 p, limbs one carry pass from normalised),
//     which both formulas reproduce on their outputs (checked by the types: every product static_asserts its bounds);
//   * "is this value zero" needs the value below 2 p: the ZZ coordinate always is (a product, a loaded element or the
//     zero element), and whether the two inputs share their x is read off ZZ3 = ZZ1 ZZ2 (U2 - U1)^2 after the fact, as
//     madd29 does; the rare branch then normalises S2 - S1 to tell a doubling from a cancellation;
//   * the buffers these kernels exchange (k_accum's unit sums `part`, the segment sums `seg`) hold x R' mod p (R' =
//     2^(NL B), this form's own Montgomery radix) in the 32-bit words of a field element, so a load is a re-slicing of
//     bits and only a store pays a product (by one, to come below 2 p).  The rows the host reads are written in the
//     library's stored form x R (store_final).
#pragma once
#include "curve_quad.cuh"
#include "field29.cuh"

namespace zg {

#ifndef ZG_QUAD_RADIX29
#define ZG_QUAD_RADIX29 1
#endif

// the product of these kernels: 0 the instruction form (one chain per column: best at two or more waves per SIMD),
// 1 plain C (the compiler splits a column as it sees fit), 2 plain C with the reduction's products on a chain of their own
#ifndef ZG_QUAD29_MUL
#define ZG_QUAD29_MUL 0
#endif

template <class P>
struct Quad29 {
  using K = K29<P>;
  static constexpr uint64_t LN = K::LNORM;
  using N2 = E29<P, 2, LN>;
  using C = E29<P, 10, LN + 15>;

  template <class A, class Bt>
  ZG_HD static N2 mul(const A& a, const Bt& b) {
#if ZG_QUAD29_MUL == 1
    return f29::mul_chains(a, b);
#elif ZG_QUAD29_MUL == 2
    return f29::mul_chains2(a, b);
#else
    return f29::mul(a, b);
#endif
  }
  template <class A>
  ZG_HD static N2 sqr(const A& a) {
#if ZG_QUAD29_MUL == 0
    return f29::sqr(a);
#else
    return mul(a, a);
#endif
  }
  template <class E>
  ZG_HD static C can(const E& e) {
    return f29::widen<C::vb, C::lmax>(e);
  }
  ZG_HD static C zero() {
    C r;
#pragma unroll
    for (int i = 0; i < K::NL; i++) r.v[i] = 0;
    return r;
  }
  // take_b ? b : a under the wider of the two static bounds
  template <uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
  ZG_HD static auto sel(bool take_b, const E29<P, VA, LA>& a, const E29<P, VB, LB>& b) {
    constexpr uint64_t V = VA > VB ? VA : VB, L = LA > LB ? LA : LB;
    return f29::select(f29::widen<V, L>(a), f29::widen<V, L>(b), take_b);
  }
  // congruent to zero, for a value that is below 2 p with normalised limbs -- true of the ZZ coordinate of every point
  // these kernels hold (a product, a loaded element in [0, p), or zero), which is the only lane whose answer is used
  ZG_HD static bool zz_zero(const C& c) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < K::NL; i++) {
      z |= c.v[i];
      e |= c.v[i] ^ K::PL.v[i];
    }
    return z == 0 || e == 0;
  }

  // this lane's coordinate of 2 * (the quad's point)
  ZG_HD static C dbl(const C& c, unsigned q) {
    const bool inf = quad_bcast_flag(zz_zero(c), 2) != 0;
    // round 1: lane 0 X^2, lane 1 Y^2
    const N2 m1 = sqr(c);
    const N2 YY = quad_perm<quad_bcast(1)>(m1), XX = quad_perm<quad_bcast(0)>(m1);
    const auto V = f29::carry(f29::dbl(f29::dbl(YY)));         // U^2, U = 2 Y
    const auto M = f29::carry(f29::add(f29::dbl(XX), XX));     // 3 X^2
    const auto U = f29::carry(f29::dbl(c));                    // meaningful on lane 1
    // round 2: X V, U V, ZZ V, M M
    const auto a2 = sel(q == 3, sel(q == 1, c, U), M);
    const auto b2 = sel(q == 3, V, M);
    const N2 m2 = mul(a2, b2);  // S, W, ZZ3, MM
    // round 3: -, M (S - X3), W Y, W ZZZ
    const N2 S = quad_perm<quad_bcast(0)>(m2), W = quad_perm<quad_bcast(1)>(m2);
    const auto X3 = f29::carry(f29::sub(quad_perm<quad_bcast(3)>(m2), f29::dbl(S)));
    const auto a3 = sel(q == 1, W, M);
    const auto b3 = sel(q == 3, sel(q == 1, quad_perm<quad_bcast(1)>(c), f29::carry(f29::sub(S, X3))), c);
    const N2 m3 = mul(a3, b3);  // -, t, u, ZZZ3
    const auto Y3 = f29::carry(f29::sub(m3, quad_perm<quad_bcast(2)>(m3)));
    const C r = can(sel(q == 3, sel(q == 2, sel(q == 1, X3, Y3), m2), m3));
    return f29::select(r, c, inf);
  }

  // this lane's coordinate of P1 + P2
  ZG_HD static C add(const C& c1, const C& c2, unsigned q) {
    const bool inf1 = quad_bcast_flag(zz_zero(c1), 2) != 0;
    const bool inf2 = quad_bcast_flag(zz_zero(c2), 2) != 0;
    // round 1: X1 ZZ2, ZZ1 X2, Y1 ZZZ2, ZZZ1 Y2
    const N2 m1 = mul(quad_perm<quad_ctrl(0, 2, 1, 3)>(c1), quad_perm<quad_ctrl(2, 0, 3, 1)>(c2));  // U1, U2, S1, S2
    // P on lanes 0, 1; R on lanes 2, 3
    const auto d = f29::carry(f29::sub(quad_perm<quad_ctrl(1, 1, 3, 3)>(m1), quad_perm<quad_ctrl(0, 0, 2, 2)>(m1)));
    // round 2: P P, ZZ1 ZZ2, R R, ZZZ1 ZZZ2
    const bool odd = (q & 1u) != 0;
    const auto a2 = sel(odd, d, quad_perm<quad_ctrl(0, 2, 0, 3)>(c1));
    const auto b2 = sel(odd, d, quad_perm<quad_ctrl(0, 2, 0, 3)>(c2));
    const N2 m2 = mul(a2, b2);  // PP, ZZ12, RR, ZZZ12
    // round 3: PP times P, U1, ZZ12
    const N2 PP = quad_perm<quad_bcast(0)>(m2);
    const auto o3 = sel(q >= 2, sel(q == 1, d, quad_perm<quad_bcast(0)>(m1)), quad_perm<quad_bcast(1)>(m2));
    const N2 m3 = mul(PP, o3);  // PPP, Q, ZZ3, (unused)
    // round 4: -, R (Q - X3), S1 PPP, ZZZ12 PPP
    const N2 PPP = quad_perm<quad_bcast(0)>(m3), Q = quad_perm<quad_bcast(1)>(m3);
    const auto X3 = f29::carry(f29::sub(quad_perm<quad_bcast(2)>(m2), f29::add(PPP, f29::dbl(Q))));
    const auto a4 = sel(q >= 2, quad_perm<quad_bcast(2)>(d), sel(q == 3, m1, m2));
    const auto b4 = sel(q >= 2, f29::carry(f29::sub(Q, X3)), PPP);
    const N2 m4 = mul(a4, b4);  // -, t, u, ZZZ3
    const auto Y3 = f29::carry(f29::sub(m4, quad_perm<quad_bcast(2)>(m4)));
    C r = can(sel(q == 3, sel(q == 2, sel(q == 1, X3, Y3), m3), m4));
    // exceptional cases.  ZZ3 = ZZ1 ZZ2 P^2 is congruent to zero exactly when an input is the identity or P is: with both
    // inputs finite the points share their x -- the same point (R = 0: double it) or opposite points (the identity)
    const bool same_x = quad_bcast_flag(f29::is_zero_mod_p(m3), 2) != 0 && !inf1 && !inf2;
    if (quad_wave_any(same_x)) {  // rare: entered by the wave, computed by every quad, kept where it applies
      const bool r_zero = quad_bcast_flag(f29::is_zero_mod_p(mul(d, f29::one<P>())), 2) != 0;
      const C dd = dbl(c1, q);
      r = f29::select(r, f29::select(zero(), dd, r_zero), same_x);
    }
    r = f29::select(r, c2, inf1);
    r = f29::select(r, c1, inf2);
    return r;
  }

  // a coordinate in a buffer of these kernels: x R' mod p in [0, p), as 32-bit words
  ZG_HD static C load(const XYZZ<Fp<P>>* p, unsigned q) {
    return can(f29::unpack_raw(reinterpret_cast<const Fp<P>*>(p)[q]));
  }
  ZG_HD static void store(XYZZ<Fp<P>>* p, unsigned q, const C& c) {
    reinterpret_cast<Fp<P>*>(p)[q] = f29::to_words(f29::canon(f29::mul(c, f29::one<P>())));
  }
  // the library's stored form (x R), for what leaves these kernels
  ZG_HD static void store_final(XYZZ<Fp<P>>* p, unsigned q, const C& c) {
    reinterpret_cast<Fp<P>*>(p)[q] = f29::pack(c);
  }
};

// The operations the quad kernels are written against (msm.cuh chooses per coordinate field): 32-bit limbs
// (curve_quad.cuh) or the reduced radix
template <class F>
struct QuadOps32 {
  static constexpr bool rprime_buffers = false;
  using Coord = F;
  ZG_HD static Coord zero() { return F::zero(); }
  ZG_HD static Coord add(const Coord& a, const Coord& b, unsigned q) { return quad_add(a, b, q); }
  ZG_HD static Coord dbl(const Coord& a, unsigned q) { return quad_dbl(a, q); }
  ZG_HD static Coord load(const XYZZ<F>* p, unsigned q) { return quad_load(p, q); }
  ZG_HD static void store(XYZZ<F>* p, unsigned q, const Coord& c) { quad_store(p, q, c); }
  ZG_HD static void store_final(XYZZ<F>* p, unsigned q, const Coord& c) { quad_store(p, q, c); }
};
template <class P>
struct QuadOps29 {
  static constexpr bool rprime_buffers = true;
  using Q = Quad29<P>;
  using Coord = typename Q::C;
  ZG_HD static Coord zero() { return Q::zero(); }
  ZG_HD static Coord add(const Coord& a, const Coord& b, unsigned q) { return Q::add(a, b, q); }
  ZG_HD static Coord dbl(const Coord& a, unsigned q) { return Q::dbl(a, q); }
  ZG_HD static Coord load(const XYZZ<Fp<P>>* p, unsigned q) { return Q::load(p, q); }
  ZG_HD static void store(XYZZ<Fp<P>>* p, unsigned q, const Coord& c) { Q::store(p, q, c); }
  ZG_HD static void store_final(XYZZ<Fp<P>>* p, unsigned q, const Coord& c) { Q::store_final(p, q, c); }
};

}  // namespace zg
