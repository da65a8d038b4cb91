// Reduced-radix prime-field arithmetic for the bucket-accumulation kernel and the evaluation-domain transforms (gfx950).
//
// Why a second representation.  With full 32-bit limbs every 32x32 product fills its 64-bit result, so the sums of a
// Montgomery product need one carry instruction per multiply (mont_asm.h: 306 instructions for the 136 products of an
// 8-limb field; v_add_co / v_addc_co issue at the rate of v_mad_u64_u32, profiles/r1_peak_valu.jsonl).  Here an element
// is NL limbs of B < 32 bits (alt_bn128: 9 x 29, BLS12-377 Fq: 14 x 28) held in 32-bit registers.  A product of two
// limbs is 2B bits, so a 64-bit accumulator takes a whole column of the schoolbook product -- 2 NL products -- through
// v_mad_u64_u32's own 64-bit addend with no carry instruction at all; carries move once per column (one shift).
// Additions and subtractions are limb-wise with no carry chain either: limbs may grow past B bits ("lazy" values) and
// are brought back by a carry pass only where the next use needs it.
//
// Montgomery radix R' = 2^(NL B) (2^261, 2^392): R'/p is 169 (alt_bn128) and 38,996 (BLS12-377), so a product of any
// two values below about 13 p is again below 2 p and NO conditional subtraction is ever needed inside the kernel.
//
// Safety is checked by the compiler: a value's type E<K, VB, LMAX> carries VB (the value is < VB * p) and LMAX (every
// limb is <= LMAX); every operation static_asserts that its column sums stay below 2^64, its limbs below 2^32 and its
// operands within the Montgomery bound, and computes the bounds of its result.  Formulas are written with `auto`.
//
// Stored data stays in the library's one format (32-bit limbs, R = 2^(32 N): libff's mont_repr).  unpack() reads such an
// element shifted left by SHIFT = NL B - 32 N bits, which multiplies it by R'/R: the stored residue x R becomes x R'
// (below 2^SHIFT p) for free.  pack() multiplies by R mod p the Montgomery way (x R' -> x R), reduces to [0, p) and
// writes 32-bit limbs.
//
// Replaces (inside k_accum and the transforms' butterflies) the same libff Fp_model arithmetic field.cuh restates; results are the same field
// elements, so every MSM output stays bit-identical (tests/test_gpu_msm*.py; host build checked by tests/test_field29.py through zg_selftest_reduced_radix).
#pragma once
#include <stdint.h>

#include "field.cuh"

namespace zg {

// Parameters of the reduced-radix form of field P (curve_constants.h): limb count and width
template <class P>
struct Radix;
template <>
struct Radix<Bn254FqParams> {
  static constexpr int NL = 9, B = 29;
};
template <>
struct Radix<Bls377FqParams> {
  static constexpr int NL = 14, B = 28;
};
// the scalar fields (evaluation-domain transforms, ntt.cuh): 254 and 253 bits
template <>
struct Radix<Bn254FrParams> {
  static constexpr int NL = 9, B = 29;
};
template <>
struct Radix<Bls377FrParams> {
  static constexpr int NL = 9, B = 29;
};

template <class P>
struct K29 {
  using Params = P;
  static constexpr int N = P::N;  // 32-bit limbs of the stored form
  static constexpr int NL = Radix<P>::NL, B = Radix<P>::B;
  static constexpr int SHIFT = NL * B - 32 * N;
  static constexpr uint32_t MASK = (1u << B) - 1u;
  static constexpr uint64_t LNORM = MASK;  // limb bound of a normalised value
  static_assert(SHIFT >= 0 && SHIFT < B, "radix does not cover the stored form");

  struct Limbs {
    uint32_t v[NL];
  };
  // k * p in radix 2^B (k small)
  static constexpr Limbs times_p(uint64_t k) {
    Limbs r{};
    // p as a bit string from its 32-bit limbs
    unsigned __int128 carry = 0;
    for (int i = 0; i < NL; i++) {
      // limb i of p
      uint64_t limb = 0;
      const int bit0 = i * B;
      for (int b = 0; b < B; b++) {
        const int bit = bit0 + b;
        if (bit < 32 * N && ((P::P[bit / 32] >> (bit % 32)) & 1u)) limb |= 1ull << b;
      }
      const unsigned __int128 t = (unsigned __int128)limb * k + carry;
      r.v[i] = (uint32_t)((uint64_t)t & MASK);
      carry = t >> B;
    }
    r.v[NL - 1] += (uint32_t)((uint64_t)carry << B);  // the top limb takes what is left (k p < 2^(NL B) for the k used)
    return r;
  }
  static constexpr Limbs PL = times_p(1);
  // -p^-1 mod 2^B from -p^-1 mod 2^32
  static constexpr uint32_t INV = P::INV & MASK;
  static constexpr uint64_t top_of_p() { return PL.v[NL - 1]; }  // p >> (B (NL - 1))
  // floor(R' / p) - 1, conservatively from the top limb: p >= top * 2^(B (NL-1)), R' = 2^B * 2^(B (NL-1))
  static constexpr uint64_t mont_room() { return ((1ull << B) / (top_of_p() + 1)) - 1; }

  // does a column of the product fit 64 bits?  NL products of limbs (a top limb is bounded by the value: < V (p_top + 1))
  // + NL products m_i p_j + the carry from the column below
  static constexpr bool column_fits(uint64_t va, uint64_t la, uint64_t vb, uint64_t lb) {
    const uint64_t ta = va * (top_of_p() + 1), tb = vb * (top_of_p() + 1);
    const unsigned __int128 ea = la > ta ? la : ta, eb = lb > tb ? lb : tb;
    return (unsigned __int128)NL * ea * eb + (unsigned __int128)NL * LNORM * LNORM + ((unsigned __int128)1 << (64 - B)) <
           ((unsigned __int128)1 << 64);
  }

  // the general form: a column of the sum of several products; each product contributes NL col_term()s
  static constexpr unsigned __int128 col_term(uint64_t va, uint64_t la, uint64_t vb, uint64_t lb) {
    const uint64_t t = top_of_p() + 1;
    const unsigned __int128 ea = la > va * t ? la : va * t, eb = lb > vb * t ? lb : vb * t;
    return ea * eb;
  }
  static constexpr bool columns_fit(unsigned __int128 terms) {
    return (unsigned __int128)NL * terms + (unsigned __int128)NL * LNORM * LNORM + ((unsigned __int128)1 << (64 - B)) <
           ((unsigned __int128)1 << 64);
  }
  // the same for the sum of two products sharing their columns (mul_add)
  static constexpr bool column_fits2(uint64_t va, uint64_t la, uint64_t vb, uint64_t lb, uint64_t vc, uint64_t lc, uint64_t vd,
                                     uint64_t ld) {
    const uint64_t t = top_of_p() + 1;
    const unsigned __int128 ea = la > va * t ? la : va * t, eb = lb > vb * t ? lb : vb * t;
    const unsigned __int128 ec = lc > vc * t ? lc : vc * t, ed = ld > vd * t ? ld : vd * t;
    return (unsigned __int128)NL * (ea * eb + ec * ed) + (unsigned __int128)NL * LNORM * LNORM + ((unsigned __int128)1 << (64 - B)) <
           ((unsigned __int128)1 << 64);
  }

  // k p in the redundant form used by sub(): every limb below the top raised by c 2^B - c (limb 0 by c 2^B) against c
  // borrowed from the limb above, so that z_i >= c (2^B - 1) + (digit i of k p) for i < NL - 1
  static constexpr Limbs sub_const(uint64_t k, uint64_t c) {
    Limbs z = times_p(k);
    for (int i = 0; i < NL - 1; i++) z.v[i] += (uint32_t)(c << B) - (i ? (uint32_t)c : 0u);
    z.v[NL - 1] -= (uint32_t)c;
    return z;
  }
  // The constant for subtracting a value < vb p with limbs <= lb: the smallest (c, k) with every lower limb of
  // sub_const(k, c) >= lb and k >= vb + 1 (then its top limb covers the subtrahend's: (k - vb) p_top > c).  A limb bound
  // a few units above c (2^B - 1) -- the trace of a carry pass -- is met by the digits of k p themselves.
  struct SubPlan {
    uint64_t k, c;
  };
  static constexpr bool sub_const_covers(uint64_t k, uint64_t c, uint64_t lb) {
    const Limbs z = sub_const(k, c);
    for (int i = 0; i < NL - 1; i++)
      if (z.v[i] < lb) return false;
    return true;
  }
  static constexpr SubPlan sub_plan(uint64_t vb, uint64_t lb) {
    uint64_t c = lb / LNORM;
    if (c == 0) c = 1;
    for (;; c++)
      for (uint64_t k = vb + 1; k < vb + 5; k++)
        if (sub_const_covers(k, c, lb)) return SubPlan{k, c};
  }
};
template <class P>
constexpr typename K29<P>::Limbs K29<P>::PL;

// value < VB p, every limb <= LMAX
template <class P, uint64_t VB, uint64_t LMAX>
struct E29 {
  using K = K29<P>;
  static constexpr uint64_t vb = VB, lmax = LMAX;
  uint32_t v[K::NL];
};

namespace f29 {

#ifndef ZG_F29_ASM
#define ZG_F29_ASM 1
#endif
// acc += a * b in ONE v_mad_u64_u32 on the accumulator itself.  Written as an instruction because the compiler, given
// `acc += a * b` twice, splits a column over two accumulators and joins them with a 64-bit add per column (162 extra
// v_lshl_add_u64 per mixed addition); one chain per column is enough with two or more waves on a SIMD.
ZG_HD void mac(uint64_t& acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_F29_ASM
  uint64_t carry_out;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry_out) : "v"(a), "v"(b));
#else
  acc += (uint64_t)a * b;
#endif
}
// the same with a compile-time constant factor (a limb of p), held in a scalar register
ZG_HD void mac_const(uint64_t& acc, uint32_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_F29_ASM
  uint64_t carry_out;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry_out) : "v"(a), "s"(k));
#else
  acc += (uint64_t)a * k;
#endif
}

// Several products of one column in ONE statement.  The compiler puts a wait state (s_nop) behind every inline-asm
// statement whose result the next instruction reads (gfx940's dst_sel forwarding rule, applied to asm it cannot look
// into); v_mad_u64_u32 has no dst_sel, so a column's chain written as one statement needs none of them.
#ifndef ZG_F29_ASM_BLOCKS
#define ZG_F29_ASM_BLOCKS 1
#endif
ZG_HD void mac2(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_F29_ASM && ZG_F29_ASM_BLOCKS
  uint64_t co;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0" : "+v"(acc), "=s"(co) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
#else
  mac(acc, a0, b0);
  mac(acc, a1, b1);
#endif
}
ZG_HD void mac4(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_F29_ASM && ZG_F29_ASM_BLOCKS
  uint64_t co;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
      : "+v"(acc), "=s"(co)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
#else
  mac2(acc, a0, b0, a1, b1);
  mac2(acc, a2, b2, a3, b3);
#endif
}
ZG_HD void mac2_const(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_F29_ASM && ZG_F29_ASM_BLOCKS
  uint64_t co;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0" : "+v"(acc), "=&s"(co) : "v"(a0), "s"(k0), "v"(a1), "s"(k1));
#else
  mac_const(acc, a0, k0);
  mac_const(acc, a1, k1);
#endif
}
ZG_HD void mac4_const(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1, uint32_t a2, uint32_t k2, uint32_t a3, uint32_t k3) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_F29_ASM && ZG_F29_ASM_BLOCKS
  uint64_t co;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
      : "+v"(acc), "=&s"(co)  // early clobber: the first instruction writes it while later ones still read their k
      : "v"(a0), "s"(k0), "v"(a1), "s"(k1), "v"(a2), "s"(k2), "v"(a3), "s"(k3));
#else
  mac2_const(acc, a0, k0, a1, k1);
  mac2_const(acc, a2, k2, a3, k3);
#endif
}

// acc += sum over i in [I0, HI] of x[i] * y[K - i]  (one column's share of a product), in blocks of four.  The indices
// are template arguments: as loop variables the 14-limb field's columns were left with dynamic register indexing.
template <int I0, int HI, int K, int NLIMBS>
ZG_HD void mac_column(uint64_t& acc, const uint32_t (&x)[NLIMBS], const uint32_t (&y)[NLIMBS]) {
  constexpr int n = HI - I0 + 1;
  if constexpr (n >= 4) {
    mac4(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1], x[I0 + 2], y[K - I0 - 2], x[I0 + 3], y[K - I0 - 3]);
    mac_column<I0 + 4, HI, K, NLIMBS>(acc, x, y);
  } else if constexpr (n == 3) {
    mac2(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1]);
    mac(acc, x[I0 + 2], y[K - I0 - 2]);
  } else if constexpr (n == 2) {
    mac2(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1]);
  } else if constexpr (n == 1) {
    mac(acc, x[I0], y[K - I0]);
  }
}
template <int I0, int HI, int K, int NLIMBS>
ZG_HD void mac_column_const(uint64_t& acc, const uint32_t (&x)[NLIMBS], const uint32_t (&y)[NLIMBS]) {
  constexpr int n = HI - I0 + 1;
  if constexpr (n >= 4) {
    mac4_const(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1], x[I0 + 2], y[K - I0 - 2], x[I0 + 3], y[K - I0 - 3]);
    mac_column_const<I0 + 4, HI, K, NLIMBS>(acc, x, y);
  } else if constexpr (n == 3) {
    mac2_const(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1]);
    mac_const(acc, x[I0 + 2], y[K - I0 - 2]);
  } else if constexpr (n == 2) {
    mac2_const(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1]);
  } else if constexpr (n == 1) {
    mac_const(acc, x[I0], y[K - I0]);
  }
}

template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
ZG_HD auto add(const E29<P, VA, LA>& a, const E29<P, VB, LB>& b) {
  using K = K29<P>;
  static_assert(LA + LB < (1ull << 32), "add: limb overflow");
  E29<P, VA + VB, LA + LB> r;
#pragma unroll
  for (int i = 0; i < K::NL; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}

template <class P, uint64_t VA, uint64_t LA>
ZG_HD auto dbl(const E29<P, VA, LA>& a) {
  using K = K29<P>;
  static_assert(2 * LA < (1ull << 32), "dbl: limb overflow");
  E29<P, 2 * VA, 2 * LA> r;
#pragma unroll
  for (int i = 0; i < K::NL; i++) r.v[i] = a.v[i] << 1;
  return r;
}

// a - b + k p >= 0 limb by limb (k p in a redundant form whose limbs cover b's: K29::sub_plan)
template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
ZG_HD auto sub(const E29<P, VA, LA>& a, const E29<P, VB, LB>& b) {
  using K = K29<P>;
  constexpr typename K::SubPlan plan = K::sub_plan(VB, LB);
  constexpr uint64_t c = plan.c, k = plan.k;
  constexpr uint64_t zmax = K::LNORM + (c << K::B);
  static_assert(LA + zmax < (1ull << 32), "sub: limb overflow");
  static_assert((k + VA) * (K::top_of_p() + 1) + c < (1ull << 32), "sub: top limb overflow");
  static_assert(K::top_of_p() > c + 1, "sub: top limb of p too small to lend");
  constexpr typename K::Limbs z = K::sub_const(k, c);
  E29<P, VA + k, LA + zmax> r;
#pragma unroll
  for (int i = 0; i < K::NL; i++) r.v[i] = a.v[i] + z.v[i] - b.v[i];
  return r;
}

// k p - a  (a lazily negated value; same bounds as sub from zero)
template <class P, uint64_t VA, uint64_t LA>
ZG_HD auto neg(const E29<P, VA, LA>& a) {
  E29<P, 0, 0> z;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) z.v[i] = 0;
  return sub(z, a);
}

// one parallel carry pass: limbs come back to <= 2^B - 1 + (LMAX >> B); the value is unchanged
template <class P, uint64_t VA, uint64_t LA>
ZG_HD auto carry(const E29<P, VA, LA>& a) {
  using K = K29<P>;
  E29<P, VA, K::LNORM + (LA >> K::B)> r;
  r.v[0] = a.v[0] & K::MASK;
#pragma unroll
  for (int i = 1; i < K::NL - 1; i++) r.v[i] = (a.v[i] & K::MASK) + (a.v[i - 1] >> K::B);
  r.v[K::NL - 1] = a.v[K::NL - 1] + (a.v[K::NL - 2] >> K::B);
  return r;
}

// widen the static bounds (to give the two sides of a select one type)
template <uint64_t VB, uint64_t LB, class P, uint64_t VA, uint64_t LA>
ZG_HD E29<P, VB, LB> widen(const E29<P, VA, LA>& a) {
  static_assert(VB >= VA && LB >= LA, "widen: bounds may only grow");
  E29<P, VB, LB> r;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) r.v[i] = a.v[i];
  return r;
}

// flag ? b : a, limb-wise through a mask (v_bfi_b32; v_cndmask_b32 on vcc is five times slower on gfx950)
template <class P, uint64_t VA, uint64_t LA>
ZG_HD E29<P, VA, LA> select(const E29<P, VA, LA>& a, const E29<P, VA, LA>& b, bool flag) {
  const uint32_t m = flag ? 0xFFFFFFFFu : 0u;
  E29<P, VA, LA> r;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
  return r;
}

// Montgomery product a b / R' + (multiple of p), by columns: column k collects a_i b_j and m_i p_j with i + j = k in
// one 64-bit accumulator, m_k clears its low B bits, the rest moves to column k + 1.  Result < 2 p, limbs normalised.
template <class P, int K_, class A, class Bt>
ZG_HD void mul_columns_from(uint64_t& acc, const A& a, const Bt& b, uint32_t (&m)[K29<P>::NL], E29<P, 2, K29<P>::LNORM>& r) {
  using K = K29<P>;
  constexpr int NL = K::NL, B = K::B;
  // a_i b_(k-i), i in [max(0, k-NL+1), min(k, NL-1)];  m_i p_(k-i), i in [max(0, k-NL+1), min(k-1, NL-1)]
  constexpr int lo = K_ - NL + 1 > 0 ? K_ - NL + 1 : 0;
  mac_column<lo, (K_ < NL - 1 ? K_ : NL - 1), K_, NL>(acc, a.v, b.v);
  mac_column_const<lo, (K_ - 1 < NL - 1 ? K_ - 1 : NL - 1), K_, NL>(acc, m, K::PL.v);
  if constexpr (K_ < NL) {
    m[K_] = ((uint32_t)acc * K::INV) & K::MASK;
    mac_const(acc, m[K_], K::PL.v[0]);
  } else {
    r.v[K_ - NL] = (uint32_t)acc & K::MASK;
  }
  acc >>= B;
  if constexpr (K_ + 1 < 2 * NL - 1) mul_columns_from<P, K_ + 1>(acc, a, b, m, r);
}
template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
ZG_HD E29<P, 2, K29<P>::LNORM> mul(const E29<P, VA, LA>& a, const E29<P, VB, LB>& b) {
  using K = K29<P>;
  constexpr int NL = K::NL;
  // column sum: NL products of limbs (top limbs are value-bounded and far smaller) + NL products m_i p_j + carry-in
  static_assert(K::column_fits(VA, LA, VB, LB), "mul: a column overflows 64 bits -- carry() an operand first");
  static_assert(VA * VB <= K::mont_room(), "mul: operands too large for a result below 2 p");
  uint32_t m[NL];
  E29<P, 2, K::LNORM> r;
  uint64_t acc = 0;
  mul_columns_from<P, 0>(acc, a, b, m, r);
  r.v[NL - 1] = (uint32_t)acc;
  return r;
}

// a b + c d in ONE Montgomery reduction: the columns of both products share their accumulators, so the second product
// costs its NL^2 limb products but no second round of m_i p_j (a third of a product's multiplies).  Result < 2 p.
template <class P, int K_, class A, class Bt, class C, class D>
ZG_HD void mul_add_columns_from(uint64_t& acc, const A& a, const Bt& b, const C& c, const D& d, uint32_t (&m)[K29<P>::NL],
                                E29<P, 2, K29<P>::LNORM>& r) {
  using K = K29<P>;
  constexpr int NL = K::NL, B = K::B;
  constexpr int lo = K_ - NL + 1 > 0 ? K_ - NL + 1 : 0;
  constexpr int hi = K_ < NL - 1 ? K_ : NL - 1;
  mac_column<lo, hi, K_, NL>(acc, a.v, b.v);
  mac_column<lo, hi, K_, NL>(acc, c.v, d.v);
  mac_column_const<lo, (K_ - 1 < NL - 1 ? K_ - 1 : NL - 1), K_, NL>(acc, m, K::PL.v);
  if constexpr (K_ < NL) {
    m[K_] = ((uint32_t)acc * K::INV) & K::MASK;
    mac_const(acc, m[K_], K::PL.v[0]);
  } else {
    r.v[K_ - NL] = (uint32_t)acc & K::MASK;
  }
  acc >>= B;
  if constexpr (K_ + 1 < 2 * NL - 1) mul_add_columns_from<P, K_ + 1>(acc, a, b, c, d, m, r);
}
template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB, uint64_t VC, uint64_t LC, uint64_t VD, uint64_t LD>
ZG_HD E29<P, 2, K29<P>::LNORM> mul_add(const E29<P, VA, LA>& a, const E29<P, VB, LB>& b, const E29<P, VC, LC>& c,
                                       const E29<P, VD, LD>& d) {
  using K = K29<P>;
  constexpr int NL = K::NL;
  static_assert(K::column_fits2(VA, LA, VB, LB, VC, LC, VD, LD), "mul_add: a column overflows 64 bits -- carry() an operand first");
  static_assert(VA * VB + VC * VD <= K::mont_room(), "mul_add: operands too large for a result below 2 p");
  uint32_t m[NL];
  E29<P, 2, K::LNORM> r;
  uint64_t acc = 0;
  mul_add_columns_from<P, 0>(acc, a, b, c, d, m, r);
  r.v[NL - 1] = (uint32_t)acc;
  return r;
}

// The column loop of the Montgomery product with the limb products supplied by the caller: macs(chain, i, j) adds every
// term of weight 2^(B (i + j)) -- one limb product per operand pair -- to the column.  This is the form of the G2
// accumulator (accum29x2.cuh), whose registers leave room for ONE wave per SIMD: there a single dependent chain of
// v_mad_u64_u32 (and the wait states the assembler's hazard rules put around the instruction form of mac()) issues at a
// third of the multiplier's rate, so the products here are plain C, every operand pair has a chain of its own, joined
// once per column, and the compiler interleaves them (alt_bn128 G2, 2^22 points: k_accum 16.1 ms with the instruction
// form, 11.8 ms this way; the 32-bit-limb path 15.7 ms).
ZG_HD void mac_c(uint64_t& acc, uint32_t a, uint32_t b) { acc += (uint64_t)a * b; }
#ifndef ZG_F29_CHAINS
#define ZG_F29_CHAINS 1
#endif
template <class P, int NPAIRS, class MacFn>
ZG_HD E29<P, 2, K29<P>::LNORM> mont_columns(MacFn macs) {
  using K = K29<P>;
  constexpr int NL = K::NL, B = K::B;
  constexpr int NC = ZG_F29_CHAINS ? NPAIRS : 1;
  uint32_t m[NL];
  E29<P, 2, K::LNORM> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    uint64_t chain[NC];
    chain[0] = acc;
#pragma unroll
    for (int c = 1; c < NC; c++) chain[c] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j >= 0 && j < NL) macs(chain, i, j);
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (i < k && j >= 1 && j < NL) mac_c(chain[NC - 1], m[i], K::PL.v[j]);
    }
    acc = chain[0];
#pragma unroll
    for (int c = 1; c < NC; c++) acc += chain[c];
    if (k < NL) {
      m[k] = ((uint32_t)acc * K::INV) & K::MASK;
      mac_c(acc, m[k], K::PL.v[0]);
    } else {
      r.v[k - NL] = (uint32_t)acc & K::MASK;
    }
    acc >>= B;
  }
  r.v[NL - 1] = (uint32_t)acc;
  return r;
}

// a b in the same form (the Fq2 square's single products)
template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
ZG_HD E29<P, 2, K29<P>::LNORM> mul_chains(const E29<P, VA, LA>& a, const E29<P, VB, LB>& b) {
  using K = K29<P>;
  static_assert(K::column_fits(VA, LA, VB, LB), "mul: a column overflows 64 bits -- carry() an operand first");
  static_assert(VA * VB <= K::mont_room(), "mul: operands too large for a result below 2 p");
  return mont_columns<P, 1>([&](uint64_t* chain, int i, int j) { mac_c(chain[0], a.v[i], b.v[j]); });
}

// a b with the limb products on one chain and the m_i p_j products on another: two independent chains per column for the
// latency-bound quad kernels (curve_quad29.cuh), which run at one or two waves per SIMD
template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB>
ZG_HD E29<P, 2, K29<P>::LNORM> mul_chains2(const E29<P, VA, LA>& a, const E29<P, VB, LB>& b) {
  using K = K29<P>;
  static_assert(K::column_fits(VA, LA, VB, LB), "mul: a column overflows 64 bits -- carry() an operand first");
  static_assert(VA * VB <= K::mont_room(), "mul: operands too large for a result below 2 p");
  return mont_columns<P, 2>([&](uint64_t* chain, int i, int j) { mac_c(chain[0], a.v[i], b.v[j]); });
}

// a b + c d with a chain per product (mul_add above keeps one: best with two or more waves on a SIMD)
template <class P, uint64_t VA, uint64_t LA, uint64_t VB, uint64_t LB, uint64_t VC, uint64_t LC, uint64_t VD, uint64_t LD>
ZG_HD E29<P, 2, K29<P>::LNORM> mul_add_chains(const E29<P, VA, LA>& a, const E29<P, VB, LB>& b, const E29<P, VC, LC>& c,
                                              const E29<P, VD, LD>& d) {
  using K = K29<P>;
  static_assert(K::column_fits2(VA, LA, VB, LB, VC, LC, VD, LD), "mul_add: a column overflows 64 bits -- carry() an operand first");
  static_assert(VA * VB + VC * VD <= K::mont_room(), "mul_add: operands too large for a result below 2 p");
  return mont_columns<P, 2>([&](uint64_t* chain, int i, int j) {
    constexpr int S = ZG_F29_CHAINS ? 1 : 0;
    mac_c(chain[0], a.v[i], b.v[j]);
    mac_c(chain[1 * S], c.v[i], d.v[j]);
  });
}

// a1 b1 + a2 b2 + a3 b3 and a1 b1 + ... + a4 b4 in one reduction (the Fq2 formulas of accum29x2.cuh)
template <class P, uint64_t V1, uint64_t L1, uint64_t W1, uint64_t M1, uint64_t V2, uint64_t L2, uint64_t W2, uint64_t M2, uint64_t V3,
          uint64_t L3, uint64_t W3, uint64_t M3>
ZG_HD E29<P, 2, K29<P>::LNORM> mul_add3(const E29<P, V1, L1>& a1, const E29<P, W1, M1>& b1, const E29<P, V2, L2>& a2,
                                        const E29<P, W2, M2>& b2, const E29<P, V3, L3>& a3, const E29<P, W3, M3>& b3) {
  using K = K29<P>;
  static_assert(K::columns_fit(K::col_term(V1, L1, W1, M1) + K::col_term(V2, L2, W2, M2) + K::col_term(V3, L3, W3, M3)),
                "mul_add3: a column overflows 64 bits -- carry() an operand first");
  static_assert(V1 * W1 + V2 * W2 + V3 * W3 <= K::mont_room(), "mul_add3: operands too large for a result below 2 p");
  return mont_columns<P, 3>([&](uint64_t* chain, int i, int j) {
    constexpr int S = ZG_F29_CHAINS ? 1 : 0;
    mac_c(chain[0], a1.v[i], b1.v[j]);
    mac_c(chain[1 * S], a2.v[i], b2.v[j]);
    mac_c(chain[2 * S], a3.v[i], b3.v[j]);
  });
}
template <class P, uint64_t V1, uint64_t L1, uint64_t W1, uint64_t M1, uint64_t V2, uint64_t L2, uint64_t W2, uint64_t M2, uint64_t V3,
          uint64_t L3, uint64_t W3, uint64_t M3, uint64_t V4, uint64_t L4, uint64_t W4, uint64_t M4>
ZG_HD E29<P, 2, K29<P>::LNORM> mul_add4(const E29<P, V1, L1>& a1, const E29<P, W1, M1>& b1, const E29<P, V2, L2>& a2,
                                        const E29<P, W2, M2>& b2, const E29<P, V3, L3>& a3, const E29<P, W3, M3>& b3,
                                        const E29<P, V4, L4>& a4, const E29<P, W4, M4>& b4) {
  using K = K29<P>;
  static_assert(K::columns_fit(K::col_term(V1, L1, W1, M1) + K::col_term(V2, L2, W2, M2) + K::col_term(V3, L3, W3, M3) +
                               K::col_term(V4, L4, W4, M4)),
                "mul_add4: a column overflows 64 bits -- carry() an operand first");
  static_assert(V1 * W1 + V2 * W2 + V3 * W3 + V4 * W4 <= K::mont_room(), "mul_add4: operands too large for a result below 2 p");
  return mont_columns<P, 4>([&](uint64_t* chain, int i, int j) {
    constexpr int S = ZG_F29_CHAINS ? 1 : 0;
    mac_c(chain[0], a1.v[i], b1.v[j]);
    mac_c(chain[1 * S], a2.v[i], b2.v[j]);
    mac_c(chain[2 * S], a3.v[i], b3.v[j]);
    mac_c(chain[3 * S], a4.v[i], b4.v[j]);
  });
}

// a^2: the cross products once, against the doubled operand
template <class P, int K_, class A>
ZG_HD void sqr_columns_from(uint64_t& acc, const A& a, const uint32_t (&a2)[K29<P>::NL], uint32_t (&m)[K29<P>::NL],
                            E29<P, 2, K29<P>::LNORM>& r) {
  using K = K29<P>;
  constexpr int NL = K::NL, B = K::B;
  constexpr int lo = K_ - NL + 1 > 0 ? K_ - NL + 1 : 0;
  // 2 a_i a_(k-i) for i < k - i, then a_(k/2)^2 on even columns
  mac_column<lo, (K_ - 1) / 2 - (K_ == 0 ? 1 : 0), K_, NL>(acc, a2, a.v);
  if constexpr (K_ % 2 == 0) mac(acc, a.v[K_ / 2], a.v[K_ / 2]);
  mac_column_const<lo, (K_ - 1 < NL - 1 ? K_ - 1 : NL - 1), K_, NL>(acc, m, K::PL.v);
  if constexpr (K_ < NL) {
    m[K_] = ((uint32_t)acc * K::INV) & K::MASK;
    mac_const(acc, m[K_], K::PL.v[0]);
  } else {
    r.v[K_ - NL] = (uint32_t)acc & K::MASK;
  }
  acc >>= B;
  if constexpr (K_ + 1 < 2 * NL - 1) sqr_columns_from<P, K_ + 1>(acc, a, a2, m, r);
}
template <class P, uint64_t VA, uint64_t LA>
ZG_HD E29<P, 2, K29<P>::LNORM> sqr(const E29<P, VA, LA>& a) {
  using K = K29<P>;
  constexpr int NL = K::NL;
  static_assert(K::column_fits(VA, LA, VA, LA), "sqr: a column overflows 64 bits -- carry() the operand first");
  static_assert(VA * VA <= K::mont_room(), "sqr: operand too large for a result below 2 p");
  static_assert(2 * LA < (1ull << 32), "sqr: doubled limb overflow");
  uint32_t m[NL], a2[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
  E29<P, 2, K::LNORM> r;
  uint64_t acc = 0;
  sqr_columns_from<P, 0>(acc, a, a2, m, r);
  r.v[NL - 1] = (uint32_t)acc;
  return r;
}

// ---- conversions to and from the stored form (32-bit limbs, Montgomery R = 2^(32 N)) ----------------------------
// the stored integer shifted left by SH bits, read as limbs; value < 2^SH p (the stored integer is below p), limbs
// normalised
template <int SH, class P>
ZG_HD E29<P, (1ull << SH), K29<P>::LNORM> unpack_shifted(const Fp<P>& s) {
  using K = K29<P>;
  static_assert(SH >= 0 && SH <= K::SHIFT, "unpack: the shifted value must fit the limbs");
  E29<P, (1ull << SH), K::LNORM> r;
#pragma unroll
  for (int i = 0; i < K::NL; i++) {
    const int bit = i * K::B - SH;  // first bit of limb i in the stored integer (negative: zeros shifted in)
    uint32_t w;
    if (bit < 0) {
      w = s.v[0] << (-bit);
    } else {
      const int q = bit / 32, sh = bit % 32;
      const uint32_t lo = q < K::N ? s.v[q] : 0u;
      const uint32_t hi = q + 1 < K::N ? s.v[q + 1] : 0u;
      w = sh ? (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) : lo;
    }
    r.v[i] = w & K::MASK;
  }
  return r;
}
// stored x R  ->  x R' = (x R) 2^SHIFT: value < 2^SHIFT p
template <class P>
ZG_HD auto unpack(const Fp<P>& s) {
  return unpack_shifted<K29<P>::SHIFT>(s);
}
// the stored integer as it is (x R read as a value of this form: mul(y R', unpack_raw(x R)) = y x R, the stored form of
// the product); value < p
template <class P>
ZG_HD auto unpack_raw(const Fp<P>& s) {
  return unpack_shifted<0>(s);
}

// ---- product by a CONSTANT through its precomputed quotient ---------------------------------------------------------
// For a constant w in [0, p) the table holds w itself (plain integer) and wq = floor(w 2^(B (NL+1)) / p).  Then
//   q = floor(a wq / 2^(B (NL+1)))   -- the upper triangle of the limb product a x wq and two guard columns
//   r = a w - q p                    -- exact, 0 <= r < 3 p, so the low NL limbs of a w + q (2^(B NL) - p) are r itself
// 6 NL^2 / 4 + 3 NL limb products (144 for NL = 9) against the 2 NL^2 + NL (171) of the Montgomery product, no m_k = acc * inv
// step, and the three triangles are independent up to q.  The value's form is whatever a's is (a x R' stays x w R'): the
// transforms' butterflies multiply by twiddles only, which is what this is for (ntt.cuh).  q is at most two below the
// true quotient: one from wq's floor and a / 2^(B (NL+1)) < 1, one from the dropped columns, whose total is below
// NL LA 2^B / 2^B of a unit of column NL-1 -- far below the 2^(2B) that would reach q.
template <class P>
struct Shoup29 {
  uint32_t w[K29<P>::NL];      // the constant: plain integer in [0, p), normalised limbs
  uint32_t q[K29<P>::NL + 1];  // floor(w 2^(B (NL + 1)) / p), normalised limbs
};

// The table entry of the constant whose PLAIN value (not the Montgomery form) is wp: bit-serial division, for building
// tables (one lane per entry on the device, or on the host)
template <class P>
ZG_HD Shoup29<P> shoup_from_plain(const Fp<P>& wp) {
  using K = K29<P>;
  constexpr int N = P::N;
  static_assert((P::P[N - 1] >> 31) == 0, "2 rem must fit the words");
  Shoup29<P> s;
  const auto wl = unpack_raw(wp);
#pragma unroll
  for (int i = 0; i < K::NL; i++) s.w[i] = wl.v[i];
  for (int i = 0; i <= K::NL; i++) s.q[i] = 0;
  uint32_t rem[N];
  for (int i = 0; i < N; i++) rem[i] = wp.v[i];
  for (int bit = K::B * (K::NL + 1) - 1; bit >= 0; bit--) {
    uint32_t cy = 0;
    for (int i = 0; i < N; i++) {  // rem = 2 rem
      const uint32_t t = (rem[i] << 1) | cy;
      cy = rem[i] >> 31;
      rem[i] = t;
    }
    uint32_t d[N];
    uint32_t br = 0;
    for (int i = 0; i < N; i++) {  // d = rem - p
      const uint64_t x = (uint64_t)rem[i] - P::P[i] - br;
      d[i] = (uint32_t)x;
      br = (uint32_t)(x >> 63);
    }
    if (!br) {
      for (int i = 0; i < N; i++) rem[i] = d[i];
      s.q[bit / K::B] |= 1u << (bit % K::B);
    }
  }
  return s;
}

// acc += sum over i in [I0, HI] of x[i] * y[K - i], the arrays of different lengths
template <int I0, int HI, int K, int NX, int NY>
ZG_HD void mac_column_xy(uint64_t& acc, const uint32_t (&x)[NX], const uint32_t (&y)[NY]) {
  constexpr int n = HI - I0 + 1;
  if constexpr (n >= 4) {
    mac4(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1], x[I0 + 2], y[K - I0 - 2], x[I0 + 3], y[K - I0 - 3]);
    mac_column_xy<I0 + 4, HI, K, NX, NY>(acc, x, y);
  } else if constexpr (n == 3) {
    mac2(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1]);
    mac(acc, x[I0 + 2], y[K - I0 - 2]);
  } else if constexpr (n == 2) {
    mac2(acc, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1]);
  } else if constexpr (n == 1) {
    mac(acc, x[I0], y[K - I0]);
  }
}

template <class P>
struct ShoupConsts {
  using K = K29<P>;
  // 2^(B NL) - p as normalised limbs
  static constexpr typename K::Limbs neg_p() {
    typename K::Limbs r{};
    uint32_t cy = 1;
    for (int i = 0; i < K::NL; i++) {
      const uint32_t t = ((~K::PL.v[i]) & K::MASK) + cy;
      r.v[i] = t & K::MASK;
      cy = t >> K::B;
    }
    return r;
  }
  static constexpr typename K::Limbs NP = neg_p();
};
template <class P>
constexpr typename K29<P>::Limbs ShoupConsts<P>::NP;

// columns NL-1 .. 2 NL-1 of a x wq: the quotient's limbs
template <class P, int K_, class A>
ZG_HD void shoup_q_columns_from(uint64_t& acc, const A& a, const Shoup29<P>& c, uint32_t (&q)[K29<P>::NL]) {
  using K = K29<P>;
  constexpr int NL = K::NL, B = K::B;
  // a_i wq_(k-i): i <= NL-1, k - i <= NL
  constexpr int lo = K_ - NL > 0 ? K_ - NL : 0;
  mac_column_xy<lo, NL - 1, K_, NL, NL + 1>(acc, a.v, c.q);
  if constexpr (K_ >= NL + 1) q[K_ - NL - 1] = (uint32_t)acc & K::MASK;
  acc >>= B;
  if constexpr (K_ + 1 <= 2 * NL - 1) shoup_q_columns_from<P, K_ + 1>(acc, a, c, q);
}
// columns 0 .. NL-1 of a x w + q x (2^(B NL) - p)
template <class P, int K_, class A>
ZG_HD void shoup_r_columns_from(uint64_t& acc, const A& a, const Shoup29<P>& c, const uint32_t (&q)[K29<P>::NL],
                                E29<P, 3, K29<P>::LNORM>& r) {
  using K = K29<P>;
  constexpr int NL = K::NL, B = K::B;
  mac_column_xy<0, K_, K_, NL, NL>(acc, a.v, c.w);
  mac_column_const<0, K_, K_, NL>(acc, q, ShoupConsts<P>::NP.v);
  r.v[K_] = (uint32_t)acc & K::MASK;
  acc >>= B;
  if constexpr (K_ + 1 < NL) shoup_r_columns_from<P, K_ + 1>(acc, a, c, q, r);
}
template <class P, uint64_t VA, uint64_t LA>
ZG_HD E29<P, 3, K29<P>::LNORM> mul_shoup(const E29<P, VA, LA>& a, const Shoup29<P>& c) {
  using K = K29<P>;
  constexpr int NL = K::NL;
  constexpr uint64_t t = K::top_of_p() + 1;
  constexpr unsigned __int128 ea = LA > VA * t ? LA : VA * t;  // a's limbs, the top one bounded by the value
  constexpr unsigned __int128 qt = VA * t;                      // the quotient's top limb: q <= a w / p < a
  static_assert(VA < (1ull << K::B), "mul_shoup: a must be below 2^(B (NL+1))");
  static_assert(qt < ((unsigned __int128)1 << 32), "mul_shoup: the quotient's top limb overflows");
  static_assert((unsigned __int128)NL * ea * K::LNORM + (unsigned __int128)(NL - 1) * K::LNORM * K::LNORM + qt * K::LNORM +
                        ((unsigned __int128)1 << (64 - K::B)) < ((unsigned __int128)1 << 64),
                "mul_shoup: a column overflows 64 bits -- carry() the operand first");
  static_assert(3 * t <= (1ull << K::B), "mul_shoup: 3 p must fit NL limbs");
  uint32_t q[NL];
  uint64_t acc = 0;
  shoup_q_columns_from<P, NL - 1>(acc, a, c, q);
  q[NL - 1] = (uint32_t)acc;
  E29<P, 3, K::LNORM> r;
  acc = 0;
  shoup_r_columns_from<P, 0>(acc, a, c, q, r);
  return r;
}

// R mod p (plain; the factor that takes x R' back to x R through mul()) and R' mod p (the Montgomery one of this form)
// as normalised limbs
template <class P>
struct Consts29 {
  using K = K29<P>;
  static constexpr typename K::Limbs from_words(const uint32_t* w) {
    typename K::Limbs r{};
    for (int i = 0; i < K::NL; i++) {
      uint64_t limb = 0;
      for (int b = 0; b < K::B; b++) {
        const int bit = i * K::B + b;
        if (bit < 32 * K::N && ((w[bit / 32] >> (bit % 32)) & 1u)) limb |= 1ull << b;
      }
      r.v[i] = (uint32_t)limb;
    }
    return r;
  }
  // 2 a mod p for a < p (normalised limbs)
  static constexpr typename K::Limbs dbl_mod_p(typename K::Limbs a) {
    typename K::Limbs d{}, e{};
    uint32_t cy = 0;
    for (int i = 0; i < K::NL; i++) {
      const uint32_t t = (a.v[i] << 1) + cy;
      d.v[i] = i < K::NL - 1 ? (t & K::MASK) : t;
      cy = i < K::NL - 1 ? (t >> K::B) : 0;
    }
    int64_t br = 0;
    for (int i = 0; i < K::NL; i++) {
      const int64_t x = (int64_t)d.v[i] - (int64_t)K::PL.v[i] + br;
      if (i < K::NL - 1) {
        e.v[i] = (uint32_t)((uint64_t)x & K::MASK);
        br = x < 0 ? -1 : 0;
      } else {
        e.v[i] = (uint32_t)x;
        br = x < 0 ? -1 : 0;
      }
    }
    return br ? d : e;
  }
  static constexpr typename K::Limbs one_prime() {
    typename K::Limbs r = from_words(P::R1);
    for (int i = 0; i < K::SHIFT; i++) r = dbl_mod_p(r);
    return r;
  }
  static constexpr typename K::Limbs R_MOD_P = from_words(P::R1);
  static constexpr typename K::Limbs ONE_PRIME = one_prime();
};
template <class P>
constexpr typename K29<P>::Limbs Consts29<P>::R_MOD_P;
template <class P>
constexpr typename K29<P>::Limbs Consts29<P>::ONE_PRIME;

// R mod p (plain integer): mul(x R', this) = x R
template <class P>
ZG_HD E29<P, 1, K29<P>::LNORM> r_conv() {
  E29<P, 1, K29<P>::LNORM> r;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) r.v[i] = Consts29<P>::R_MOD_P.v[i];
  return r;
}
// R' mod p: the Montgomery one of this form; mul(v, one()) = v brought below 2 p
template <class P>
ZG_HD E29<P, 1, K29<P>::LNORM> one() {
  E29<P, 1, K29<P>::LNORM> r;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) r.v[i] = Consts29<P>::ONE_PRIME.v[i];
  return r;
}
template <class P>
ZG_HD E29<P, 0, 0> zero() {
  E29<P, 0, 0> r;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) r.v[i] = 0;
  return r;
}

// t < 2 p with normalised limbs -> the representative in [0, p)
template <class P>
ZG_HD E29<P, 1, K29<P>::LNORM> canon(const E29<P, 2, K29<P>::LNORM>& t) {
  using K = K29<P>;
  // d = t - p with a borrow chain on B-bit limbs; keep t when it borrows
  uint32_t d[K::NL];
  int32_t br = 0;
#pragma unroll
  for (int i = 0; i < K::NL; i++) {
    const int32_t x = (int32_t)t.v[i] - (int32_t)K::PL.v[i] + br;
    if (i < K::NL - 1) {
      d[i] = (uint32_t)x & K::MASK;
      br = x >> K::B;  // 0 or -1
    } else {
      d[i] = (uint32_t)x;
      br = x >> 31;
    }
  }
  const uint32_t keep = (uint32_t)br;  // all ones when t < p
  E29<P, 1, K::LNORM> r;
#pragma unroll
  for (int i = 0; i < K::NL; i++) r.v[i] = (t.v[i] & keep) | (d[i] & ~keep);
  return r;
}

// -a for a in [0, p): p - a, or 0
template <class P>
ZG_HD E29<P, 1, K29<P>::LNORM> neg_canon(const E29<P, 1, K29<P>::LNORM>& a) {
  using K = K29<P>;
  uint32_t nz = 0;
  E29<P, 1, K::LNORM> r;
  int32_t br = 0;
#pragma unroll
  for (int i = 0; i < K::NL; i++) {
    nz |= a.v[i];
    const int32_t x = (int32_t)K::PL.v[i] - (int32_t)a.v[i] + br;
    r.v[i] = i < K::NL - 1 ? ((uint32_t)x & K::MASK) : (uint32_t)x;
    br = x >> K::B;
  }
  const uint32_t m = nz ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int i = 0; i < K::NL; i++) r.v[i] &= m;
  return r;
}

// a value in [0, p) as the 32-bit words of the stored form
template <class P>
ZG_HD Fp<P> to_words(const E29<P, 1, K29<P>::LNORM>& a) {
  using K = K29<P>;
  uint32_t l[K::NL + 1];
#pragma unroll
  for (int i = 0; i < K::NL; i++) l[i] = a.v[i];
  l[K::NL] = 0;
  Fp<P> r;
#pragma unroll
  for (int j = 0; j < K::N; j++) {
    const int bit = 32 * j;
    const int q = bit / K::B, sh = bit % K::B;
    uint64_t w = (uint64_t)l[q] >> sh;
    if (q + 1 <= K::NL) w |= (uint64_t)l[q + 1] << (K::B - sh);
    if (q + 2 <= K::NL && 2 * K::B - sh < 32) w |= (uint64_t)l[q + 2] << (2 * K::B - sh);
    r.v[j] = (uint32_t)w;
  }
  return r;
}

// x R' (any value the Montgomery bound admits against 1 p) -> stored x R in [0, p)
template <class P, uint64_t VA, uint64_t LA>
ZG_HD Fp<P> pack(const E29<P, VA, LA>& a) {
  return to_words(canon(mul(a, r_conv<P>())));
}

// The same limbs under narrower static bounds that the CALLER has established by an argument the types cannot carry
// (ntt.cuh: a bound that grows with the number of stages already run); every use states that argument.
template <uint64_t VB, uint64_t LB, class P, uint64_t VA, uint64_t LA>
ZG_HD E29<P, VB, LB> assume_bounds(const E29<P, VA, LA>& a) {
  E29<P, VB, LB> r;
#pragma unroll
  for (int i = 0; i < K29<P>::NL; i++) r.v[i] = a.v[i];
  return r;
}

// value (< 2 p, normalised limbs) congruent to zero?  i.e. equal to 0 or to p
template <class P>
ZG_HD bool is_zero_mod_p(const E29<P, 2, K29<P>::LNORM>& a) {
  using K = K29<P>;
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < K::NL; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ K::PL.v[i];
  }
  return z == 0 || e == 0;
}

}  // namespace f29
}  // namespace zg
