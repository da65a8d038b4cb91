// Multi-limb Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (and the host tail).
//
// Replaces libff's Fp_model / Fp2_model (absent from /root/reference: depends/libsnark is an empty
// submodule; usage sites libzeth/mpc/groth16/powersoftau_utils.tcc:21-34 show mont_repr / Rsquared).
// R = 2^(32*N) equals libff's R = 2^(64*ceil(bits/64)) for every field used here, so an element's
// Montgomery limbs are byte-identical to libff's in-memory mont_repr on a little-endian host.
//
// The multiplier is a CIOS loop in the "no-carry" form (valid because the top limb of every modulus is
// < 2^31 - 1): per outer step one v_mad_u64_u32 for a[j]*b[i] and one for m*p[j], carries kept in the
// upper half of the 64-bit accumulators.  All loops are compile-time unrolled; the modulus limbs fold
// to literals.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve_constants.h"
#include "mont_asm.h"

#ifndef ZG_ASM_FIELD
#define ZG_ASM_FIELD 1  // gfx950 inline-assembly multiplier / carry chains on the device (mont_asm.h)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define ZG_HD __host__ __device__ __forceinline__
#define ZG_HD_COLD __host__ __device__ __noinline__
#else
#define ZG_HD __host__ __device__ inline
#define ZG_HD_COLD __host__ __device__ inline
#endif

namespace zg {

template <class P>
struct Fp {
  static constexpr int N = P::N;
  using Params = P;
  uint32_t v[N];

  ZG_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
  }
  ZG_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::R1[i];
    return r;
  }
  ZG_HD static Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = P::R2[i];
    return r;
  }
  ZG_HD static Fp raw_one() {
    Fp r = zero();
    r.v[0] = 1;
    return r;
  }
  ZG_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i];
    return o == 0;
  }
  ZG_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
  ZG_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // s in [0, 2p) -> [0, p)
  ZG_HD static Fp reduce_once(const Fp& s) {
    Fp d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)s.v[i] - P::P[i] - br;
      d.v[i] = (uint32_t)t;
      br = (t >> 32) & 1u;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = br ? s.v[i] : d.v[i];
    return r;
  }
  ZG_HD static Fp add(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_ASM_FIELD
    if constexpr (N == 8 || N == 12) {
      uint32_t p[N];
#pragma unroll
      for (int i = 0; i < N; i++) p[i] = P::P[i];
      Fp r;
      if constexpr (N == 8) zg_mont_add_8(r.v, a.v, b.v, p);
      else zg_mont_add_12(r.v, a.v, b.v, p);
      return r;
    }
#endif
    Fp s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      c += (uint64_t)a.v[i] + b.v[i];
      s.v[i] = (uint32_t)c;
      c >>= 32;
    }
    return reduce_once(s);
  }
  ZG_HD static Fp dbl(const Fp& a) { return add(a, a); }
  ZG_HD static Fp sub(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_ASM_FIELD
    if constexpr (N == 8 || N == 12) {
      uint32_t p[N];
#pragma unroll
      for (int i = 0; i < N; i++) p[i] = P::P[i];
      Fp r;
      if constexpr (N == 8) zg_mont_sub_8(r.v, a.v, b.v, p);
      else zg_mont_sub_12(r.v, a.v, b.v, p);
      return r;
    }
#endif
    Fp d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
      d.v[i] = (uint32_t)t;
      br = (t >> 32) & 1u;
    }
    // add p back when the subtraction borrowed
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      c += (uint64_t)d.v[i] + (P::P[i] & mask);
      d.v[i] = (uint32_t)c;
      c >>= 32;
    }
    return d;
  }
  ZG_HD static Fp neg(const Fp& a) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_ASM_FIELD
    if constexpr (N == 8 || N == 12) return cneg(a, true);
#endif
    Fp d;
    uint64_t br = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      nz |= a.v[i];
      uint64_t t = (uint64_t)P::P[i] - a.v[i] - br;
      d.v[i] = (uint32_t)t;
      br = (t >> 32) & 1u;
    }
    uint32_t mask = nz ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int i = 0; i < N; i++) d.v[i] &= mask;
    return d;
  }
  // conditional negate (used for signed-digit bucket entries)
  ZG_HD static Fp cneg(const Fp& a, bool flag) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_ASM_FIELD
    if constexpr (N == 8 || N == 12) {
      uint32_t p[N];
#pragma unroll
      for (int i = 0; i < N; i++) p[i] = P::P[i];
      Fp r;
      const uint32_t mask = flag ? 0xFFFFFFFFu : 0u;
      if constexpr (N == 8) zg_mont_cneg_8(r.v, a.v, p, mask);
      else zg_mont_cneg_12(r.v, a.v, p, mask);
      return r;
    }
#endif
    Fp n = neg(a);
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = flag ? n.v[i] : a.v[i];
    return r;
  }

  ZG_HD static Fp mul(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__) && ZG_ASM_FIELD
    if constexpr (N == 8 || N == 12) {
      uint32_t p[N];
#pragma unroll
      for (int i = 0; i < N; i++) p[i] = P::P[i];
      Fp r;
      if constexpr (N == 8) zg_mont_mul_8(r.v, a.v, b.v, p, P::INV);
      else zg_mont_mul_12(r.v, a.v, b.v, p, P::INV);
      return r;
    }
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
    // host tail: the same Montgomery product (same R = 2^(32 N)) on 64-bit limbs with unsigned __int128
    if constexpr (N % 2 == 0) {
      constexpr int M = N / 2;
      typedef unsigned __int128 u128;
      uint64_t A[M], B[M], Pm[M], t64[M + 2];
      for (int i = 0; i < M; i++) {
        A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
        Pm[i] = (uint64_t)P::P[2 * i] | ((uint64_t)P::P[2 * i + 1] << 32);
      }
      for (int i = 0; i < M + 2; i++) t64[i] = 0;
      for (int i = 0; i < M; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < M; j++) {
          u128 x = (u128)A[j] * B[i] + t64[j] + carry;
          t64[j] = (uint64_t)x;
          carry = (uint64_t)(x >> 64);
        }
        u128 x = (u128)t64[M] + carry;
        t64[M] = (uint64_t)x;
        t64[M + 1] = (uint64_t)(x >> 64);
        const uint64_t m = t64[0] * P::INV64;
        x = (u128)m * Pm[0] + t64[0];
        carry = (uint64_t)(x >> 64);
        for (int j = 1; j < M; j++) {
          x = (u128)m * Pm[j] + t64[j] + carry;
          t64[j - 1] = (uint64_t)x;
          carry = (uint64_t)(x >> 64);
        }
        x = (u128)t64[M] + carry;
        t64[M - 1] = (uint64_t)x;
        t64[M] = t64[M + 1] + (uint64_t)(x >> 64);
      }
      Fp r;
      for (int i = 0; i < M; i++) {
        r.v[2 * i] = (uint32_t)t64[i];
        r.v[2 * i + 1] = (uint32_t)(t64[i] >> 32);
      }
      return reduce_once(r);  // t < 2p < R, so t64[M] == 0
    }
#endif
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t bi = b.v[i];
      uint64_t A = (uint64_t)a.v[0] * bi + t[0];
      const uint32_t t0 = (uint32_t)A;
      A >>= 32;
      const uint32_t m = t0 * P::INV;
      uint64_t C = (uint64_t)m * P::P[0] + t0;
      C >>= 32;
#pragma unroll
      for (int j = 1; j < N; j++) {
        A += (uint64_t)a.v[j] * bi + t[j];
        const uint32_t tj = (uint32_t)A;
        A >>= 32;
        C += (uint64_t)m * P::P[j] + tj;
        t[j - 1] = (uint32_t)C;
        C >>= 32;
      }
      t[N - 1] = (uint32_t)(C + A);
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return reduce_once(r);
  }
  ZG_HD static Fp sqr(const Fp& a) { return mul(a, a); }

  ZG_HD static Fp to_mont(const Fp& plain) { return mul(plain, r2()); }
  ZG_HD static Fp from_mont(const Fp& m) { return mul(m, raw_one()); }

  // small-constant multiples
  ZG_HD static Fp mul3(const Fp& a) { return add(dbl(a), a); }

  // a^e for a little-endian limb exponent (variable time; host tail and setup kernels only)
  ZG_HD_COLD static Fp pow_limbs(const Fp& a, const uint32_t* e, int nlimbs) {
    Fp r = one();
    bool started = false;
    for (int i = nlimbs - 1; i >= 0; i--) {
      for (int b = 31; b >= 0; b--) {
        if (started) r = sqr(r);
        if ((e[i] >> b) & 1u) {
          r = started ? mul(r, a) : a;
          started = true;
        }
      }
    }
    return r;
  }
  // Fermat inversion a^(p-2); inv(0) = 0
  ZG_HD_COLD static Fp inv(const Fp& a) {
    uint32_t e[N];
#pragma unroll
    for (int i = 0; i < N; i++) e[i] = P::PM2[i];
    return pow_limbs(a, e, N);
  }
};

// Fq2 = Fq[u] / (u^2 + NEG_NR), NEG_NR = 1 (alt_bn128) or 5 (BLS12-377); element c0 + c1 u.
template <class F, int NEG_NR>
struct Fp2 {
  using Base = F;
  F c0, c1;

  ZG_HD static Fp2 zero() { return Fp2{F::zero(), F::zero()}; }
  ZG_HD static Fp2 one() { return Fp2{F::one(), F::zero()}; }
  ZG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZG_HD bool operator==(const Fp2& b) const { return c0 == b.c0 && c1 == b.c1; }
  ZG_HD bool operator!=(const Fp2& b) const { return !(*this == b); }
  ZG_HD static Fp2 add(const Fp2& a, const Fp2& b) { return Fp2{F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
  ZG_HD static Fp2 sub(const Fp2& a, const Fp2& b) { return Fp2{F::sub(a.c0, b.c0), F::sub(a.c1, b.c1)}; }
  ZG_HD static Fp2 dbl(const Fp2& a) { return Fp2{F::dbl(a.c0), F::dbl(a.c1)}; }
  ZG_HD static Fp2 neg(const Fp2& a) { return Fp2{F::neg(a.c0), F::neg(a.c1)}; }
  ZG_HD static Fp2 cneg(const Fp2& a, bool f) { return Fp2{F::cneg(a.c0, f), F::cneg(a.c1, f)}; }
  ZG_HD static Fp2 mul3(const Fp2& a) { return add(dbl(a), a); }
  // x * NEG_NR (the caller subtracts)
  ZG_HD static F mul_neg_nr(const F& x) {
    if (NEG_NR == 1) return x;
    F x2 = F::dbl(x);
    F x4 = F::dbl(x2);
    return F::add(x4, x);  // NEG_NR == 5
  }
  ZG_HD static Fp2 mul(const Fp2& a, const Fp2& b) {
    F v0 = F::mul(a.c0, b.c0);
    F v1 = F::mul(a.c1, b.c1);
    F s = F::mul(F::add(a.c0, a.c1), F::add(b.c0, b.c1));
    Fp2 r;
    r.c0 = F::sub(v0, mul_neg_nr(v1));
    r.c1 = F::sub(F::sub(s, v0), v1);
    return r;
  }
  ZG_HD static Fp2 sqr(const Fp2& a) {
    // c0 = a0^2 - k a1^2 = (a0 + a1)(a0 - k a1) - a0 a1 + k a0 a1 ; c1 = 2 a0 a1   (k = NEG_NR)
    F v0 = F::mul(a.c0, a.c1);
    F t = F::mul(F::add(a.c0, a.c1), F::sub(a.c0, mul_neg_nr(a.c1)));
    Fp2 r;
    r.c0 = F::add(F::sub(t, v0), mul_neg_nr(v0));
    r.c1 = F::dbl(v0);
    return r;
  }
  ZG_HD_COLD static Fp2 inv(const Fp2& a) {
    // 1/(a0 + a1 u) = (a0 - a1 u) / (a0^2 + k a1^2)
    F n = F::add(F::sqr(a.c0), mul_neg_nr(F::sqr(a.c1)));
    F ni = F::inv(n);
    return Fp2{F::mul(a.c0, ni), F::neg(F::mul(a.c1, ni))};
  }
};

using Bn254Fq = Fp<Bn254FqParams>;
using Bn254Fr = Fp<Bn254FrParams>;
using Bls377Fq = Fp<Bls377FqParams>;
using Bls377Fr = Fp<Bls377FrParams>;
using Bn254Fq2 = Fp2<Bn254Fq, 1>;
using Bls377Fq2 = Fp2<Bls377Fq, 5>;

}  // namespace zg
