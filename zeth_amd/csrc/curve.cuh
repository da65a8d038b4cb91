// Short-Weierstrass group law y^2 = x^3 + b (a = 0) in extended Jacobian ("XYZZ") coordinates,
// generic over the coordinate field (Fq for G1, Fq2 for G2).
//
// Restates the group operations libff's alt_bn128_G1/G2 and bls12_377_G1/G2 provide to
// libff::multi_exp* (absent submodule; Zeth's call sites: libzeth/core/multi_exp.tcc:13-35).
// libff uses Jacobian (X,Y,Z); the coordinates here differ but every result is normalised to the
// unique affine point before it leaves the library, so outputs are representation-independent.
//
// Affine storage: X || Y in Montgomery form, the identity is encoded as X = Y = 0 (not on the curve
// because b != 0).  XYZZ: x = X/ZZ, y = Y/ZZZ, identity <=> ZZ == 0.
#pragma once
#include "field.cuh"

namespace zg {

template <class F>
struct Affine {
  F x, y;
  ZG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  ZG_HD static Affine inf() { return Affine{F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  ZG_HD bool is_inf() const { return zz.is_zero(); }
  ZG_HD static XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
  ZG_HD static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return XYZZ{p.x, p.y, F::one(), F::one()};
  }
};

// 2 * (affine p), p != identity  (EFD mdbl-2008-s-1, a = 0)
template <class F>
ZG_HD XYZZ<F> xyzz_mdbl(const Affine<F>& p) {
  F U = F::dbl(p.y);
  F V = F::sqr(U);
  F W = F::mul(U, V);
  F S = F::mul(p.x, V);
  F M = F::mul3(F::sqr(p.x));
  XYZZ<F> r;
  r.x = F::sub(F::sqr(M), F::dbl(S));
  r.y = F::sub(F::mul(M, F::sub(S, r.x)), F::mul(W, p.y));
  r.zz = V;
  r.zzz = W;
  return r;
}

// 2 * a  (EFD dbl-2008-s-1, a = 0).  The _inl forms are inlined into the latency-bound reduction kernels, where a
// call (arguments through scratch memory) about doubles the time of each step of a dependent chain; everywhere
// else the out-of-line forms keep the code small.
template <class F>
ZG_HD XYZZ<F> xyzz_dbl_inl(const XYZZ<F>& a) {
  if (a.is_inf()) return a;
  F U = F::dbl(a.y);
  F V = F::sqr(U);
  F W = F::mul(U, V);
  F S = F::mul(a.x, V);
  F M = F::mul3(F::sqr(a.x));
  XYZZ<F> r;
  r.x = F::sub(F::sqr(M), F::dbl(S));
  r.y = F::sub(F::mul(M, F::sub(S, r.x)), F::mul(W, a.y));
  r.zz = F::mul(V, a.zz);
  r.zzz = F::mul(W, a.zzz);
  return r;
}
template <class F>
ZG_HD_COLD XYZZ<F> xyzz_dbl(const XYZZ<F>& a) {
  return xyzz_dbl_inl(a);
}

// acc += affine p  (EFD madd-2008-s; 8M + 2S), all exceptional cases handled
template <class F>
ZG_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& p) {
  if (p.is_inf()) return;
  if (acc.is_inf()) {
    acc = XYZZ<F>{p.x, p.y, F::one(), F::one()};
    return;
  }
  F U2 = F::mul(p.x, acc.zz);
  F S2 = F::mul(p.y, acc.zzz);
  F Pd = F::sub(U2, acc.x);
  F R = F::sub(S2, acc.y);
  if (Pd.is_zero()) {
    if (R.is_zero()) {
      acc = xyzz_mdbl(p);
    } else {
      acc = XYZZ<F>::inf();
    }
    return;
  }
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

// acc += b  (EFD add-2008-s; 12M + 2S)
template <class F>
ZG_HD_COLD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& b);
template <class F>
ZG_HD void xyzz_add_inl(XYZZ<F>& acc, const XYZZ<F>& b) {
  if (b.is_inf()) return;
  if (acc.is_inf()) {
    acc = b;
    return;
  }
  F U1 = F::mul(acc.x, b.zz);
  F U2 = F::mul(b.x, acc.zz);
  F S1 = F::mul(acc.y, b.zzz);
  F S2 = F::mul(b.y, acc.zzz);
  F Pd = F::sub(U2, U1);
  F R = F::sub(S2, S1);
  if (Pd.is_zero()) {
    if (R.is_zero()) {
      acc = xyzz_dbl(acc);  // out of line: rare
    } else {
      acc = XYZZ<F>::inf();
    }
    return;
  }
  F PP = F::sqr(Pd);
  F PPP = F::mul(Pd, PP);
  F Q = F::mul(U1, PP);
  F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
  F Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
  acc.x = X3;
  acc.y = Y3;
  acc.zz = F::mul(F::mul(acc.zz, b.zz), PP);
  acc.zzz = F::mul(F::mul(acc.zzz, b.zzz), PPP);
}
template <class F>
ZG_HD_COLD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& b) {
  xyzz_add_inl(acc, b);
}

template <class F>
ZG_HD XYZZ<F> xyzz_neg(const XYZZ<F>& a) {
  return XYZZ<F>{a.x, F::neg(a.y), a.zz, a.zzz};
}

// k * a for a small unsigned k (double-and-add from the top bit)
template <class F>
ZG_HD_COLD XYZZ<F> xyzz_mul_small(const XYZZ<F>& a, uint32_t k) {
  XYZZ<F> r = XYZZ<F>::inf();
  for (int b = 31; b >= 0; b--) {
    r = xyzz_dbl(r);
    if ((k >> b) & 1u) xyzz_add(r, a);
  }
  return r;
}

// k * a for a little-endian limb scalar (plain form)
template <class F>
ZG_HD_COLD XYZZ<F> xyzz_mul_limbs(const XYZZ<F>& a, const uint32_t* k, int nlimbs) {
  XYZZ<F> r = XYZZ<F>::inf();
  for (int i = nlimbs - 1; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      r = xyzz_dbl(r);
      if ((k[i] >> b) & 1u) xyzz_add(r, a);
    }
  }
  return r;
}

template <class F>
ZG_HD_COLD Affine<F> xyzz_to_affine(const XYZZ<F>& a) {
  if (a.is_inf()) return Affine<F>::inf();
  // one inversion: zz^3 = zzz^2, so 1/z = zz / zzz and 1/zz = (1/z)^2
  F izzz = F::inv(a.zzz);
  F iz = F::mul(a.zz, izzz);
  F izz = F::sqr(iz);
  return Affine<F>{F::mul(a.x, izz), F::mul(a.y, izzz)};
}

}  // namespace zg
