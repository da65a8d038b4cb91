// Point arithmetic spread over the four lanes of a quad, for the latency-bound kernels of the MSM (k_combine_*,
// k_reduce1, k_reduce2): chains of dependent FULL additions in which one lane per point leaves most of the device idle
// and pays the 14 field products of an addition one after another.
//
// Lane q of a quad (q = lane & 3) holds ONE coordinate of an XYZZ point -- q = 0: X, 1: Y, 2: ZZ, 3: ZZZ -- so a point in
// memory (four consecutive field elements) is read and written by its quad as one contiguous run.  An addition
// (EFD add-2008-s, the formula of xyzz_add in curve.cuh) is then four rounds of ONE field product per lane, the operands
// moved between the lanes with DPP quad permutes; a doubling (dbl-2008-s-1) is three rounds.  Same group law, same
// exceptional cases, so every sum is the same group element as the one-lane code's.
//
//   add   round 1  U1 = X1 ZZ2      U2 = X2 ZZ1      S1 = Y1 ZZZ2     S2 = Y2 ZZZ1
//         round 2  PP = P^2         ZZ12 = ZZ1 ZZ2   RR = R^2         ZZZ12 = ZZZ1 ZZZ2       P = U2 - U1, R = S2 - S1
//         round 3  PPP = P PP       Q = U1 PP        ZZ3 = ZZ12 PP    --
//         round 4  --               t = R (Q - X3)   u = S1 PPP       ZZZ3 = ZZZ12 PPP        X3 = RR - PPP - 2 Q
//         result   X3               t - u            ZZ3              ZZZ3
//   dbl   round 1  XX = X^2         V = U^2          --               --                      U = 2 Y
//         round 2  S = X V          W = U V          ZZ3 = V ZZ       MM = M^2                M = 3 XX
//         round 3  --               t = M (S - X3)   u = W Y          ZZZ3 = W ZZZ            X3 = MM - 2 S
//         result   X3               t - u            ZZ3              ZZZ3
//
// Every function here must be called by all four lanes of a quad together (quad-uniform control flow).
#pragma once
#include <hip/hip_runtime.h>

#include "curve.cuh"

#if !defined(__HIP_DEVICE_COMPILE__)
#include <atomic>
#include <cstring>
#include <thread>
#endif

// Every cross-lane move stays an instruction of its own (an empty asm statement on its result).  Left alone, the compiler
// folds a move into the subtraction that consumes it (v_subrev_u32_dpp ... quad_perm), which the limb-wise differences
// of curve_quad29.cuh invite: with those folded instructions the alt_bn128 kernels returned wrong sums on gfx950, while
// the host build of the same templates, and the device build with the moves kept apart, agree with the one-lane
// formulas (zg_selftest_quad_lanes / zg_selftest_quad_lanes_device).
#ifndef ZG_QUAD_DPP_STANDALONE
#define ZG_QUAD_DPP_STANDALONE 1
#endif

namespace zg {

#if !defined(__HIP_DEVICE_COMPILE__)
// Host stand-in for the cross-lane moves, so that the CPU self-test (selftest.hip) runs the very templates the kernels
// instantiate: the four lanes of ONE quad are four host threads that meet at every exchange.
struct QuadHost {
  uint32_t slot[4][64];
  std::atomic<unsigned> arrived{0}, phase{0};
  void barrier() {
    const unsigned ph = phase.load();
    if (arrived.fetch_add(1) == 3) {
      arrived.store(0);
      phase.store(ph + 1);
    } else {
      while (phase.load() == ph) std::this_thread::yield();
    }
  }
  template <class T>
  T exchange(const T& v, int lane, int from) {
    static_assert(sizeof(T) <= sizeof(slot[0]), "exchange slot too small");
    memcpy(slot[lane], &v, sizeof(T));
    barrier();
    T r;
    memcpy(&r, slot[from], sizeof(T));
    barrier();
    return r;
  }
};
inline QuadHost*& quad_host() {
  static QuadHost* h = nullptr;
  return h;
}
inline int& quad_host_lane() {
  thread_local int lane = 0;
  return lane;
}
#endif

// lane i of every quad receives v from lane (CTRL >> 2 i) & 3 of its quad
template <int CTRL, class F>
ZG_HD F quad_perm(const F& v) {
  static_assert(sizeof(F) % 4 == 0, "field elements are arrays of 32-bit words");
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = sizeof(F) / 4;
  F r;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
  for (int i = 0; i < NW; i++) {
    d[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)s[i], CTRL, 0xF, 0xF, true);
#if ZG_QUAD_DPP_STANDALONE
    asm volatile("" : "+v"(d[i]));
#endif
  }
  return r;
#else
  const int lane = quad_host_lane();
  return quad_host()->exchange(v, lane, (CTRL >> (2 * lane)) & 3);
#endif
}
constexpr int quad_ctrl(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int quad_bcast(int a) { return quad_ctrl(a, a, a, a); }

ZG_HD int quad_bcast_flag(bool f, int lane) {
  const int v = f ? 1 : 0;
#if !defined(__HIP_DEVICE_COMPILE__)
  return quad_host()->exchange(v, quad_host_lane(), lane);
#else
  int r;
  switch (lane) {
    case 0: r = __builtin_amdgcn_mov_dpp(v, quad_bcast(0), 0xF, 0xF, true); break;
    case 1: r = __builtin_amdgcn_mov_dpp(v, quad_bcast(1), 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_mov_dpp(v, quad_bcast(2), 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_mov_dpp(v, quad_bcast(3), 0xF, 0xF, true); break;
  }
#if ZG_QUAD_DPP_STANDALONE
  asm volatile("" : "+v"(r));  // kept an instruction of its own, like the moves of quad_perm
#endif
  return r;
#endif
}
// true when the flag is set on any lane of the wave (host: of the quad): the condition of a branch that every quad must
// take together because the code behind it exchanges data between lanes
ZG_HD bool quad_wave_any(bool f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(f) != 0ull;
#else
  bool any = false;
  for (int l = 0; l < 4; l++) any |= quad_host()->exchange((int)f, quad_host_lane(), l) != 0;
  return any;
#endif
}

template <class F>
ZG_HD F fsel(bool take_b, const F& a, const F& b) {
  constexpr int NW = sizeof(F) / 4;
  F r;
  const uint32_t* pa = reinterpret_cast<const uint32_t*>(&a);
  const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
  const uint32_t m = take_b ? 0xFFFFFFFFu : 0u;  // through a mask (v_bfi_b32): v_cndmask_b32 is several times slower on gfx950
#pragma unroll
  for (int i = 0; i < NW; i++) d[i] = (pb[i] & m) | (pa[i] & ~m);
  return r;
}

// this lane's coordinate of 2 * (the quad's point); q = lane & 3
template <class F>
ZG_HD F quad_dbl(const F& c, unsigned q) {
  const bool inf = quad_bcast_flag(c.is_zero(), 2) != 0;  // ZZ == 0
  const F U = F::dbl(c);                                  // meaningful on lane 1: 2 Y
  // round 1: lane 0 X X, lane 1 U U
  const F a1 = fsel(q == 1, c, U);
  const F m1 = F::mul(a1, a1);
  // round 2
  const F V = quad_perm<quad_bcast(1)>(m1);
  const F M = F::mul3(quad_perm<quad_bcast(0)>(m1));
  const F a2 = fsel(q == 3, fsel(q == 1, c, U), M);  // X, U, ZZ, M
  const F b2 = fsel(q == 3, V, M);
  const F m2 = F::mul(a2, b2);  // S, W, ZZ3, MM
  // round 3
  const F S = quad_perm<quad_bcast(0)>(m2);
  const F X3 = F::sub(quad_perm<quad_bcast(3)>(m2), F::dbl(S));
  const F W = quad_perm<quad_bcast(1)>(m2);
  const F a3 = fsel(q == 1, W, M);                                                  // -, M, W, W
  const F b3 = fsel(q == 3, fsel(q == 1, quad_perm<quad_bcast(1)>(c), F::sub(S, X3)), c);  // -, S - X3, Y, ZZZ
  const F m3 = F::mul(a3, b3);  // -, t, u, ZZZ3
  const F Y3 = F::sub(m3, quad_perm<quad_bcast(2)>(m3));
  const F r = fsel(q == 3, fsel(q == 2, fsel(q == 1, X3, Y3), m2), m3);
  return fsel(inf, r, c);
}

// this lane's coordinate of P1 + P2 (c1, c2: this lane's coordinates of the two points)
template <class F>
ZG_HD F quad_add(const F& c1, const F& c2, unsigned q) {
  const bool inf1 = quad_bcast_flag(c1.is_zero(), 2) != 0;
  const bool inf2 = quad_bcast_flag(c2.is_zero(), 2) != 0;
  // round 1: X1 ZZ2, ZZ1 X2, Y1 ZZZ2, ZZZ1 Y2
  const F g1 = quad_perm<quad_ctrl(0, 2, 1, 3)>(c1);
  const F g2 = quad_perm<quad_ctrl(2, 0, 3, 1)>(c2);
  const F m1 = F::mul(g1, g2);  // U1, U2, S1, S2
  // P on lanes 0, 1; R on lanes 2, 3
  const F d = F::sub(quad_perm<quad_ctrl(1, 1, 3, 3)>(m1), quad_perm<quad_ctrl(0, 0, 2, 2)>(m1));
  const bool p_zero = quad_bcast_flag(d.is_zero(), 0) != 0;
  const bool r_zero = quad_bcast_flag(d.is_zero(), 2) != 0;
  // round 2: P P, ZZ1 ZZ2, R R, ZZZ1 ZZZ2
  const bool odd = (q & 1u) != 0;
  const F a2 = fsel(odd, d, quad_perm<quad_ctrl(0, 2, 0, 3)>(c1));
  const F b2 = fsel(odd, d, quad_perm<quad_ctrl(0, 2, 0, 3)>(c2));
  const F m2 = F::mul(a2, b2);  // PP, ZZ12, RR, ZZZ12
  // round 3: PP times P, U1, ZZ12
  const F PP = quad_perm<quad_bcast(0)>(m2);
  const F o3 = fsel(q >= 2, fsel(q == 1, d, quad_perm<quad_bcast(0)>(m1)), quad_perm<quad_bcast(1)>(m2));
  const F m3 = F::mul(PP, o3);  // PPP, Q, ZZ3, (unused)
  // round 4
  const F PPP = quad_perm<quad_bcast(0)>(m3);
  const F Q = quad_perm<quad_bcast(1)>(m3);
  const F X3 = F::sub(F::sub(quad_perm<quad_bcast(2)>(m2), PPP), F::dbl(Q));
  const F a4 = fsel(q >= 2, quad_perm<quad_bcast(2)>(d), fsel(q == 3, m1, m2));  // -, R, S1, ZZZ12
  const F b4 = fsel(q >= 2, F::sub(Q, X3), PPP);
  const F m4 = F::mul(a4, b4);  // -, t, u, ZZZ3
  const F Y3 = F::sub(m4, quad_perm<quad_bcast(2)>(m4));
  F r = fsel(q == 3, fsel(q == 2, fsel(q == 1, X3, Y3), m3), m4);
  // exceptional cases
  if (p_zero && !inf1 && !inf2) {  // same x: the same point (double it) or opposite points (the identity)
    r = F::zero();
    if (quad_wave_any(r_zero)) {  // rare: entered by the wave, computed by every quad, kept where it applies
      const F dd = quad_dbl(c1, q);
      if (r_zero) r = dd;
    }
  }
  r = fsel(inf1, r, c2);
  r = fsel(inf2, r, c1);
  return r;
}

// this lane's coordinate of a point in memory / store it
template <class F>
ZG_HD F quad_load(const XYZZ<F>* p, unsigned q) {
  return reinterpret_cast<const F*>(p)[q];
}
template <class F>
ZG_HD void quad_store(XYZZ<F>* p, unsigned q, const F& c) {
  reinterpret_cast<F*>(p)[q] = c;
}

}  // namespace zg
