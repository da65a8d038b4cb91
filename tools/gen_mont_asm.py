#!/usr/bin/env python3
"""Generate zeth_amd/csrc/mont_asm.h: gfx950 inline-assembly Montgomery multiplication (CIOS, split-phase) and
carry-chain add/sub for 8- and 12-limb fields.

Why assembly: hipcc turns the C multiplier into 128 v_mad_u64_u32 + 120 64-bit adds + 340 v_mov (zero-extension
of 32-bit words into register pairs).  Here each row is N independent v_mad_u64_u32 whose 64-bit addend is the
register pair {t_j, 0}, followed by ONE v_add_co/v_addc_co carry chain (2 instructions per 32x32 MAC).  The pairs
{t_j, 0} need adjacent registers with a constant-zero upper half, which inline-asm operands cannot express, so the
temporaries are fixed physical VGPRs named in the clobber list."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TB = 64  # first temporary VGPR


def gen_mul(N):
    # operand numbering: %0..%N-1 = r (out), %N..%2N-1 = a, %2N..%3N-1 = b, %3N..%4N-1 = p (sgpr), %4N = inv (sgpr)
    r = lambda j: "%%%d" % j
    a = lambda j: "%%%d" % (N + j)
    b = lambda j: "%%%d" % (2 * N + j)
    p = lambda j: "%%%d" % (3 * N + j)
    inv = "%%%d" % (4 * N)
    T = lambda j: TB + 2 * j  # t_j in v[T(j)], zero in v[T(j)+1], j = 0..N
    PB = TB + 2 * (N + 1)
    P = lambda j: PB + 2 * j  # pair j
    M = PB + 2 * N
    last = M
    L = []
    tp = lambda j: "v[%d:%d]" % (T(j), T(j) + 1)
    pp = lambda j: "v[%d:%d]" % (P(j), P(j) + 1)
    lo = lambda j: "v%d" % P(j)
    hi = lambda j: "v%d" % (P(j) + 1)
    t = lambda j: "v%d" % T(j)
    for j in range(N + 1):
        L.append("v_mov_b32 v%d, 0" % (T(j) + 1))
    for i in range(N):
        for j in range(N):
            if i == 0:
                L.append("v_mad_u64_u32 %s, vcc, %s, %s, 0" % (pp(j), a(j), b(i)))
            else:
                L.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (pp(j), a(j), b(i), tp(j)))
        L.append("v_mov_b32 %s, %s" % (t(0), lo(0)))
        L.append("v_add_co_u32 %s, vcc, %s, %s" % (t(1), lo(1), hi(0)))
        for j in range(2, N):
            L.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (t(j), lo(j), hi(j - 1)))
        L.append("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (t(N), hi(N - 1)))
        L.append("v_mul_lo_u32 v%d, %s, %s" % (M, t(0), inv))
        for j in range(N):
            L.append("v_mad_u64_u32 %s, vcc, v%d, %s, %s" % (pp(j), M, p(j), tp(j)))
        L.append("v_add_co_u32 %s, vcc, %s, %s" % (t(0), lo(1), hi(0)))
        for j in range(2, N):
            L.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (t(j - 1), lo(j), hi(j - 1)))
        L.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (t(N - 1), t(N), hi(N - 1)))
    # conditional subtraction: d = t - p ; r = borrow ? t : d   (p moved to VGPRs: a VOP2 carry op cannot also read an SGPR)
    for j in range(N):
        L.append("v_mov_b32 v%d, %s" % (P(0) + N + j, p(j)))
    L.append("v_sub_co_u32 %s, vcc, %s, v%d" % (lo(0), t(0), P(0) + N))
    for j in range(1, N):
        L.append("v_subb_co_u32 v%d, vcc, %s, v%d, vcc" % (P(0) + j, t(j), P(0) + N + j))
    # mask = borrow ? ~0 : 0 ; r = (t & mask) | (d & ~mask).  (v_cndmask_b32 on vcc measured ~20 cycles per wave
    # instruction on gfx950 against 4 for v_bfi_b32: profiles/r1_peak_valu.jsonl)
    L.append("v_subb_co_u32 v%d, vcc, 0, 0, vcc" % M)
    for j in range(N):
        L.append("v_bfi_b32 %s, v%d, %s, v%d" % (r(j), M, t(j), P(0) + j))
    clob = ", ".join('"v%d"' % k for k in range(TB, last + 1))
    outs = ", ".join('"=v"(r[%d])' % j for j in range(N))
    ins = ", ".join(['"v"(a[%d])' % j for j in range(N)] + ['"v"(b[%d])' % j for j in range(N)] +
                    ['"s"(p[%d])' % j for j in range(N)] + ['"s"(inv)'])
    body = "\n".join('      "%s\\n"' % s for s in L)
    return """
__device__ __forceinline__ void zg_mont_mul_%d(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b,
                                              const uint32_t* p, uint32_t inv) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(
%s
      : %s
      : %s
      : "vcc", %s);
#endif
}
""" % (N, body, outs, ins, clob), len(L)


def gen_addsub(N):
    # add: s = a + b; d = s - p; r = borrow ? s : d.   temporaries: fixed v[TB .. TB+2N)
    out = []
    r = lambda j: "%%%d" % j
    a = lambda j: "%%%d" % (N + j)
    b = lambda j: "%%%d" % (2 * N + j)
    p = lambda j: "%%%d" % (3 * N + j)
    S = lambda j: "v%d" % (TB + j)
    D = lambda j: "v%d" % (TB + N + j)
    L = ["v_add_co_u32 %s, vcc, %s, %s" % (S(0), a(0), b(0))]
    for j in range(1, N):
        L.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (S(j), a(j), b(j)))
    L.append("v_sub_co_u32 %s, vcc, %s, %s" % (D(0), S(0), p(0)))
    for j in range(1, N):
        L.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (D(j), S(j), p(j)))
    MK = "v%d" % (TB + 2 * N)
    L.append("v_subb_co_u32 %s, vcc, 0, 0, vcc" % MK)
    for j in range(N):
        L.append("v_bfi_b32 %s, %s, %s, %s" % (r(j), MK, S(j), D(j)))
    clob = ", ".join('"v%d"' % k for k in range(TB, TB + 2 * N + 1))
    outs = ", ".join('"=v"(r[%d])' % j for j in range(N))
    ins = ", ".join(['"v"(a[%d])' % j for j in range(N)] + ['"v"(b[%d])' % j for j in range(N)] + ['"v"(p[%d])' % j for j in range(N)])
    body = "\n".join('      "%s\\n"' % s for s in L)
    out.append("""
__device__ __forceinline__ void zg_mont_add_%d(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b, const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(
%s
      : %s
      : %s
      : "vcc", %s);
#endif
}
""" % (N, body, outs, ins, clob))
    # sub: d = a - b; borrow -> mask; r = d + (p & mask)
    L = ["v_sub_co_u32 %s, vcc, %s, %s" % (D(0), a(0), b(0))]
    for j in range(1, N):
        L.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (D(j), a(j), b(j)))
    # S(j) = borrow ? p_j : 0
    L.append("v_subb_co_u32 %s, vcc, 0, 0, vcc" % MK)
    for j in range(N):
        L.append("v_and_b32 %s, %s, %s" % (S(j), MK, p(j)))
    L.append("v_add_co_u32 %s, vcc, %s, %s" % (r(0), D(0), S(0)))
    for j in range(1, N):
        L.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (r(j), D(j), S(j)))
    outs_ec = ", ".join('"=&v"(r[%d])' % j for j in range(N))
    body = "\n".join('      "%s\\n"' % s for s in L)
    out.append("""
__device__ __forceinline__ void zg_mont_sub_%d(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b, const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(
%s
      : %s
      : %s
      : "vcc", %s);
#endif
}
""" % (N, body, outs_ec, ins, clob))
    # cneg: r = (flagmask && a != 0) ? p - a : a      (flagmask: all-ones or zero)
    L = ["v_sub_co_u32 %s, vcc, %s, %s" % (D(0), p(0), a(0))]
    for j in range(1, N):
        L.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (D(j), p(j), a(j)))
    L.append("v_or_b32 %s, %s, %s" % (S(0), a(0), a(1)))
    for j in range(2, N):
        L.append("v_or_b32 %s, %s, %s" % (S(0), S(0), a(j)))
    L.append("v_cmp_ne_u32 vcc, 0, %s" % S(0))
    L.append("v_subb_co_u32 %s, vcc, 0, 0, vcc" % MK)  # all-ones when a != 0
    L.append("v_and_b32 %s, %s, %%%d" % (MK, MK, 4 * N))
    for j in range(N):
        L.append("v_bfi_b32 %s, %s, %s, %s" % (r(j), MK, D(j), a(j)))
    outs_ec = ", ".join('"=&v"(r[%d])' % j for j in range(N))
    ins_c = ", ".join(['"v"(a[%d])' % j for j in range(N)] + ['"v"(a[%d])' % j for j in range(N)] +
                      ['"v"(p[%d])' % j for j in range(N)] + ['"v"(flagmask)'])
    body = "\n".join('      "%s\\n"' % s for s in L)
    out.append("""
__device__ __forceinline__ void zg_mont_cneg_%d(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* p, uint32_t flagmask) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(
%s
      : %s
      : %s
      : "vcc", %s);
#endif
}
""" % (N, body, outs_ec, ins_c, clob))
    return "".join(out)


def main():
    parts = ["// GENERATED by tools/gen_mont_asm.py -- do not edit.\n#pragma once\n#include <stdint.h>\n"]
    for N in (8, 12):
        src, n = gen_mul(N)
        parts.append("// %d-limb Montgomery multiplication: %d instructions\n" % (N, n))
        parts.append(src)
        parts.append(gen_addsub(N))
        path = os.path.join(ROOT, "zeth_amd", "csrc", "mont_asm.h")
    open(path, "w").write("".join(parts))
    print("wrote", path)


if __name__ == "__main__":
    main()
