#!/usr/bin/env python3
"""MSM size sweep (BASELINE.json configs[1]: BN254 G1 2^16..2^26; plus G2 and BLS12-377 points) and NTT timings.
Every MSM result is checked against the closed form (sum a_i (t0 + i d)) G.  One JSON line per measurement."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zeth_amd  # noqa: E402
from zeth_amd.synth import closed_form_scalar, random_scalars_host  # noqa: E402
from zeth_model import BLS12_377, BN254  # noqa: E402

MAC = {(0, 1): 21760, (0, 2): 65280, (1, 1): 48000, (1, 2): 144000}  # SURVEY 8d, per point
PEAK = 3.85e13


def msm_case(ctx, cid, gid, lg, reps=3, check=True, table=None):
    C = BN254 if cid == 0 else BLS12_377
    G = C.G1 if gid == 1 else C.G2
    n = 1 << lg
    t0, d = 0x1234567890ABCDEF1234567, 0xFEDCBA987654321
    bases = zeth_amd.Bases.generate(ctx, cid, gid, n, t0, d)
    sc = zeth_amd.DeviceBuffer.random_scalars(ctx, cid, n, lg)
    build_s = None
    if table is not None:
        t = time.perf_counter()
        bases.precompute(table)
        build_s = time.perf_counter() - t
    times = []
    res = None
    for _ in range(reps + 1):
        t = time.perf_counter()
        res = bases.msm(sc, n=n)
        times.append(time.perf_counter() - t)
    times = sorted(times[1:])
    ph = dict(ctx.last_timings())
    ok = None
    if check:
        ok = res == G.mul(closed_form_scalar(random_scalars_host(n, C.r, lg), t0, d, C.r), G.gen)
    tinfo = bases.table_info() if table is not None else None
    bases.free()
    sc.free()
    ms = times[len(times) // 2] * 1e3
    out = {"op": "msm", "curve": C.name, "group": "G%d" % gid, "log_n": lg, "ms": ms, "points_per_s": n / (ms * 1e-3),
           "accum_ms": ph.get("accum"), "mac_frac_whole_call": n * MAC[(cid, gid)] / (ms * 1e-3) / PEAK,
           "mac_frac_accum": n * MAC[(cid, gid)] / (ph["accum"] * 1e-3) / PEAK, "verified": ok, "phases_ms": ph}
    if table is not None:
        c, D = tinfo
        out.update({"table": {"window_bits": c, "levels": D, "build_s": build_s}})
        out["mac_frac_accum"] = None  # the per-point MAC count of SURVEY 8d assumes 16-17 digits per scalar
    print(json.dumps(out), flush=True)


def ntt_case(ctx, cid, lg, reps=5):
    C = BN254 if cid == 0 else BLS12_377
    m = 1 << lg
    buf = zeth_amd.DeviceBuffer.random_scalars(ctx, cid, m, 99)
    dom = zeth_amd.EvaluationDomain(ctx, cid, m)
    L = ctx._L
    import ctypes

    def run(inv, coset):
        zeth_amd.lib._check(L.zg_ntt(ctx._h, cid, ctypes.c_void_p(buf.ptr), lg, inv, coset, zeth_amd.lib.FORM_MONT, 1))

    run(0, 0)
    for inv, coset, name in ((0, 0, "FFT"), (1, 0, "iFFT"), (0, 1, "cosetFFT"), (1, 1, "icosetFFT")):
        ks = []
        for _ in range(reps):
            run(inv, coset)
            ph = dict(ctx.last_timings())
            ks.append((ph["ntt_pass_a"] + ph["ntt_pass_b"], ph["ntt_pass_a"], ph["ntt_pass_b"]))
        ks.sort()
        k_ms, a_ms, b_ms = ks[len(ks) // 2]
        out = {"op": "ntt", "variant": name, "curve": C.name, "log_m": lg, "kernels_ms": k_ms, "pass_a_ms": a_ms, "pass_b_ms": b_ms,
               "hbm_algorithmic_GBps": 128.0 * m / (k_ms * 1e-3) / 1e9, "hbm_frac": 128.0 * m / (k_ms * 1e-3) / 8e12,
               "mac_frac": (m / 2) * lg * 136 / (k_ms * 1e-3) / PEAK,
               "note": "two passes, each reads and writes the vector once: 128*m bytes; (m/2) log2(m) butterflies x 136 MAC"}
        print(json.dumps(out), flush=True)
    dom = None
    buf.free()


def main():
    ctx = zeth_amd.Context(0)
    ctx.set_profiling(True)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "g1"):
        for lg in range(16, 27):
            msm_case(ctx, 0, 1, lg)
    if which == "g1t":  # fixed-base table mode; optional digit width and sizes
        cw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
        sizes = [int(a) for a in sys.argv[3:]] or list(range(16, 27))
        for lg in sizes:
            msm_case(ctx, 0, 1, lg, table=cw)
    if which == "othert":
        msm_case(ctx, 0, 2, 22, table=0)
        msm_case(ctx, 1, 1, 22, table=0)
        msm_case(ctx, 1, 2, 20, table=0)
    if which in ("all", "other"):
        msm_case(ctx, 0, 2, 22)
        msm_case(ctx, 1, 1, 22)
        msm_case(ctx, 1, 2, 20)
    if which == "nttsizes":  # nttsizes LOG_M...
        for a in sys.argv[2:]:
            ntt_case(ctx, 0, int(a))
    if which == "ntt1":  # one size, for profiling: ntt1 LOG_M [REPS]
        ntt_case(ctx, 0, int(sys.argv[2]), reps=int(sys.argv[3]) if len(sys.argv) > 3 else 5)
    if which in ("all", "ntt"):
        for lg in (16, 19, 22, 24):
            ntt_case(ctx, 0, lg)
        ntt_case(ctx, 1, 22)


if __name__ == "__main__":
    main()
