"""Sweep the work-unit length L of the bucket accumulation (ZG_MSM_UNIT_LEN) over MSM sizes; one process per L because
the knob is read per plan.  python tools/tune_unit_len.py LOGN..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, time
sys.path.insert(0, %r)
import zeth_amd
ctx = zeth_amd.Context(0); ctx.set_profiling(True)
for lg in %r:
    n = 1 << lg
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, 12345, 6789)
    sc = zeth_amd.DeviceBuffer.random_scalars(ctx, 0, n, lg)
    best = 1e9
    for _ in range(4):
        t = time.perf_counter(); bases.msm(sc, n=n); best = min(best, time.perf_counter() - t)
    ph = dict(ctx.last_timings())
    print("  2^%%d: %%.2f ms (accum %%.2f combine %%.2f units %%.2f scan %%.2f)" %% (lg, best * 1e3, ph["accum"], ph["combine"], ph["units"], ph["scan"]), flush=True)
    bases.free(); sc.free()
'''
lgs = [int(x) for x in sys.argv[1:]] or [20, 22, 24]
for L in (0, 32, 64, 128, 256, 512):
    env = dict(os.environ)
    if L:
        env["ZG_MSM_UNIT_LEN"] = str(L)
    print("L =", L or "default", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE % (ROOT, lgs)], env=env)
    if r.returncode:
        sys.exit(r.returncode)
