import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import zeth_amd
from zeth_amd.synth import random_scalars_host, closed_form_scalar
from zeth_model import BN254, BLS12_377

curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
group = int(sys.argv[2]) if len(sys.argv) > 2 else 1
logs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 20, 22, 24]
C = BN254 if curve == 0 else BLS12_377
G = C.G1 if group == 1 else C.G2
ctx = zeth_amd.Context(0)
ctx.set_profiling(True)
t0, d = 0x1234567890ABCDEF1234567, 0xFEDCBA987654321
for lg in logs:
    n = 1 << lg
    t = time.time()
    bases = zeth_amd.Bases.generate(ctx, curve, group, n, t0, d)
    tg = time.time() - t
    sc = zeth_amd.DeviceBuffer.random_scalars(ctx, curve, n, lg)
    res = None
    for it in range(3):
        t = time.time()
        res = bases.msm(sc, n=n)
        dt = time.time() - t
    k = closed_form_scalar(random_scalars_host(n, C.r, lg), t0, d, C.r)
    ok = res == G.mul(k, G.gen)
    print("n=2^%d gen %.2fs msm %.2f ms ok=%s" % (lg, tg, dt * 1e3, ok), " ".join("%s=%.3f" % x for x in ctx.last_timings()), flush=True)
    bases.free(); sc.free()
