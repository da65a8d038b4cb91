import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zeth_amd
ctx = zeth_amd.Context(0); ctx.set_profiling(True)
for lg in [int(x) for x in sys.argv[1:]] or (14, 16, 17, 18, 19, 20, 21):
    n = 1 << lg
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, 12345, 6789)
    sc = zeth_amd.DeviceBuffer.random_scalars(ctx, 0, n, lg)
    row = []
    for c in (0, 10, 11, 12, 13, 14, 15, 16):
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); bases.msm(sc, n=n, window_bits=c); best = min(best, time.perf_counter() - t)
        ph = dict(ctx.last_timings())
        row.append("c=%d: %.2f (acc %.2f red %.2f)" % (c, best * 1e3, ph["accum"], ph["reduce1"] + ph["reduce2"] + ph["combine"]))
    print("2^%d-1:" % lg, " | ".join(row), flush=True)
    bases.free(); sc.free()
