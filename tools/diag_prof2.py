import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.zeros(4, device="cuda")
import numpy as np
import zeth_amd
from zeth_amd.synth import random_scalars_host, scalars_to_ints, closed_form_scalar
from zeth_model import BN254 as C
ctx = zeth_amd.Context(0)
print([l.split()[-1].split("/")[-3:] for l in open("/proc/self/maps") if ("amdhip64" in l or "hsa-runtime" in l) and "r-xp" in l])
t0, d = 12345, 67891
for n in (1024, 1 << 16):
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, t0, d)
    sc = zeth_amd.DeviceBuffer.random_scalars(ctx, 0, n, 3)
    w = random_scalars_host(n, C.r, 3)
    exp = C.G1.mul(closed_form_scalar(w, t0, d, C.r), C.G1.gen)
    raw = sc.download()
    print(n, "gen equal", raw == w.tobytes(), "dev ok", bases.msm(sc) == exp, "host ok", bases.msm(raw) == exp, "dev again", bases.msm(sc) == exp, flush=True)
    sc2 = zeth_amd.DeviceBuffer.from_bytes(ctx, raw)
    r2 = bases.msm(sc2)
    r1 = bases.msm(sc)
    print("   uploaded-buffer ok", r2 == exp, "| dev result is None:", r1 is None, "| dev == uploaded:", r1 == r2, flush=True)
    half = bases.msm(sc, n=n // 2); halfh = bases.msm(raw[: 32 * (n // 2)])
    print("   half dev == half host", half == halfh, flush=True)
