"""Host-side helpers: the numpy mirror of the device scalar generator, the closed-form checker, point-range sharding
and the multi-rank exchange (world_size 2, gloo, CPU)."""
import os
import subprocess
import sys

import numpy as np

from zeth_amd.distributed import shard_range
from zeth_amd.synth import closed_form_scalar, random_scalars_host, scalars_to_ints
from zeth_model import BLS12_377, BN254

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_scalars_are_reduced_and_reproducible():
    for C in (BN254, BLS12_377):
        w = random_scalars_host(5000, C.r, seed=11)
        ints = scalars_to_ints(w)
        assert all(0 <= x < C.r for x in ints)
        assert len(set(ints)) == 5000
        assert np.array_equal(w, random_scalars_host(5000, C.r, seed=11))
        assert np.array_equal(w[100:200], random_scalars_host(100, C.r, seed=111))  # counter-based: seed + i
        assert max(ints).bit_length() == C.r.bit_length()


def test_closed_form_matches_big_int_sum():
    C = BN254
    w = random_scalars_host(3000, C.r, seed=4)
    ints = scalars_to_ints(w)
    t0, d = 0x1234567890ABCDEF, C.r - 5
    assert closed_form_scalar(w, t0, d, C.r) == sum(a * (t0 + i * d) for i, a in enumerate(ints)) % C.r


def test_shard_ranges_partition():
    for n in (0, 1, 7, 16, 1 << 24):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1


def test_two_rank_exchange_gloo(tmp_path):
    """Rank-local partial MSMs (CPU oracle standing in for the device, which is absent here) all-gathered over
    gloo and summed by the product's host tail equal the single-rank result."""
    code = r'''
import os, sys, random
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
import torch.distributed as dist
import cpu_oracle
from zeth_model import BN254 as C
from zeth_amd.distributed import all_gather_sum, all_gather_sum_bytes, shard_range
from zeth_amd.lib import decode_point, encode_point
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = random.Random(99)
n = 37
G = C.G1
pts = []
P = G.mul(rng.randrange(1, C.r), G.gen); D = G.mul(rng.randrange(1, C.r), G.gen)
for _ in range(n):
    pts.append(P); P = G.add(P, D)
sc = [rng.randrange(C.r) for _ in range(n)]
first, count = shard_range(n, world, rank)
part = cpu_oracle.msm(0, 1, pts[first:first + count], sc[first:first + count])
total = all_gather_sum(0, 1, part, "cpu")
assert total == cpu_oracle.msm(0, 1, pts, sc), rank
# the raw-bytes exchange bench.py uses per step (one zg_host_group_sum call for the combine)
raw = all_gather_sum_bytes(0, 1, encode_point(0, 1, part), "cpu")
assert decode_point(0, 1, raw) == total, rank
assert decode_point(0, 1, all_gather_sum_bytes(0, 1, encode_point(0, 1, pts[0] if rank == 0 else G.neg(pts[0])), "cpu")) is None
# cancelling partials: rank 0 holds P, rank 1 holds -P
total = all_gather_sum(0, 1, pts[0] if rank == 0 else G.neg(pts[0]), "cpu")
assert total is None
dist.destroy_process_group()
print("rank", rank, "ok")
''' % (ROOT, ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    script = tmp_path / "two_rank.py"
    script.write_text(code)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2
