"""GPU parity for the fixed-base table mode of the MSM (zg_bases_precompute, include/zeth_gpu.h): the same zg_msm calls
as tests/test_gpu_msm.py on bases that carry a table must return the same group element -- the oracle's
libzeth::multi_exp (libzeth/core/multi_exp.tcc:13-35) on small inputs, the closed form on generated bases.  Bit-exact."""
import random

import pytest

pytestmark = pytest.mark.gpu

from zeth_model import BLS12_377, BN254  # noqa: E402

CURVES = {"alt-bn128": (0, BN254), "bls12-377": (1, BLS12_377)}


@pytest.fixture(scope="module")
def ctx():
    import zeth_amd

    c = zeth_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("curve,group,n,window_bits", [
    ("alt-bn128", 1, 1 << 16, 0), ("alt-bn128", 1, 5000, 12), ("alt-bn128", 1, 1 << 15, 17), ("alt-bn128", 1, 1 << 14, 22),
    ("alt-bn128", 2, 1 << 13, 0), ("bls12-377", 1, 1 << 13, 16), ("bls12-377", 2, 1 << 12, 13)])
def test_table_msm_closed_form(ctx, curve, group, n, window_bits):
    """P_i = (t0 + i d) G: table mode against the closed form, against the plain mode (explicit window width), on a
    sub-range, and the table levels themselves (level 0 = the bases, read back unchanged)."""
    import zeth_amd

    cid, C = CURVES[curve]
    G = C.G1 if group == 1 else C.G2
    rng = random.Random(n * 31 + group + window_bits)
    t0, d = rng.randrange(C.r), rng.randrange(C.r)
    bases = zeth_amd.Bases.generate(ctx, cid, group, n, t0, d)
    assert bases.table_info() == (0, 0)
    bases.precompute(window_bits)
    c, levels = bases.table_info()
    assert (window_bits == 0 or c == window_bits) and levels == (C.r.bit_length() + 2 + c - 1) // c
    bases.precompute(0)  # no-op
    with pytest.raises(zeth_amd.ZethGpuError):
        bases.precompute(c + 1 if c < 22 else c - 1)
    for i in (0, 1, n - 1):
        assert bases.read(i, 1)[0] == G.mul((t0 + i * d) % C.r, G.gen), i
    sc = [rng.randrange(C.r) for _ in range(n)]
    k = sum(a * (t0 + i * d) for i, a in enumerate(sc)) % C.r
    want = G.mul(k, G.gen)
    assert bases.msm(sc) == want
    assert bases.msm(sc, window_bits=13) == want  # plain mode on level 0 of the table
    # a range inside the bases (multi_exp.tcc:28-34): n/2 scalars from base n/4
    m, f = n // 2, n // 4
    k = sum(a * (t0 + (f + i) * d) for i, a in enumerate(sc[:m])) % C.r
    assert bases.msm(sc[:m], first=f) == G.mul(k, G.gen)
    # short calls fall back to the plain form
    k = sum(a * (t0 + (3 + i) * d) for i, a in enumerate(sc[:100])) % C.r
    assert bases.msm(sc[:100], first=3) == G.mul(k, G.gen)
    assert bases.msm([]) is None
    bases.free()


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
def test_table_msm_arbitrary_points_and_edge_scalars(ctx, curve):
    """Uploaded (not generated) bases with duplicates, negations and identity entries; zero / one / r-1 / bit / small
    scalars: against the oracle's multi_exp."""
    import zeth_amd

    cid, C = CURVES[curve]
    G = C.G1
    rng = random.Random(99 + cid)
    n = 2048
    pts = []
    P = G.mul(rng.randrange(1, C.r), G.gen)
    D = G.mul(rng.randrange(1, C.r), G.gen)
    for _ in range(n):
        pts.append(P)
        P = G.add(P, D)
    pts[10] = pts[3]
    pts[11] = G.neg(pts[3])
    pts[12] = None
    pts[2047] = None
    bases = zeth_amd.Bases.from_points(ctx, cid, 1, pts).precompute(12)
    r = C.r
    cases = {
        "random": [rng.randrange(r) for _ in range(n)],
        "all_zero": [0] * n,
        "all_one": [1] * n,
        "all_minus_one": [r - 1] * n,
        "bits": [rng.randrange(2) for _ in range(n)],
        "small": [rng.randrange(16) for _ in range(n)],
        "cancel": [5 if i in (3, 11) else 0 for i in range(n)],
        "double": [7 if i in (3, 10, 13) else 0 for i in range(n)],
        "top": [r - 1 - rng.randrange(4) for _ in range(n)],
    }
    for name, sc in cases.items():
        assert bases.msm(sc) == G.multi_exp(pts, sc), name
    bases.free()


@pytest.mark.parametrize("window_bits", [0, 16, 22])
def test_table_msm_skewed_scalars(ctx, window_bits):
    """0/1-heavy and tiny-valued scalars (JoinSplit witness shape): one bucket group holds almost every entry, most
    groups are empty; all scalars equal: every digit position lands in one bucket."""
    import zeth_amd

    C = BN254
    rng = random.Random(4242 + window_bits)
    n = 40000
    t0, d = rng.randrange(C.r), rng.randrange(C.r)
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, t0, d).precompute(window_bits)
    sc = []
    for i in range(n):
        u = rng.random()
        sc.append(0 if u < 0.3 else 1 if u < 0.8 else C.r - 1 if u < 0.85 else rng.randrange(1 << 20) if u < 0.95 else rng.randrange(C.r))
    k = sum(a * (t0 + i * d) for i, a in enumerate(sc)) % C.r
    assert bases.msm(sc) == C.G1.mul(k, C.G1.gen)
    same = [0x123456789ABCDEF] * n
    k = sum(a * (t0 + i * d) for i, a in enumerate(same)) % C.r
    assert bases.msm(same) == C.G1.mul(k, C.G1.gen)
    bases.free()


def test_table_msm_degenerate_scalar_sets_at_size(ctx):
    """SURVEY 8d edge-case sets at 2^20 points in table mode (c = 19), checked by the closed form."""
    import numpy as np

    import zeth_amd

    C = BN254
    r = C.r
    n = 1 << 20
    t0, d = 0x1F2E3D4C5B6A79, 0x1122334455667
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, t0, d).precompute()
    s1 = (n * t0 + d * (n * (n - 1) // 2)) % r

    def run(words):
        return bases.msm(words.astype("<u8").tobytes())

    def const(v):
        w = np.zeros((n, 4), dtype=np.uint64)
        for k in range(4):
            w[:, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        return w

    assert run(const(1)) == C.G1.mul(s1, C.G1.gen)
    assert run(const(r - 1)) == C.G1.mul((r - s1) % r, C.G1.gen)
    assert run(const(0)) is None
    k = 0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % r
    assert run(const(k)) == C.G1.mul(k * s1 % r, C.G1.gen)
    rng = np.random.default_rng(5)
    vals = np.where(rng.random(n) < 0.9, rng.integers(0, 2, n), rng.integers(2, 1 << 16, n)).astype(np.uint64)
    w = np.zeros((n, 4), dtype=np.uint64)
    w[:, 0] = vals
    idx = np.arange(n, dtype=object)
    total = (int(vals.astype(object).sum()) * t0 + d * int((vals.astype(object) * idx).sum())) % r
    assert run(w) == C.G1.mul(total, C.G1.gen)
    bases.free()


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
def test_table_msm_scalar_forms_and_memories(ctx, curve):
    """The same scalars as plain integers in host memory, as Montgomery residues (libff's in-memory form, what a Zeth
    caller hands over), and resident in device memory (the benchmark's path): one group element."""
    import zeth_amd
    from zeth_amd.lib import FORM_MONT
    from zeth_amd.synth import closed_form_scalar, random_scalars_host

    cid, C = CURVES[curve]
    n = 1 << 15
    t0, d = 0x1234567890ABCDEF1234567, 0xFEDCBA987654321
    bases = zeth_amd.Bases.generate(ctx, cid, 1, n, t0, d).precompute(20)
    dev = zeth_amd.DeviceBuffer.random_scalars(ctx, cid, n, 77)
    words = random_scalars_host(n, C.r, 77)
    want = C.G1.mul(closed_form_scalar(words, t0, d, C.r), C.G1.gen)
    assert bases.msm(dev, n=n) == want
    ints = [int(w[0]) | int(w[1]) << 64 | int(w[2]) << 128 | int(w[3]) << 192 for w in words]
    assert bases.msm(ints) == want
    mont = [(s << 256) % C.r for s in ints]
    assert bases.msm(mont, form=FORM_MONT) == want
    # a sub-range from device memory: scalars [0, n/2) against bases [n/4, 3n/4)
    k = sum(a * (t0 + (n // 4 + i) * d) for i, a in enumerate(ints[: n // 2])) % C.r
    assert bases.msm(dev, first=n // 4, n=n // 2) == C.G1.mul(k, C.G1.gen)
    dev.free()
    bases.free()
