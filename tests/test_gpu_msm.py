"""GPU parity for the MSM path through the C ABI (include/zeth_gpu.h): zg_msm vs the big-integer oracle
(libzeth::multi_exp semantics, libzeth/core/multi_exp.tcc:13-35) on the same seeded inputs.
Bit-exact: the affine result of a group sum is unique."""
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


def _points(G, rng, n, order):
    """n pseudo-random multiples of the generator built with few group ops."""
    pts = []
    P = G.mul(rng.randrange(1, order), G.gen)
    D = G.mul(rng.randrange(1, order), G.gen)
    for _ in range(n):
        pts.append(P)
        P = G.add(P, D)
        if rng.random() < 0.05:
            D = G.dbl(D)
    return pts


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
@pytest.mark.parametrize("group", [1, 2])
def test_msm_small_random(ctx, curve, group):
    import zeth_amd

    cid, C = CURVES[curve]
    G = C.G1 if group == 1 else C.G2
    rng = random.Random(1000 + cid * 2 + group)
    n = 300 if group == 1 else 120
    pts = _points(G, rng, n, C.r)
    bases = zeth_amd.Bases.from_points(ctx, cid, group, pts)
    assert bases.read(0, 3) == pts[:3]
    for trial, m in enumerate([n, 1, 2, 17]):
        sc = [rng.randrange(C.r) for _ in range(m)]
        assert bases.msm(sc) == G.multi_exp(pts, sc), (trial, m)
    # prefix use of the bases (multi_exp.tcc:28-34) and an offset window
    sc = [rng.randrange(C.r) for _ in range(10)]
    assert bases.msm(sc, first=5) == G.multi_exp(pts[5:], sc)
    bases.free()


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
def test_msm_edge_scalars(ctx, curve):
    """zero / one / r-1 / small / duplicate-point / cancelling inputs: the exceptional branches of the
    mixed addition (doubling, P + (-P), identity bases) and the 0/1-heavy witness shape of section 3.1."""
    import zeth_amd

    cid, C = CURVES[curve]
    G = C.G1
    rng = random.Random(77 + cid)
    pts = _points(G, rng, 64, C.r)
    # duplicates, negations and identity entries among the bases
    pts[10] = pts[3]
    pts[11] = G.neg(pts[3])
    pts[12] = None
    pts[13] = pts[3]
    bases = zeth_amd.Bases.from_points(ctx, cid, 1, pts)
    r = C.r
    cases = {
        "all_zero": [0] * 64,
        "all_one": [1] * 64,
        "all_minus_one": [r - 1] * 64,
        "bits": [rng.randrange(2) for _ in range(64)],
        "small": [rng.randrange(16) for _ in range(64)],
        "same_scalar_dups": [5 if i in (3, 10, 11, 13) else 0 for i in range(64)],
        "cancel": [7 if i in (3, 11) else 0 for i in range(64)],
        "mixed": [rng.choice([0, 1, r - 1, rng.randrange(r)]) for _ in range(64)],
    }
    for name, sc in cases.items():
        for c in (0, 7, 16):
            assert bases.msm(sc, window_bits=c) == G.multi_exp(pts, sc), (name, c)
    assert bases.msm([]) is None
    bases.free()


def test_msm_rejects_more_scalars_than_bases(ctx):
    import zeth_amd

    C = BN254
    bases = zeth_amd.Bases.from_points(ctx, 0, 1, [C.G1.gen, C.G1.dbl(C.G1.gen)])
    with pytest.raises(zeth_amd.ZethGpuError) as e:
        bases.msm([1, 2, 3])
    assert e.value.status == -1
    bases.free()


@pytest.mark.parametrize("curve,group,n", [("alt-bn128", 1, 1 << 16), ("alt-bn128", 2, 1 << 14), ("bls12-377", 1, 1 << 14), ("bls12-377", 2, 1 << 12)])
def test_msm_generated_bases_closed_form(ctx, curve, group, n):
    """Synthetic bases P_i = (t0 + i d) G: sum a_i P_i must equal (sum a_i (t0 + i d)) G."""
    import zeth_amd

    cid, C = CURVES[curve]
    G = C.G1 if group == 1 else C.G2
    rng = random.Random(n + group)
    t0, d = rng.randrange(C.r), rng.randrange(C.r)
    bases = zeth_amd.Bases.generate(ctx, cid, group, n, t0, d)
    for i in (0, 1, 63, 64, 65, n - 1):
        assert bases.read(i, 1)[0] == G.mul((t0 + i * d) % C.r, G.gen), i
    sc = [rng.randrange(C.r) for _ in range(n)]
    k = sum(a * (t0 + i * d) for i, a in enumerate(sc)) % C.r
    assert bases.msm(sc) == G.mul(k, G.gen)
    # linearity: msm(a) + msm(b) == msm(a + b)
    sb = [rng.randrange(C.r) for _ in range(n)]
    sab = [(a + b) % C.r for a, b in zip(sc, sb)]
    assert zeth_amd.host_group_op(cid, group, "add", bases.msm(sc), bases.msm(sb)) == bases.msm(sab)
    bases.free()


@pytest.mark.parametrize("window_bits", [0, 10, 13, 16])
def test_msm_skewed_scalars_two_level_sort(ctx, window_bits):
    """0/1-heavy and tiny-valued scalars (the JoinSplit witness shape) at a size that takes the two-level sort and
    the split-bucket tree: a few buckets hold almost every entry, most partitions are empty."""
    import zeth_amd

    C = BN254
    rng = random.Random(4242 + window_bits)
    n = 40000
    t0, d = rng.randrange(C.r), rng.randrange(C.r)
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, t0, d)
    sc = []
    for i in range(n):
        u = rng.random()
        sc.append(0 if u < 0.3 else 1 if u < 0.8 else C.r - 1 if u < 0.85 else rng.randrange(1 << 20) if u < 0.95 else rng.randrange(C.r))
    k = sum(a * (t0 + i * d) for i, a in enumerate(sc)) % C.r
    assert bases.msm(sc, window_bits=window_bits) == C.G1.mul(k, C.G1.gen)
    # all scalars equal: every entry of a window lands in one bucket
    same = [0x123456789ABCDEF] * n
    k = sum(a * (t0 + i * d) for i, a in enumerate(same)) % C.r
    assert bases.msm(same, window_bits=window_bits) == C.G1.mul(k, C.G1.gen)
    bases.free()


def test_msm_degenerate_scalar_sets_at_size(ctx):
    """SURVEY 8d edge-case sets at 2^20 points, checked by the closed form: all ones (one bucket holds every entry:
    thousands of units, both levels of the combine tree), all r - 1, all zero, one constant in every window, and a
    JoinSplit-like mix (90 % bits).  Scalars are uploaded as bytes; the closed form uses sums of arithmetic series."""
    import numpy as np

    import zeth_amd

    C = BN254
    r = C.r
    n = 1 << 20
    t0, d = 0x1F2E3D4C5B6A79, 0x1122334455667
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, t0, d)
    s1 = (n * t0 + d * (n * (n - 1) // 2)) % r  # sum_i (t0 + i d)

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
    # 90 % bits, 10 % small values: sum over the selected indices
    rng = np.random.default_rng(5)
    vals = np.where(rng.random(n) < 0.9, rng.integers(0, 2, n), rng.integers(2, 1 << 16, n)).astype(np.uint64)
    w = np.zeros((n, 4), dtype=np.uint64)
    w[:, 0] = vals
    idx = np.arange(n, dtype=object)
    total = (int(vals.astype(object).sum()) * t0 + d * int((vals.astype(object) * idx).sum())) % r
    assert run(w) == C.G1.mul(total, C.G1.gen)
    bases.free()
