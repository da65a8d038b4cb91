"""GPU parity for the evaluation-domain transforms through the C ABI (zg_ntt) against the oracles:
basic_radix2_domain semantics (libzeth/core/evaluator_from_lagrange.tcc:43; witness-map usage SURVEY 3.1 5a).
Bit-exact: field arithmetic is exact."""
import random

import pytest

pytestmark = pytest.mark.gpu

import cpu_oracle  # noqa: E402
from zeth_model import BLS12_377, BN254, Domain  # noqa: E402

CURVES = {"alt-bn128": (0, BN254), "bls12-377": (1, BLS12_377)}


@pytest.fixture(scope="module")
def ctx():
    import zeth_amd

    c = zeth_amd.Context(0)
    yield c
    c.close()


def _variants(dom, C, a):
    out = {}
    for name in ("FFT", "iFFT", "cosetFFT", "icosetFFT"):
        v = list(a)
        if name in ("cosetFFT", "icosetFFT"):
            getattr(dom, name)(v, C.fr_gen)
        else:
            getattr(dom, name)(v)
        out[name] = v
    return out


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
@pytest.mark.parametrize("m", [1, 2, 4, 16, 64, 1024])
def test_ntt_matches_model(ctx, curve, m):
    import zeth_amd

    cid, C = CURVES[curve]
    rng = random.Random(m + cid)
    a = [rng.randrange(C.r) for _ in range(m)]
    a[0] = 0
    if m > 2:
        a[1] = C.r - 1
    want = _variants(Domain(C, m), C, a) if m > 1 else {k: list(a) for k in ("FFT", "iFFT", "cosetFFT", "icosetFFT")}
    dom = zeth_amd.EvaluationDomain(ctx, cid, m)
    for name, w in want.items():
        assert getattr(dom, name)(list(a)) == w, name


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
@pytest.mark.parametrize("log_m", [12, 13, 16, 19])
def test_ntt_matches_cpu_oracle_large(ctx, curve, log_m):
    """Two-pass sizes (N1 x N2 decomposition) against the C++ oracle; round trips and linearity on top."""
    import zeth_amd
    from zeth_amd.synth import random_scalars_host

    cid, C = CURVES[curve]
    m = 1 << log_m
    words = random_scalars_host(m, C.r, seed=log_m * 7 + cid)
    host = words.tobytes()
    dom = zeth_amd.EvaluationDomain(ctx, cid, m)
    for inverse, coset, name in ((False, False, "FFT"), (True, False, "iFFT"), (False, True, "cosetFFT"), (True, True, "icosetFFT")):
        ref = bytearray(host)
        cpu_oracle.fft_bytes(cid, ref, m, inverse, coset)
        buf = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
        getattr(dom, name)(buf)
        assert buf.download() == bytes(ref), name
        buf.free()
    # round trips on the device
    buf = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
    dom.FFT(buf)
    dom.iFFT(buf)
    assert buf.download() == host
    dom.cosetFFT(buf)
    dom.icosetFFT(buf)
    assert buf.download() == host
    buf.free()


def test_ntt_rejects_bad_sizes(ctx):
    import zeth_amd

    with pytest.raises(ValueError):
        zeth_amd.EvaluationDomain(ctx, 0, 3)
    dom = zeth_amd.EvaluationDomain(ctx, 0, 1 << 29)  # beyond alt_bn128's two-adicity 28
    buf = zeth_amd.DeviceBuffer(ctx, 64)
    with pytest.raises(zeth_amd.ZethGpuError):
        ctx._L.zg_ntt.restype  # noqa: B018
        zeth_amd.lib._check(ctx._L.zg_ntt(ctx._h, 0, buf.ptr, 29, 0, 0, 0, 1))
    buf.free()
    assert dom.m == 1 << 29
