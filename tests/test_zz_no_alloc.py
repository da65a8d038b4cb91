"""zg_ntt_noalloc (the entry point that allocates nothing: static device scratch, log n <= 16) against the existing
entry point zg_ntt on the same device-resident input, bit for bit, through the C ABI.  Two calls in a row, the second
with a smaller input; then the cached-table path, the other curve, and the size limit.  One small case is also held
against the CPU oracle.  No counterpart in the reference (it exists for callers that cannot call hipMalloc)."""
import pytest

pytestmark = pytest.mark.gpu

import cpu_oracle  # noqa: E402
from zeth_model import BLS12_377, BN254  # noqa: E402

CURVES = {"alt-bn128": (0, BN254), "bls12-377": (1, BLS12_377)}
VARIANTS = ((False, False), (True, False), (False, True), (True, True))


def _both(ctx, cid, C, log_m, seed):
    """every variant through both entry points on the same input; returns the number of comparisons made"""
    import zeth_amd
    from zeth_amd.synth import random_scalars_host

    m = 1 << log_m
    host = random_scalars_host(m, C.r, seed=seed).tobytes()
    dom = zeth_amd.EvaluationDomain(ctx, cid, m)
    done = 0
    for inverse, coset in VARIANTS:
        want = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
        dom._run(want, inverse, coset)
        got = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
        dom.run_noalloc(got, inverse, coset)
        assert got.download() == want.download(), (log_m, inverse, coset)
        want.free()
        got.free()
        done += 1
    return done


def test_noalloc_ntt_equals_zg_ntt_two_calls_second_smaller():
    import zeth_amd

    ctx = zeth_amd.Context(0)
    try:
        cid, C = CURVES["alt-bn128"]
        assert _both(ctx, cid, C, 16, seed=11) == 4  # the maximum size
        assert _both(ctx, cid, C, 10, seed=12) == 4  # then a smaller one: the scratch still holds the 2^16 tables
        assert _both(ctx, cid, C, 10, seed=13) == 4  # same domain again: cached tables
        cid2, C2 = CURVES["bls12-377"]
        assert _both(ctx, cid2, C2, 13, seed=14) == 4  # other curve, two-pass size
        assert _both(ctx, cid, C, 0, seed=15) == 4  # m = 1
        assert _both(ctx, cid, C, 16, seed=16) == 4  # and back up
    finally:
        ctx.close()


def test_noalloc_ntt_matches_cpu_oracle_and_rejects_large():
    import zeth_amd
    from zeth_amd.synth import random_scalars_host

    ctx = zeth_amd.Context(0)
    try:
        cid, C = CURVES["bls12-377"]
        m = 1 << 12
        host = random_scalars_host(m, C.r, seed=21).tobytes()
        dom = zeth_amd.EvaluationDomain(ctx, cid, m)
        for inverse, coset in VARIANTS:
            ref = bytearray(host)
            cpu_oracle.fft_bytes(cid, ref, m, inverse, coset)
            buf = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
            dom.run_noalloc(buf, inverse, coset)
            assert buf.download() == bytes(ref)
            buf.free()
        big = zeth_amd.DeviceBuffer(ctx, 32 << 17)
        with pytest.raises(zeth_amd.ZethGpuError) as e:
            zeth_amd.EvaluationDomain(ctx, cid, 1 << 17).run_noalloc(big, False, False)
        assert e.value.status == -6  # ZG_ERR_UNSUPPORTED: beyond the static scratch
        big.free()
    finally:
        ctx.close()


def test_noalloc_ntt_from_two_contexts_at_once():
    """Two contexts on one GPU (the serving-loop case) call the entry point from two threads with different domains: the
    calls share the GPU's static scratch, so the library makes them take turns -- every result equals zg_ntt's."""
    import threading

    import zeth_amd
    from zeth_amd.synth import random_scalars_host

    cid, C = CURVES["alt-bn128"]
    errors = []

    def worker(k):
        ctx = zeth_amd.Context(0)
        try:
            for it in range(6):
                log_m = (9, 12)[(k + it) % 2] if it % 3 else (14, 11)[k]
                m = 1 << log_m
                host = random_scalars_host(m, C.r, seed=100 * k + it).tobytes()
                dom = zeth_amd.EvaluationDomain(ctx, cid, m)
                inverse, coset = VARIANTS[(k + it) % 4]
                want = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
                dom._run(want, inverse, coset)
                got = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
                dom.run_noalloc(got, inverse, coset)
                if got.download() != want.download():
                    errors.append((k, it, log_m))
                want.free()
                got.free()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
        finally:
            ctx.close()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
