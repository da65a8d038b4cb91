"""GPU parity at the shapes of BASELINE.json's configs that the small-case tests do not reach, through the C ABI:

  configs[4]  several proofs in flight: K threads, one zg_pk_clone each, proving CONCURRENTLY on one resident key
              (prover_server shares one key between its handlers, prover_server/prover_server.cpp:51,192-201,451)
  configs[3]  the 2^22-constraint BLS12-377 proof's building blocks: evaluation-domain transforms of 2^20 and 2^22
              points (four-step split N1 = N2 = 2^11) and a proof whose H query takes the fixed-base table at its
              DEFAULT threshold
  configs[2]  G2 multi-exponentiation at 2^18 points (closed form)

Oracles: oracle/zeth_cpu.cpp (C++ restatement, tests/test_oracle_cpu.py ties it to the model) and the closed form
(sum a_i t_i) G.  Bit-exact.  Matches the reference's own checks in shape: libzeth/tests/prover/prover_test.cpp:148-185
(prove, then verify), libzeth/tests/core/evaluator_from_lagrange_test.cpp:21-66 (iFFT + multi_exp)."""
import ctypes
import os
import sys
import threading

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import cpu_oracle  # noqa: E402
from zeth_model import BLS12_377, BN254  # noqa: E402

CURVES = {"alt-bn128": (0, BN254), "bls12-377": (1, BLS12_377)}


@pytest.fixture(scope="module")
def ctx():
    import zeth_amd

    c = zeth_amd.Context(0)
    yield c
    c.close()


def _key_and_oracle_proofs(ctx, cid, constraints, randomness):
    """JoinSplit-shaped circuit (zeth_amd/synth_circuit.py), real key from fixed secrets built on the device, and the
    CPU oracle's proof bytes for every (r, s) in `randomness`."""
    import bench_prove
    import zeth_amd.lib as zl
    from zeth_amd.keygen import CURVE_INFO, generate_setup_from_secrets
    from zeth_amd.synth_circuit import joinsplit_shaped

    info = CURVE_INFO[cid]
    cs, primary, aux = joinsplit_shaped(info.r, seed=3, constraints=constraints)
    dpk, vk, extra = generate_setup_from_secrets(ctx, cid, cs, t=0x7A657468, alpha=0xA1, beta=0xB2, delta=0xD3, return_scalars=True)
    threads = min(16, cpu_oracle.max_threads())
    want = []
    for r_rand, s_rand in randomness:
        raw, _ = bench_prove.cpu_prove(cid, cs, extra, primary, aux, zl.int_to_le(r_rand, 32), zl.int_to_le(s_rand, 32), threads)
        want.append(raw)
    return cs, primary, aux, dpk, want


def test_concurrent_clones_prove_bit_exact(ctx):
    """configs[4]: four threads, one clone of the resident key each, proofs overlapping on the device; every proof
    byte-equal to the CPU oracle's for its (r, s)."""
    import zeth_amd.lib as zl

    cid = 0
    randomness = [(0x1111 + 7 * k, 0x2222 + 13 * k) for k in range(3)]
    cs, primary, aux, dpk, want = _key_and_oracle_proofs(ctx, cid, 15000, randomness)
    assert cs.domain_size() == 1 << 14
    n = zl.fq_bytes(cid)
    pb = ctypes.create_string_buffer(zl.encode_scalars(primary), 32 * len(primary))
    ab = ctypes.create_string_buffer(zl.encode_scalars(aux), 32 * len(aux))
    K, rounds = 4, 6
    clones = [dpk] + [dpk.clone() for _ in range(K - 1)]
    errs = []
    start = threading.Barrier(K)

    def worker(k):
        try:
            out = ctypes.create_string_buffer(8 * n)
            start.wait()
            for it in range(rounds):
                which = (k + it) % len(randomness)
                r_rand, s_rand = randomness[which]
                zl._check(ctx._L.zg_prove(clones[k]._h, pb, len(primary), ab, len(aux), zl.FORM_PLAIN, zl.int_to_le(r_rand, 32),
                                          zl.int_to_le(s_rand, 32), zl.FORM_PLAIN, out))
                if out.raw != want[which]:
                    errs.append("thread %d round %d: proof differs from the oracle's" % (k, it))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
            start.abort()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for c in clones[1:]:
        c.free()
    dpk.free()
    assert not errs, errs


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
@pytest.mark.parametrize("log_m", [20, 22])
def test_ntt_large_matches_cpu_oracle(ctx, curve, log_m):
    """configs[3] transform sizes (2^22 = the domain of the 2^22-constraint proof) against the C++ oracle, all four
    variants, plus the round trips."""
    import zeth_amd
    from zeth_amd.synth import random_scalars_host

    cid, C = CURVES[curve]
    m = 1 << log_m
    host = random_scalars_host(m, C.r, seed=log_m * 11 + cid).tobytes()
    dom = zeth_amd.EvaluationDomain(ctx, cid, m)
    for inverse, coset, name in ((False, False, "FFT"), (True, False, "iFFT"), (False, True, "cosetFFT"), (True, True, "icosetFFT")):
        ref = bytearray(host)
        cpu_oracle.fft_bytes(cid, ref, m, inverse, coset)
        buf = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
        getattr(dom, name)(buf)
        assert buf.download() == bytes(ref), name
        buf.free()
    buf = zeth_amd.DeviceBuffer.from_bytes(ctx, host)
    dom.cosetFFT(buf)
    dom.icosetFFT(buf)
    dom.FFT(buf)
    dom.iFFT(buf)
    assert buf.download() == host
    buf.free()


def test_bls377_proof_with_default_h_table(ctx, monkeypatch):
    """configs[3] in the small: a BLS12-377 proof over a 2^18-point domain, where zg_pk_create keeps the H query as a
    fixed-base table by its DEFAULT rule (no ZG_PK_TABLE_MIN), against the C++ oracle."""
    import zeth_amd.lib as zl

    monkeypatch.delenv("ZG_PK_TABLE_MIN", raising=False)
    cid = 1
    randomness = [(0xABCDEF0123456789, 0x9876543210FEDCBA)]
    cs, primary, aux, dpk, want = _key_and_oracle_proofs(ctx, cid, 140000, randomness)
    assert cs.domain_size() == 1 << 18
    n = zl.fq_bytes(cid)
    out = ctypes.create_string_buffer(8 * n)
    pb = ctypes.create_string_buffer(zl.encode_scalars(primary), 32 * len(primary))
    ab = ctypes.create_string_buffer(zl.encode_scalars(aux), 32 * len(aux))
    for _ in range(2):  # the second proof sizes its windows from the first one's scalar statistics
        zl._check(ctx._L.zg_prove(dpk._h, pb, len(primary), ab, len(aux), zl.FORM_PLAIN, zl.int_to_le(randomness[0][0], 32),
                                  zl.int_to_le(randomness[0][1], 32), zl.FORM_PLAIN, out))
        assert out.raw == want[0]
    dpk.free()


def test_bls377_proof_2p20_staged_upload_and_early_lanes(ctx, monkeypatch):
    """configs[3] at a quarter of its size, in the SHIPPED library's own configuration for large proofs: a BLS12-377
    proof over a 2^20-point domain whose 42 MB assignment takes the four-thread staged upload, with the A and L queries
    evaluated over its first half on the two early lanes while the second half is uploaded (prover.cuh), every query on
    its fixed-base table -- against the C++ oracle, from the key and from a clone."""
    import zeth_amd.lib as zl

    monkeypatch.delenv("ZG_PK_TABLE_MIN", raising=False)
    cid = 1
    randomness = [(0x1234567890ABCDEF1234567890ABCDEF, 0x0FEDCBA0987654321FEDCBA098765432)]
    cs, primary, aux, dpk, want = _key_and_oracle_proofs(ctx, cid, 1040000, randomness)
    assert cs.domain_size() == 1 << 20 and 32 * len(aux) >= 32 << 20
    n = zl.fq_bytes(cid)
    out = ctypes.create_string_buffer(8 * n)
    pb = ctypes.create_string_buffer(zl.encode_scalars(primary), 32 * len(primary))
    ab = ctypes.create_string_buffer(zl.encode_scalars(aux), 32 * len(aux))
    twin = dpk.clone()
    for h in (dpk._h, twin._h, dpk._h):
        zl._check(ctx._L.zg_prove(h, pb, len(primary), ab, len(aux), zl.FORM_PLAIN, zl.int_to_le(randomness[0][0], 32),
                                  zl.int_to_le(randomness[0][1], 32), zl.FORM_PLAIN, out))
        assert out.raw == want[0]
    twin.free()
    dpk.free()


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
def test_g2_msm_2p18_closed_form(ctx, curve):
    """configs[2]'s G2 leg at 2^18 points: P_i = (t0 + i d) G2, uniform scalars, windowed and with the fixed-base table."""
    import zeth_amd
    from zeth_amd.synth import closed_form_scalar, random_scalars_host

    cid, C = CURVES[curve]
    n = 1 << 18
    t0, d = 0x1234567890ABCDEF1234567 + cid, 0xFEDCBA987654321
    bases = zeth_amd.Bases.generate(ctx, cid, zeth_amd.G2, n, t0, d)
    sc = zeth_amd.DeviceBuffer.random_scalars(ctx, cid, n, 18 + cid)
    want = C.G2.mul(closed_form_scalar(random_scalars_host(n, C.r, 18 + cid), t0, d, C.r), C.G2.gen)
    assert bases.msm(sc, n=n) == want
    bases.precompute()
    assert bases.msm(sc, n=n) == want
    # a shard of the same job (point range [n/4, n/2)), as rank 1 of 4 computes it
    lo, cnt = n // 4, n // 4
    words = random_scalars_host(n, C.r, 18 + cid)[:cnt]
    part = zeth_amd.DeviceBuffer.from_bytes(ctx, words.tobytes())
    assert bases.msm(part, first=lo, n=cnt) == C.G2.mul(closed_form_scalar(words, (t0 + lo * d) % C.r, d, C.r), C.G2.gen)
    part.free()
    sc.free()
    bases.free()


@pytest.mark.parametrize("curve,group", [("alt-bn128", 1), ("alt-bn128", 2), ("bls12-377", 1)])
def test_msm_sharded_two_contexts_on_one_device(curve, group):
    """configs[2] behind the C ABI: zg_msm_sharded over two contexts (on a node: two devices), each with its point range
    and its scalar slice resident, with and without the fixed-base table; closed form and the CPU oracle on a prefix."""
    import zeth_amd
    from zeth_amd.synth import closed_form_scalar, random_scalars_host

    cid, C = CURVES[curve]
    G = C.G1 if group == 1 else C.G2
    n = 1 << 16
    t0, d = 0x5A45 + cid, 0x77777
    ctxs = [zeth_amd.Context(0), zeth_amd.Context(0)]
    cuts = [0, n // 3, n]  # unequal ranges
    words = random_scalars_host(n, C.r, 4242 + group)
    shards, scal, counts = [], [], []
    for i, c in enumerate(ctxs):
        lo, hi = cuts[i], cuts[i + 1]
        shards.append(zeth_amd.Bases.generate(c, cid, group, hi - lo, (t0 + lo * d) % C.r, d))
        scal.append(zeth_amd.DeviceBuffer.from_bytes(c, words[lo:hi].tobytes()))
        counts.append(hi - lo)
    want = G.mul(closed_form_scalar(words, t0, d, C.r), G.gen)
    assert zeth_amd.msm_sharded(shards, scal, counts) == want
    for b in shards:
        b.precompute()
    assert zeth_amd.msm_sharded(shards, scal, counts) == want
    # fewer scalars than bases in the last shard, an empty shard
    assert zeth_amd.msm_sharded(shards, scal, [counts[0], 0]) == G.mul(closed_form_scalar(words[: counts[0]], t0, d, C.r), G.gen)
    info = zeth_amd.runtime_info()
    assert info["hip_bound"] and info["hip_runtime_version"] > 0
    for x in scal + shards:
        x.free()
    for c in ctxs:
        c.close()
