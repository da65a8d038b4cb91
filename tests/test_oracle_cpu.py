"""Pins the C++ CPU restatement (oracle/zeth_cpu.cpp) against the big-integer model (oracle/zeth_model.py),
which is itself pinned by the reference's known-answer vectors (tests/test_oracle_kats.py)."""
import random

import pytest

import cpu_oracle
from zeth_model import (BLS12_377, BN254, Domain, generator_from_secrets, prove, r1cs_to_qap_witness_map,
                        simple_circuit, simple_circuit_assignment, verify)

CURVES = [(0, BN254), (1, BLS12_377)]


def _pts(G, rng, n, order):
    P = G.mul(rng.randrange(1, order), G.gen)
    D = G.mul(rng.randrange(1, order), G.gen)
    out = []
    for _ in range(n):
        out.append(P)
        P = G.add(P, D)
    return out


@pytest.mark.parametrize("cid,C", CURVES, ids=lambda x: getattr(x, "name", str(x)))
@pytest.mark.parametrize("group", [1, 2])
def test_msm_methods_agree_with_model(cid, C, group):
    G = C.G1 if group == 1 else C.G2
    rng = random.Random(31 + cid + group)
    n = 70 if group == 1 else 40
    pts = _pts(G, rng, n, C.r)
    pts[5] = None
    pts[7] = pts[6]
    pts[8] = G.neg(pts[6])
    for sc in ([rng.randrange(C.r) for _ in range(n)],
               [rng.choice([0, 1, 1, C.r - 1, rng.randrange(C.r)]) for _ in range(n)],
               [0] * n, [1] * n, [3 if i in (6, 7, 8) else 0 for i in range(n)]):
        want = G.multi_exp(pts, sc)
        for method, chunks in ((0, 1), (0, 4), (1, 1), (1, 3), (2, 1)):
            assert cpu_oracle.msm(cid, group, pts, sc, method, chunks) == want, (method, chunks)
    assert cpu_oracle.msm(cid, group, pts, [], 0, 1) is None
    # multi_exp.tcc:25-35: fewer scalars than bases uses the prefix
    sc = [rng.randrange(C.r) for _ in range(9)]
    assert cpu_oracle.msm(cid, group, pts, sc) == G.multi_exp(pts, sc)


@pytest.mark.parametrize("cid,C", CURVES, ids=lambda x: getattr(x, "name", str(x)))
def test_fft_variants_match_model(cid, C):
    rng = random.Random(9 + cid)
    for n in (2, 8, 64, 512):
        a = [rng.randrange(C.r) for _ in range(n)]
        dom = Domain(C, n)
        for inverse, coset in ((False, False), (True, False), (False, True), (True, True)):
            want = list(a)
            if not inverse and not coset:
                dom.FFT(want)
            elif inverse and not coset:
                dom.iFFT(want)
            elif not inverse and coset:
                dom.cosetFFT(want, C.fr_gen)
            else:
                dom.icosetFFT(want, C.fr_gen)
            assert cpu_oracle.fft(cid, a, inverse, coset) == want, (n, inverse, coset)


@pytest.mark.parametrize("cid,C", CURVES, ids=lambda x: getattr(x, "name", str(x)))
def test_witness_map_and_prove_match_model(cid, C):
    cs = simple_circuit(C, copies=4)  # common_snark_tests.tcc:37-62
    primary, aux = simple_circuit_assignment(C, [10, 12, 14, 16])
    assert cs.is_satisfied(primary + aux, C.r)
    assert cpu_oracle.witness_map(cid, C, cs, primary, aux) == r1cs_to_qap_witness_map(C, cs, primary, aux)
    pk, vk = generator_from_secrets(C, cs, t=0x77AA55, alpha=0x1234, beta=0x5678, delta=0x9ABC)
    r_rand, s_rand = 0xDEADBEEF12345, 0xC0FFEE6789
    want = prove(C, pk, primary, aux, r_rand, s_rand)
    got = cpu_oracle.prove(cid, C, pk, primary, aux, r_rand, s_rand, chunks=2)
    assert got == want
    if cid == 0:
        assert verify(C, vk, primary, got)
