"""Device key generation (zg_bases_batch_exp + host QAP evaluation, zeth_amd/keygen.py) against the oracle's
generator_from_secrets (key structure asserted by libzeth/tests/mpc/groth16/mpc_test.cpp:282-331), and the
JoinSplit-shaped synthetic circuit end to end at reduced scale."""
import random

import pytest

pytestmark = pytest.mark.gpu

import cpu_oracle  # noqa: E402
from zeth_model import (BLS12_377, BN254, ProvingKey, R1CS, generator_from_secrets, prove, simple_circuit,  # noqa: E402
                        simple_circuit_assignment, verify)

CURVES = {"alt-bn128": (0, BN254), "bls12-377": (1, BLS12_377)}


@pytest.fixture(scope="module")
def ctx():
    import zeth_amd

    c = zeth_amd.Context(0)
    yield c
    c.close()


def test_curve_constants_match_oracle():
    from zeth_amd.keygen import CURVE_INFO

    for cid, C in CURVES.values():
        I = CURVE_INFO[cid]
        assert (I.r, I.q, I.fr_gen, I.fr_s, I.fr_root, I.g1, I.g2) == (C.r, C.q, C.fr_gen, C.fr_s, C.fr_root, C.G1.gen, C.G2.gen)


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
def test_batch_exp(ctx, curve):
    import zeth_amd

    cid, C = CURVES[curve]
    rng = random.Random(3)
    sc = [0, 1, C.r - 1, 2] + [rng.randrange(C.r) for _ in range(70)]
    for gid, G in ((1, C.G1), (2, C.G2)):
        b = zeth_amd.Bases.batch_exp(ctx, cid, gid, sc)
        pts = b.read(0, len(sc))
        for i in (0, 1, 2, 3, 4, 63, 64, 73):
            assert pts[i] == G.mul(sc[i], G.gen), (gid, i)
        b.free()


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
def test_setup_from_secrets_matches_oracle_key(ctx, curve):
    import zeth_amd
    from zeth_amd.keygen import CsrR1cs, generate_setup_from_secrets

    cid, C = CURVES[curve]
    cs = simple_circuit(C, 4)
    primary, aux = simple_circuit_assignment(C, [10, 12, 14, 16])
    secrets = dict(t=0x1234567, alpha=0xA1FA, beta=0xBE7A, delta=0xDE17A)
    pk_model, vk_model = generator_from_secrets(C, cs, **secrets)
    csr = CsrR1cs.from_rows(cs.num_inputs, cs.num_aux, cs.constraints, C.r)
    assert csr.is_satisfied(primary + aux, C.r)
    dpk, vk, extra = generate_setup_from_secrets(ctx, cid, csr, return_scalars=True, **secrets)
    assert (vk.alpha_g1, vk.beta_g2, vk.delta_g2, vk.ABC_g1) == (vk_model.alpha_g1, vk_model.beta_g2, vk_model.delta_g2, vk_model.ABC_g1)
    dec = lambda name, g: [zeth_amd.lib.decode_point(cid, g, extra["plain"][name][i:i + zeth_amd.lib.affine_bytes(cid, g)])  # noqa: E731
                           for i in range(0, len(extra["plain"][name]), zeth_amd.lib.affine_bytes(cid, g))]
    assert dec("A", 1) == pk_model.A_query
    assert dec("H", 1) == pk_model.H_query
    assert dec("L", 1) == pk_model.L_query
    assert extra["b_idx"] == [e[0] for e in pk_model.B_query]
    assert dec("B2", 2) == [e[1] for e in pk_model.B_query]
    assert dec("B1", 1) == [e[2] for e in pk_model.B_query]
    proof = zeth_amd.Groth16Snark(ctx, cid).generate_proof(dpk, primary, aux, 77, 99)
    assert proof == prove(C, pk_model, primary, aux, 77, 99)
    dpk.free()


def test_joinsplit_shaped_small_end_to_end(ctx):
    """Reduced-scale instance of the benchmark circuit: satisfiable, 0/1-heavy, real key, proof verifies."""
    import zeth_amd
    from zeth_amd.keygen import generate_setup_from_secrets
    from zeth_amd.synth_circuit import joinsplit_shaped

    C = BN254
    cs, primary, aux = joinsplit_shaped(C.r, seed=5, scale=0.002)
    assert cs.is_satisfied(primary + aux, C.r)
    assert sum(1 for x in aux if x in (0, 1)) > 0.85 * len(aux)
    dpk, vk, extra = generate_setup_from_secrets(ctx, 0, cs, t=0xABCDEF12345, alpha=5, beta=7, delta=11, return_scalars=True)
    snark = zeth_amd.Groth16Snark(ctx, 0)
    proof = snark.generate_proof(dpk, primary, aux, 0x1111, 0x2222)
    # oracle on the same key
    ab1, ab2 = 64, 128
    pl = extra["plain"]
    pk = ProvingKey()
    rows = []
    for i in range(cs.num_constraints):
        rows.append(tuple({int(cs.col[k][t]): cs.val[k][t] for t in range(int(cs.row_ptr[k][i]), int(cs.row_ptr[k][i + 1]))} for k in range(3)))
    pk.cs = R1CS(cs.num_inputs, cs.num_variables - cs.num_inputs, rows)
    d1 = lambda b: [zeth_amd.lib.decode_point(0, 1, b[i:i + ab1]) for i in range(0, len(b), ab1)]  # noqa: E731
    d2 = lambda b: [zeth_amd.lib.decode_point(0, 2, b[i:i + ab2]) for i in range(0, len(b), ab2)]  # noqa: E731
    pk.alpha_g1, pk.beta_g1, pk.beta_g2, pk.delta_g1, pk.delta_g2 = extra["points"]
    pk.A_query, pk.H_query, pk.L_query = d1(pl["A"]), d1(pl["H"]), d1(pl["L"])
    pk.B_query = list(zip(extra["b_idx"], d2(pl["B2"]), d1(pl["B1"])))
    assert proof == cpu_oracle.prove(0, C, pk, primary, aux, 0x1111, 0x2222, chunks=4)
    assert verify(C, vk, primary, proof)
    dpk.free()
