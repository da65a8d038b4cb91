"""Model-level restatement of the reference's self-consistency tests for the path:
proof verifies (simple_test.cpp:68-73,134-154), wrong input fails, unsatisfied witness yields a
non-verifying proof silently (prover_test.cpp:703-714), iFFT+MSM == group FFT
(evaluator_from_lagrange_test.cpp:44-65), key structure from secrets (mpc_test.cpp:282-331)."""
import random

import pytest

from zeth_model import (
    BLS12_377,
    BN254,
    Domain,
    dft_naive,
    fft_inplace,
    generator_from_secrets,
    prove,
    r1cs_to_qap_witness_map,
    simple_circuit,
    simple_circuit_assignment,
    verify,
)


@pytest.mark.parametrize("C", [BN254, BLS12_377], ids=lambda c: c.name)
def test_fft_matches_dft_definition(C):
    rng = random.Random(1)
    for n in (2, 4, 16):
        a = [rng.randrange(C.r) for _ in range(n)]
        w = C.omega(n)
        b = list(a)
        fft_inplace(b, w, C.r)
        assert b == dft_naive(a, w, C.r)
        dom = Domain(C, n)
        c = list(a)
        dom.FFT(c)
        dom.iFFT(c)
        assert c == a
        g = C.fr_gen
        dom.cosetFFT(c, g)
        assert c == [sum(a[j] * pow(g * pow(w, k, C.r), j, C.r) for j in range(n)) % C.r for k in range(n)]
        dom.icosetFFT(c, g)
        assert c == a


@pytest.mark.parametrize("C", [BN254, BLS12_377], ids=lambda c: c.name)
def test_simple_circuit_prove_verify(C):
    cs = simple_circuit(C)
    primary, aux = simple_circuit_assignment(C, [78])  # simple_test.cpp:52
    assert cs.is_satisfied(primary + aux, C.r)
    assert cs.domain_size() == 8  # 3 constraints + 1 input + 1 -> 5 -> 8
    pk, vk = generator_from_secrets(C, cs, t=0x1234567, alpha=0xA1FA, beta=0xBE7A, delta=0xDE17A)
    assert len(pk.A_query) == cs.num_variables + 1
    assert len(pk.H_query) == cs.domain_size() - 1
    assert len(pk.L_query) == cs.num_variables - cs.num_inputs
    proof = prove(C, pk, primary, aux, r_rand=0x1111, s_rand=0x2222)
    assert verify(C, vk, primary, proof)
    assert not verify(C, vk, [(primary[0] + 1) % C.r], proof)
    # H really is (A*B - C)/Z: degree m-2, top coefficients zero
    Hc = r1cs_to_qap_witness_map(C, cs, primary, aux)
    assert len(Hc) == cs.domain_size() + 1 and Hc[-1] == 0 and Hc[-2] == 0
    # unsatisfied witness: no error, proof just fails to verify
    bad_aux = list(aux)
    bad_aux[1] = (bad_aux[1] + 1) % C.r
    bad = prove(C, pk, primary, bad_aux, r_rand=0x1111, s_rand=0x2222)
    assert not verify(C, vk, primary, bad)


def test_stacked_circuit_prove_verify_bn254():
    C = BN254
    cs = simple_circuit(C, 4)
    primary, aux = simple_circuit_assignment(C, [10, 12, 14, 16])  # common_snark_tests.tcc:46-54
    assert (cs.num_inputs, cs.num_aux, cs.num_constraints, cs.domain_size()) == (1, 15, 12, 16)
    assert cs.is_satisfied(primary + aux, C.r)
    pk, vk = generator_from_secrets(C, cs, t=987654321, alpha=3, beta=5, delta=7)
    proof = prove(C, pk, primary, aux, r_rand=C.r - 2, s_rand=12345)
    assert verify(C, vk, primary, proof)


def test_key_structure_from_secrets_bn254():
    # mpc_test.cpp:291-314: H_i = Z(t) t^i / delta, L_i = (beta A_j + alpha B_j + C_j)(t) / delta
    C = BN254
    cs = simple_circuit(C)
    t, alpha, beta, delta = 11, 13, 17, 19
    pk, vk = generator_from_secrets(C, cs, t, alpha, beta, delta)
    m = cs.domain_size()
    Zt = (pow(t, m, C.r) - 1) % C.r
    dinv = pow(delta, -1, C.r)
    for i in range(m - 1):
        assert pk.H_query[i] == C.G1.mul(Zt * pow(t, i, C.r) * dinv, C.G1.gen)
    assert pk.delta_g1 == C.G1.mul(delta, C.G1.gen) and vk.delta_g2 == C.G2.mul(delta, C.G2.gen)
