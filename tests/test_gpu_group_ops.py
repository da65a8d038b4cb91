"""GPU parity for the ceremony's group-element kernels through the C ABI (zg_group_fft, zg_group_scale,
zg_group_sparse_combine) against the oracle's group law, on the reference's own data and relations:

* powersoftau_challenge.4.bin (testdata/, n = 16): [tau^i] -> [L_j(tau)] as powersoftau_compute_lagrange_evaluations
  does (powersoftau_utils.tcc:830-902), compared with the definition and with iFFT + multi_exp, the comparison the
  reference makes in libzeth/tests/core/evaluator_from_lagrange_test.cpp:44-65;
* the MPC route to a proving key equals generator_from_secrets (libzeth/tests/mpc/groth16/mpc_test.cpp:334-366):
  tau powers -> Lagrange evaluations (group FFT) -> per-variable linear combinations (mpc_utils.tcc:187-242) give
  A_query; the delta^-1 scaling of phase2.tcc:447-470 gives L_query and H_query."""
import os
import random

import pytest

pytestmark = pytest.mark.gpu

import zeth_model as M  # noqa: E402
from zeth_model import BLS12_377, BN254  # noqa: E402

CURVES = {"alt-bn128": (0, BN254), "bls12-377": (1, BLS12_377)}


@pytest.fixture(scope="module")
def ctx():
    import zeth_amd

    c = zeth_amd.Context(0)
    yield c
    c.close()


def lagrange_by_definition(C, G, powers):
    n = len(powers)
    omega_inv = pow(C.omega(n), -1, C.r)
    n_inv = pow(n, -1, C.r)
    return [G.multi_exp(powers, [pow(omega_inv, i * j, C.r) * n_inv % C.r for i in range(n)]) for j in range(n)]


def test_lagrange_evaluations_of_powersoftau_file(ctx, golden_dir):
    import zeth_amd

    C = BN254
    n = 16
    pot, _ = M.pot_load(C, open(os.path.join(golden_dir, "powersoftau_challenge.4.bin"), "rb").read(), n)
    for gid, G, powers in ((1, C.G1, pot["tau_g1"][:n]), (2, C.G2, pot["tau_g2"][:n]), (1, C.G1, pot["alpha_tau_g1"][:n]),
                           (1, C.G1, pot["beta_tau_g1"][:n])):
        got = zeth_amd.group_fft(ctx, 0, gid, powers, inverse=True)
        assert got == lagrange_by_definition(C, G, powers)
        # sum_j L_j(x) = 1: the evaluations add up to the first power
        acc = None
        for P in got:
            acc = G.add(acc, P)
        assert acc == powers[0]
        # the reference's cross-check: iFFT of the unit vector e_j dotted with the powers
        bases = zeth_amd.Bases.from_points(ctx, 0, gid, powers)
        dom = zeth_amd.EvaluationDomain(ctx, 0, n)
        for j in (0, 1, 7, 15):
            coeffs = dom.iFFT([1 if i == j else 0 for i in range(n)])
            assert bases.msm(coeffs) == got[j]
        bases.free()


@pytest.mark.parametrize("curve", sorted(CURVES))
@pytest.mark.parametrize("gid", [1, 2])
def test_group_fft_matches_definition_and_inverts(ctx, curve, gid):
    import zeth_amd

    cid, C = CURVES[curve]
    G = C.G1 if gid == 1 else C.G2
    rng = random.Random(41 + gid)
    n = 8
    pts = [G.mul(rng.randrange(1, C.r), G.gen) for _ in range(n)]
    pts[3] = None  # the identity is a valid input
    omega = C.omega(n)
    fwd = zeth_amd.group_fft(ctx, cid, gid, pts, inverse=False)
    assert fwd == [G.multi_exp(pts, [pow(omega, i * j, C.r) for i in range(n)]) for j in range(n)]
    assert zeth_amd.group_fft(ctx, cid, gid, fwd, inverse=True) == pts
    # n = 1 and n = 2 edge cases
    assert zeth_amd.group_fft(ctx, cid, gid, pts[:1], inverse=True) == pts[:1]
    assert zeth_amd.group_fft(ctx, cid, gid, pts[:2], inverse=False) == [G.add(pts[0], pts[1]), G.add(pts[0], G.neg(pts[1]))]
    with pytest.raises(ValueError):
        zeth_amd.group_fft(ctx, cid, gid, pts[:3])


def test_group_fft_round_trip_at_size(ctx):
    """Size-independent property at n = 2^12 on generated bases: inverse(forward(P)) = P and sum_j L_j = P_0."""
    import zeth_amd

    C = BN254
    n = 1 << 12
    bases = zeth_amd.Bases.generate(ctx, 0, 1, n, 0x1234567, 0x89ABCDE)
    pts = bases.read(0, n)
    bases.free()
    fwd = zeth_amd.group_fft(ctx, 0, 1, pts, inverse=False)
    assert fwd[0] == C.G1.mul(sum(0x1234567 + i * 0x89ABCDE for i in range(n)) % C.r, C.G1.gen)  # out_0 = sum_i P_i
    assert zeth_amd.group_fft(ctx, 0, 1, fwd, inverse=True) == pts


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_mpc_route_to_the_key_equals_generator_from_secrets(ctx, curve):
    import zeth_amd

    cid, C = CURVES[curve]
    G1, G2, r = C.G1, C.G2, C.r
    cs = M.simple_circuit(C, 4)  # 12 constraints, 1 input: domain 16
    tau, alpha, beta, delta = 0xA1B2C3D4E5, 0x1122334455, 0x66778899AA, 0xBBCCDDEEFF
    pk, vk = M.generator_from_secrets(C, cs, tau, alpha, beta, delta)
    n = cs.domain_size()
    nc, nv, ni = cs.num_constraints, cs.num_variables, cs.num_inputs
    # dummy_powersoftau_from_secrets (powersoftau_utils.tcc:350-465)
    tau_g1 = [G1.mul(pow(tau, i, r), G1.gen) for i in range(2 * n - 1)]
    tau_g2 = [G2.mul(pow(tau, i, r), G2.gen) for i in range(n)]
    alpha_tau_g1 = [G1.mul(alpha * pow(tau, i, r) % r, G1.gen) for i in range(n)]
    beta_tau_g1 = [G1.mul(beta * pow(tau, i, r) % r, G1.gen) for i in range(n)]
    lag_g1 = zeth_amd.group_fft(ctx, cid, 1, tau_g1[:n], inverse=True)
    lag_g2 = zeth_amd.group_fft(ctx, cid, 2, tau_g2, inverse=True)
    alpha_lag = zeth_amd.group_fft(ctx, cid, 1, alpha_tau_g1, inverse=True)
    beta_lag = zeth_amd.group_fft(ctx, cid, 1, beta_tau_g1, inverse=True)
    # qap.{A,B,C}_in_Lagrange_basis[j]: constraint index -> coefficient, plus the input-consistency rows of A
    rows = [[{} for _ in range(nv + 1)] for _ in range(3)]
    for i, con in enumerate(cs.constraints):
        for k in range(3):
            for idx, coeff in con[k].items():
                rows[k][idx][i] = (rows[k][idx].get(i, 0) + coeff) % r
    for j in range(ni + 1):
        rows[0][j][nc + j] = 1
    As_g1 = zeth_amd.group_sparse_combine(ctx, cid, 1, lag_g1, rows[0], r)
    Bs_g1 = zeth_amd.group_sparse_combine(ctx, cid, 1, lag_g1, rows[1], r)
    Bs_g2 = zeth_amd.group_sparse_combine(ctx, cid, 2, lag_g2, rows[1], r)
    Cs_g1 = zeth_amd.group_sparse_combine(ctx, cid, 1, lag_g1, rows[2], r)
    assert As_g1 == pk.A_query
    assert [(j, Bs_g2[j], Bs_g1[j]) for j in range(nv + 1) if Bs_g2[j] is not None] == pk.B_query
    # ABC_j = beta A_j + alpha B_j + C_j, then L = delta^-1 ABC over the auxiliary variables (phase2.tcc:447-456)
    bA = zeth_amd.group_sparse_combine(ctx, cid, 1, beta_lag, rows[0], r)
    aB = zeth_amd.group_sparse_combine(ctx, cid, 1, alpha_lag, rows[1], r)
    ABC = [G1.add(G1.add(bA[j], aB[j]), Cs_g1[j]) for j in range(nv + 1)]
    dinv = pow(delta, -1, r)
    assert zeth_amd.group_scale(ctx, cid, 1, ABC[ni + 1:], dinv) == pk.L_query
    assert ABC[: ni + 1] == vk.ABC_g1
    # H_i = delta^-1 [t(x) x^i] with t(x) x^i = x^(n+i) - x^i (mpc_utils.tcc:174-178, phase2.tcc:457-470)
    t_x_pow_i = [G1.add(tau_g1[n + i], G1.neg(tau_g1[i])) for i in range(n - 1)]
    assert zeth_amd.group_scale(ctx, cid, 1, t_x_pow_i, dinv) == pk.H_query
    assert zeth_amd.group_scale(ctx, cid, 2, [G2.gen, None], delta) == [pk.delta_g2, None]


def test_group_ops_argument_errors(ctx):
    import zeth_amd

    C = BN254
    pts = [C.G1.gen, C.G1.mul(2, C.G1.gen)]
    with pytest.raises(zeth_amd.ZethGpuError):
        zeth_amd.group_sparse_combine(ctx, 0, 1, pts, [{5: 1}], C.r)  # column outside the point set
    assert zeth_amd.group_sparse_combine(ctx, 0, 1, pts, [{}, {0: 1, 1: C.r - 1}], C.r) == [None, C.G1.neg(C.G1.gen)]
    assert zeth_amd.group_scale(ctx, 0, 1, [], 5) == []
