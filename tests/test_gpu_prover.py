"""GPU parity for the prover path through the C ABI (zg_pk_create / zg_witness_map / zg_prove) against the
oracles, on the reference's own fixture circuits (libzeth/tests/circuits/simple_test.tcc:15-62, 4x stacked as in
libzeth/tests/snarks/common_snark_tests.tcc:37-62) and on random satisfiable R1CS.  With (r, s) fixed the proof is
a deterministic group element triple: bit-exact comparison, plus the KAT-pinned verifier equation."""
import random

import pytest

pytestmark = pytest.mark.gpu

import cpu_oracle  # noqa: E402
from zeth_model import (BLS12_377, BN254, R1CS, generator_from_secrets, prove, r1cs_to_qap_witness_map,  # noqa: E402
                        simple_circuit, simple_circuit_assignment, verify)

CURVES = {"alt-bn128": (0, BN254), "bls12-377": (1, BLS12_377)}


@pytest.fixture(scope="module")
def ctx():
    import zeth_amd

    c = zeth_amd.Context(0)
    yield c
    c.close()


def random_r1cs(C, rng, num_inputs, num_aux, nc, bool_frac=0.7):
    """Random satisfiable system: each constraint <a,w> * <b,w> = w_k defines a fresh-ish auxiliary variable or is a
    booleanity constraint x (1 - x) = 0; mirrors the JoinSplit's mix of bit wires and field wires."""
    r = C.r
    nv = num_inputs + num_aux
    w = [1] + [0] * nv
    defined = [0]
    cons = []
    # inputs and a first batch of aux values are free
    free = list(range(1, nv + 1))
    rng.shuffle(free)
    n_defined_by_constraint = min(nc // 2, num_aux // 2)
    targets = free[:n_defined_by_constraint]
    for v in free[n_defined_by_constraint:]:
        w[v] = rng.randrange(2) if rng.random() < bool_frac else rng.randrange(r)
        defined.append(v)
    for v in targets:
        a = {rng.choice(defined): rng.randrange(1, r) for _ in range(rng.randrange(1, 4))}
        b = {rng.choice(defined): rng.randrange(1, r) for _ in range(rng.randrange(1, 3))}
        ev = lambda lc: sum(c * w[i] for i, c in lc.items()) % r  # noqa: E731
        w[v] = ev(a) * ev(b) % r
        cons.append((a, b, {v: 1}))
        defined.append(v)
    while len(cons) < nc:
        v = rng.choice(defined[1:]) if len(defined) > 1 else 0
        if w[v] in (0, 1):
            cons.append(({v: 1}, {0: 1, v: r - 1}, {}))  # x (1 - x) = 0
        else:
            k = rng.randrange(1, r)
            cons.append(({v: k}, {0: 1}, {v: k}))
    rng.shuffle(cons)
    cs = R1CS(num_inputs, num_aux, cons)
    full = w[1:]
    assert cs.is_satisfied(full, r)
    return cs, full[:num_inputs], full[num_inputs:]


@pytest.mark.parametrize("curve", ["alt-bn128", "bls12-377"])
@pytest.mark.parametrize("copies,xs", [(1, [78]), (4, [10, 12, 14, 16])])
def test_simple_circuit_proof_bit_exact_and_verifies(ctx, curve, copies, xs):
    import zeth_amd

    cid, C = CURVES[curve]
    cs = simple_circuit(C, copies)
    primary, aux = simple_circuit_assignment(C, xs)
    pk, vk = generator_from_secrets(C, cs, t=0x1234567, alpha=0xA1FA, beta=0xBE7A, delta=0xDE17A)
    dpk = zeth_amd.ProvingKey.from_model(ctx, cid, pk, C.r)
    assert dpk.domain_size == cs.domain_size()
    snark = zeth_amd.Groth16Snark(ctx, cid)
    assert snark.witness_map(dpk, primary, aux) == r1cs_to_qap_witness_map(C, cs, primary, aux)
    r_rand, s_rand = 0x1111222233334444, 0x5555666677778888
    proof = snark.generate_proof(dpk, primary, aux, r_rand, s_rand)
    assert proof == prove(C, pk, primary, aux, r_rand, s_rand)
    assert proof == cpu_oracle.prove(cid, C, pk, primary, aux, r_rand, s_rand)
    if copies == 1:  # the pairing check in pure Python is slow: once per curve
        assert verify(C, vk, primary, proof)
        assert not verify(C, vk, [(primary[0] + 1) % C.r], proof)
    # different randomness, same statement: different proof
    assert snark.generate_proof(dpk, primary, aux, r_rand + 1, s_rand) != proof
    # fresh OS randomness path
    p2 = snark.generate_proof(dpk, primary, aux, order=C.r)
    assert p2 != proof and C.G1.is_on_curve(p2[0]) and C.G2.is_on_curve(p2[1])
    dpk.free()


def test_unsatisfied_witness_is_not_an_error_but_fails_verification(ctx):
    """prover_test.cpp:703-714: release builds emit a proof that does not verify."""
    import zeth_amd

    C = BN254
    cs = simple_circuit(C)
    primary, aux = simple_circuit_assignment(C, [78])
    pk, vk = generator_from_secrets(C, cs, t=0x7777, alpha=3, beta=5, delta=7)
    dpk = zeth_amd.ProvingKey.from_model(ctx, 0, pk, C.r)
    bad_aux = list(aux)
    bad_aux[1] = (bad_aux[1] + 1) % C.r
    proof = zeth_amd.Groth16Snark(ctx, 0).generate_proof(dpk, primary, bad_aux, 11, 13)
    assert proof == prove(C, pk, primary, bad_aux, 11, 13)
    assert not verify(C, vk, primary, proof)
    dpk.free()


def test_wrong_input_sizes_are_invalid_argument(ctx):
    import zeth_amd

    C = BN254
    cs = simple_circuit(C)
    primary, aux = simple_circuit_assignment(C, [5])
    pk, _ = generator_from_secrets(C, cs, t=9, alpha=3, beta=5, delta=7)
    dpk = zeth_amd.ProvingKey.from_model(ctx, 0, pk, C.r)
    snark = zeth_amd.Groth16Snark(ctx, 0)
    for p, a in ((primary + [1], aux), (primary, aux[:-1]), ([], primary + aux)):
        with pytest.raises(zeth_amd.ZethGpuError) as e:
            snark.generate_proof(dpk, p, a, 1, 2)
        assert e.value.status == -1
    pk.H_query = pk.H_query[:-1]
    with pytest.raises(zeth_amd.ZethGpuError):
        zeth_amd.ProvingKey.from_model(ctx, 0, pk, C.r)
    dpk.free()


@pytest.mark.parametrize("curve,ni,na,nc", [("alt-bn128", 1, 200, 300), ("alt-bn128", 3, 1500, 2500), ("bls12-377", 1, 600, 1000)])
def test_random_r1cs_witness_map_and_proof(ctx, curve, ni, na, nc):
    """0/1-heavy random assignments (the JoinSplit shape) against the C++ oracle; m up to 4096."""
    import zeth_amd

    cid, C = CURVES[curve]
    rng = random.Random(ni * 1000 + nc)
    cs, primary, aux = random_r1cs(C, rng, ni, na, nc)
    snark = zeth_amd.Groth16Snark(ctx, cid)
    pk, _ = generator_from_secrets_fast(C, cs, rng)
    dpk = zeth_amd.ProvingKey.from_model(ctx, cid, pk, C.r)
    assert snark.witness_map(dpk, primary, aux) == cpu_oracle.witness_map(cid, C, cs, primary, aux)
    r_rand, s_rand = rng.randrange(C.r), rng.randrange(C.r)
    assert snark.generate_proof(dpk, primary, aux, r_rand, s_rand) == cpu_oracle.prove(cid, C, pk, primary, aux, r_rand, s_rand, chunks=4)
    dpk.free()


@pytest.mark.parametrize("curve,ni,na,nc", [("alt-bn128", 3, 1500, 2500), ("bls12-377", 1, 600, 1000)])
def test_proof_with_h_query_table(ctx, curve, ni, na, nc, monkeypatch):
    """Large keys keep their H query as a fixed-base table (prover.cuh; threshold lowered here through
    ZG_PK_TABLE_MIN so that a small circuit takes that path): same proof bytes as the oracle, also from a clone."""
    import zeth_amd

    monkeypatch.setenv("ZG_PK_TABLE_MIN", "1023")
    cid, C = CURVES[curve]
    rng = random.Random(ni * 7 + nc)
    cs, primary, aux = random_r1cs(C, rng, ni, na, nc)
    snark = zeth_amd.Groth16Snark(ctx, cid)
    pk, _ = generator_from_secrets_fast(C, cs, rng)
    dpk = zeth_amd.ProvingKey.from_model(ctx, cid, pk, C.r)
    r_rand, s_rand = rng.randrange(C.r), rng.randrange(C.r)
    want = cpu_oracle.prove(cid, C, pk, primary, aux, r_rand, s_rand, chunks=4)
    assert snark.generate_proof(dpk, primary, aux, r_rand, s_rand) == want
    twin = dpk.clone()
    assert snark.generate_proof(twin, primary, aux, r_rand, s_rand) == want
    twin.free()
    dpk.free()


def generator_from_secrets_fast(C, cs, rng):
    """Key with the right SHAPE for parity tests (sizes, sparsity of B) built from cheap group operations: the
    prover is linear in the key, so parity against the oracle on any key of this shape exercises the same path;
    keys with the real Groth16 structure are covered by the simple-circuit tests above."""
    from zeth_model import ProvingKey

    G1, G2 = C.G1, C.G2
    m = cs.domain_size()
    nv, ni = cs.num_variables, cs.num_inputs

    def walk(G, n):
        P = G.mul(rng.randrange(1, C.r), G.gen)
        D = G.mul(rng.randrange(1, C.r), G.gen)
        out = []
        for _ in range(n):
            out.append(P)
            P = G.add(P, D)
        return out

    pk = ProvingKey()
    pk.cs = cs
    pk.alpha_g1, pk.beta_g1, pk.delta_g1 = walk(G1, 3)
    pk.beta_g2, pk.delta_g2 = walk(G2, 2)
    pk.A_query = walk(G1, nv + 1)
    for i in rng.sample(range(nv + 1), max(1, (nv + 1) // 10)):
        pk.A_query[i] = None  # variables absent from every A polynomial
    used_b = sorted({idx for _, b, _ in cs.constraints for idx in b})
    b1, b2 = walk(G1, len(used_b)), walk(G2, len(used_b))
    pk.B_query = [(idx, b2[k], b1[k]) for k, idx in enumerate(used_b)]
    pk.B_domain_size = nv + 1
    pk.H_query = walk(G1, m - 1)
    pk.L_query = walk(G1, nv - ni)
    return pk, None


def _failed_prove_body():
    """Runs in a process of its own (below) against the -DZG_DEBUG_KNOBS build, whose ZG_TEST_FAIL_LANE hook makes the
    queueing of one lane (0-3) or of the H chain (4) report a failure after the other streams were queued."""
    import os

    import zeth_amd

    C = BN254
    rng = random.Random(99)
    ctx = zeth_amd.Context(0)
    cs, primary, aux = random_r1cs(C, rng, 2, 900, 1500)
    pk, _ = generator_from_secrets_fast(C, cs, rng)
    dpk = zeth_amd.ProvingKey.from_model(ctx, 0, pk, C.r)
    snark = zeth_amd.Groth16Snark(ctx, 0)
    want = cpu_oracle.prove(0, C, pk, primary, aux, 5, 7, chunks=2)
    assert snark.generate_proof(dpk, primary, aux, 5, 7) == want
    for lane in ("0", "2", "4"):
        os.environ["ZG_TEST_FAIL_LANE"] = lane
        try:
            snark.generate_proof(dpk, primary, aux, 5, 7)
        except zeth_amd.ZethGpuError as e:
            assert e.status == -4
        else:
            raise AssertionError("the injected failure of lane %s was not reported" % lane)
        del os.environ["ZG_TEST_FAIL_LANE"]
        assert snark.generate_proof(dpk, primary, aux, 5, 7) == want
    dpk.free()
    ctx.close()
    print("failed-prove ok")


def test_failed_prove_is_followed_by_a_correct_proof():
    """An error after the lanes were queued drains every stream before it is returned: the next zg_prove on the same key
    gives the oracle's proof.  The failure is injected through a hook that only the debug build of the library has
    (build/dbg/libzeth_gpu.so, `make dbg`); the shipped library ignores the variable, which is checked too."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbg = os.path.join(root, "build", "dbg", "libzeth_gpu.so")
    assert os.path.exists(dbg), "build/dbg/libzeth_gpu.so is missing: run `make dbg`"
    code = "import sys; sys.path[:0] = [%r, %r, %r]; import test_gpu_prover as T; T._failed_prove_body()" % (
        root, os.path.join(root, "oracle"), os.path.join(root, "tests"))
    env = dict(os.environ, ZETH_GPU_LIBRARY=dbg)
    env.pop("ZG_TEST_FAIL_LANE", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110, env=env)
    assert p.returncode == 0 and "failed-prove ok" in p.stdout, p.stdout + p.stderr


def _staged_upload_body():
    """Runs in a process of its own against the -DZG_DEBUG_KNOBS build with ZG_PAR_UPLOAD_MIN_KB=1 and ZG_PAR_STAGE_KB=1: the
    assignment then takes the staged four-thread upload that the shipped library uses from 32 MB on (1 KB staging halves:
    several chunks per lane, both halves reused; ragged and nearly empty quarters)."""
    import zeth_amd

    ctx = zeth_amd.Context(0)
    # (the last case is large enough for the early half-range evaluations of the A and L queries: >= 8192 auxiliary
    # variables and, with ZG_PK_TABLE_MIN lowered, tables for both)
    for cid, C, ni, na, nc in ((0, BN254, 2, 900, 1500), (1, BLS12_377, 3, 37, 64), (0, BN254, 1, 1031, 700), (0, BN254, 2, 9001, 5000)):
        rng = random.Random(1000 + na)
        cs, primary, aux = random_r1cs(C, rng, ni, na, nc)
        pk, _ = generator_from_secrets_fast(C, cs, rng)
        dpk = zeth_amd.ProvingKey.from_model(ctx, cid, pk, C.r)
        snark = zeth_amd.Groth16Snark(ctx, cid)
        want = cpu_oracle.prove(cid, C, pk, primary, aux, 5, 7, chunks=2)
        for _ in range(2):
            assert snark.generate_proof(dpk, primary, aux, 5, 7) == want
        twin = dpk.clone()
        assert snark.generate_proof(twin, primary, aux, 5, 7) == want
        twin.free()
        dpk.free()
    ctx.close()
    print("staged-upload ok")


def test_staged_parallel_upload_gives_the_same_proof():
    """The upload path of large assignments (four lane threads through page-locked staging, prover.cuh upload_slice), forced
    at test sizes through the debug build's knobs; at full size it is checked by tools/bench_prove.py (BLS12-377, 2^22
    constraints: bit_exact_vs_cpu_oracle in profiles/r3_prove_bls377_2p22.json)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbg = os.path.join(root, "build", "dbg", "libzeth_gpu.so")
    assert os.path.exists(dbg), "build/dbg/libzeth_gpu.so is missing: run `make dbg`"
    code = "import sys; sys.path[:0] = [%r, %r, %r]; import test_gpu_prover as T; T._staged_upload_body()" % (
        root, os.path.join(root, "oracle"), os.path.join(root, "tests"))
    env = dict(os.environ, ZETH_GPU_LIBRARY=dbg, ZG_PAR_UPLOAD_MIN_KB="1", ZG_PAR_STAGE_KB="1", ZG_PK_TABLE_MIN="1023")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110, env=env)
    assert p.returncode == 0 and "staged-upload ok" in p.stdout, p.stdout + p.stderr


def test_shipped_library_ignores_the_fault_injection_knob(ctx, monkeypatch):
    import zeth_amd

    C = BN254
    cs = simple_circuit(C, 2)
    pk, vk = generator_from_secrets(C, cs, 11, 12, 13, 14)
    primary, aux = simple_circuit_assignment(C, [3, 4])
    dpk = zeth_amd.ProvingKey.from_model(ctx, 0, pk, C.r)
    monkeypatch.setenv("ZG_TEST_FAIL_LANE", "0")
    assert zeth_amd.Groth16Snark(ctx, 0).generate_proof(dpk, primary, aux, 5, 7) == prove(C, pk, primary, aux, 5, 7)
    dpk.free()
