"""GPU parity for the C++ host module (include/libzeth_gpu/gpu_groth16_snark.hpp) and the zeth-tool `prove`
counterpart (zeth_amd/host/zeth_gpu_tool.cpp): every result crosses Zeth's on-disk formats, is produced by the C++
adapter calling the C ABI, and is compared with the oracle bit for bit.

Mirrors the reference's own tests: proof round trip and verification on the simple circuit
(libzeth/tests/snarks/common_snark_tests.tcc:37-62, libzeth/tests/circuits/simple_test.cpp:134-154), iFFT + multi_exp
against the Lagrange evaluations on real powersoftau data (libzeth/tests/core/evaluator_from_lagrange_test.cpp:21-66
with testdata/powersoftau_challenge.4.bin)."""
import os
import random
import subprocess

import pytest

pytestmark = pytest.mark.gpu

import zeth_model as M  # noqa: E402
from zeth_model import BLS12_377, BN254  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "build", "host_adapter_test")
TOOL = os.path.join(ROOT, "build", "zeth-gpu-tool")
CURVES = {"alt-bn128": BN254, "bls12-377": BLS12_377}


def run(*args, expect=0):
    p = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=110)
    assert p.returncode == expect, "exit %d (wanted %d)\nstdout: %s\nstderr: %s" % (p.returncode, expect, p.stdout, p.stderr)
    return p


def simple_instance(C, xs):
    cs = M.simple_circuit(C, len(xs))
    pk, vk = M.generator_from_secrets(C, cs, 0x1234567, 0x2345678, 0x3456789, 0x456789A)
    primary, aux = M.simple_circuit_assignment(C, xs)
    return cs, pk, vk, primary, aux


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_tool_prove_matches_oracle(curve, tmp_path):
    C = CURVES[curve]
    cs, pk, vk, primary, aux = simple_instance(C, [10, 12, 14, 16])
    (tmp_path / "pk").write_bytes(M.proving_key_to_bytes(C, pk))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    rng = random.Random(7)
    r, s = rng.randrange(C.r), rng.randrange(C.r)
    out = run(TOOL, "prove", "-c", curve, "--profile", "--randomness", "%x" % r, "0x%x" % s, tmp_path / "pk", tmp_path / "asg",
              tmp_path / "proof")
    assert "Writing proof to file" in out.stdout and "(profile) generate_proof" in out.stdout
    want = M.prove(C, pk, primary, aux, r, s)
    assert (tmp_path / "proof").read_bytes() == M.proof_to_bytes(C, want)
    # default randomness (drawn inside the prover, as libsnark does): a different proof each run, each one valid
    seen = set()
    for k in range(2):
        run(TOOL, "prove", "--curve", curve, "-p", 1, tmp_path / "pk", tmp_path / "asg", tmp_path / ("p%d" % k))
        raw = (tmp_path / ("p%d" % k)).read_bytes()
        assert M.verify(C, vk, primary, M.proof_from_bytes(C, raw))
        seen.add(raw)
    assert len(seen) == 2


def test_tool_prove_unsatisfied_assignment_is_not_an_error(tmp_path):
    """libzeth/tests/prover/prover_test.cpp:703-714: a bad witness yields a proof that fails to verify."""
    C = BN254
    cs, pk, vk, primary, aux = simple_instance(C, [78])
    aux = list(aux)
    aux[1] = (aux[1] + 1) % C.r
    (tmp_path / "pk").write_bytes(M.proving_key_to_bytes(C, pk))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    run(TOOL, "prove", tmp_path / "pk", tmp_path / "asg", tmp_path / "proof")
    assert not M.verify(C, vk, primary, M.proof_from_bytes(C, (tmp_path / "proof").read_bytes()))
    # wrong number of values for this key: std::invalid_argument -> exit 1
    (tmp_path / "short").write_bytes(M.assignment_to_bytes(C, primary + aux[:-1]))
    assert "invalid argument" in run(TOOL, "prove", tmp_path / "pk", tmp_path / "short", tmp_path / "proof2", expect=1).stderr


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_adapter_generate_proof(curve, tmp_path):
    """groth16_snark<ppT>::generate_proof(pk, primary, auxiliary) through the stateless entry point (resident key
    reused on the second call) and device_proving_key::prove with fixed (r, s)."""
    C = CURVES[curve]
    cs, pk, vk, primary, aux = simple_instance(C, [3, 5])
    (tmp_path / "pk").write_bytes(M.proving_key_to_bytes(C, pk))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    r, s = 0x1F2E3D4C5B6A7988 % C.r, 0x123456789ABCDEF0123456789 % C.r
    run(DRIVER, "prove", curve, tmp_path / "pk", tmp_path / "asg", 1, "%x" % r, "%x" % s, tmp_path / "fixed", tmp_path / "r1", tmp_path / "r2")
    assert (tmp_path / "fixed").read_bytes() == M.proof_to_bytes(C, M.prove(C, pk, primary, aux, r, s))
    p1, p2 = (tmp_path / "r1").read_bytes(), (tmp_path / "r2").read_bytes()
    assert p1 != p2
    assert M.verify(C, vk, primary, M.proof_from_bytes(C, p1)) and M.verify(C, vk, primary, M.proof_from_bytes(C, p2))


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_adapter_generate_proof_from_protoboard_as_first_call(curve, tmp_path):
    """groth16_snark<ppT>::generate_proof(pk, pb) (groth16_snark.hpp:32-35, .tcc:27-34; what circuit_wrapper::prove
    calls, circuit_wrapper.tcc:169-170) as the FIRST call into the library; the same host key object then changed in
    place (the device copy kept by the stateless entry point must be rebuilt); device keys released and proved again;
    clean exit (no use of a destroyed context in static destructors)."""
    C = CURVES[curve]
    cs, pk, vk, primary, aux = simple_instance(C, [3, 5])
    pk2, vk2 = M.generator_from_secrets(C, cs, 0x7654321, 0x2345678, 0x3456789, 0x456789A)
    (tmp_path / "pk").write_bytes(M.proving_key_to_bytes(C, pk))
    (tmp_path / "pk2").write_bytes(M.proving_key_to_bytes(C, pk2))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    run(DRIVER, "prove_pb", curve, tmp_path / "pk", tmp_path / "asg", 1, tmp_path / "pk2", tmp_path / "p1", tmp_path / "p2", tmp_path / "p3")
    p1, p2, p3 = ((tmp_path / n).read_bytes() for n in ("p1", "p2", "p3"))
    assert M.verify(C, vk, primary, M.proof_from_bytes(C, p1))
    assert M.verify(C, vk2, primary, M.proof_from_bytes(C, p2)) and not M.verify(C, vk, primary, M.proof_from_bytes(C, p2))
    assert M.verify(C, vk2, primary, M.proof_from_bytes(C, p3))


vk_to_bytes = lambda C, vk: M.verification_key_to_bytes(C, vk)  # noqa: E731
vk_from_bytes = M.verification_key_from_bytes


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_adapter_generator_from_secrets_and_generate_setup(curve, tmp_path):
    """r1cs_gg_ppzksnark_generator_from_secrets in C++ (QAP evaluation at t on the host, batch_exp on the device) equals
    the oracle's key byte for byte in the reference's on-disk formats -- the comparison mpc_test.cpp:344-365 makes
    between the ceremony's key and the generator's; generate_setup (groth16_snark.tcc:18-25) with fresh secrets gives a
    key whose proof verifies under its own verification key."""
    C = CURVES[curve]
    cs = M.simple_circuit(C, 4)
    primary, aux = M.simple_circuit_assignment(C, [10, 12, 14, 16])
    t, alpha, beta, delta = 0x1234567ABCDEF, 0x2345678, 0x3456789, 0x456789A0F
    pk, vk = M.generator_from_secrets(C, cs, t, alpha, beta, delta)
    (tmp_path / "r1cs").write_bytes(M.r1cs_to_bytes(C, cs))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    run(DRIVER, "setup", curve, tmp_path / "r1cs", "%x" % t, "%x" % alpha, "%x" % beta, "%x" % delta, tmp_path / "pk", tmp_path / "vk",
        tmp_path / "proof", tmp_path / "asg", len(primary))
    assert (tmp_path / "pk").read_bytes() == M.proving_key_to_bytes(C, pk)
    assert (tmp_path / "vk").read_bytes() == vk_to_bytes(C, vk)
    assert (tmp_path / "pk.keypair").read_bytes() == M.keypair_to_bytes(C, pk, vk)  # keypair_write_bytes (groth16_snark.tcc:177-183)
    fresh_vk = vk_from_bytes(C, (tmp_path / "vk.fresh").read_bytes())
    assert fresh_vk.alpha_g1 != vk.alpha_g1
    proof = M.proof_from_bytes(C, (tmp_path / "proof").read_bytes())
    assert M.verify(C, fresh_vk, primary, proof)
    assert not M.verify(C, vk, primary, proof)


@pytest.mark.parametrize("curve,group", [("alt-bn128", "g1"), ("alt-bn128", "g2"), ("bls12-377", "g1"), ("bls12-377", "g2")])
def test_adapter_multi_exp(curve, group, tmp_path):
    """libzeth::multi_exp<ppT, GroupT>(gs, fs) (multi_exp.tcc:25-35): more bases than scalars is allowed and uses the
    first fs.size(); fewer is an error."""
    C = CURVES[curve]
    G = C.G1 if group == "g1" else C.G2
    to_bytes = M.g1_vector_to_bytes if group == "g1" else M.g2_vector_to_bytes
    one_to_bytes = M.g1_to_bytes if group == "g1" else M.g2_to_bytes
    rng = random.Random(11)
    n = 300
    pts = [G.mul(rng.randrange(1, C.r), G.gen) for _ in range(8)]
    gs = [pts[i % 8] if i % 17 else None for i in range(n + 5)]
    fs = [rng.choice([0, 1, C.r - 1, rng.randrange(C.r)]) for _ in range(n)]
    (tmp_path / "gs").write_bytes(to_bytes(C, gs))
    (tmp_path / "fs").write_bytes(M.assignment_to_bytes(C, fs))
    run(DRIVER, "multi_exp", curve, group, tmp_path / "gs", tmp_path / "fs", tmp_path / "out")
    assert (tmp_path / "out").read_bytes() == one_to_bytes(C, G.multi_exp(gs[:n], fs))
    (tmp_path / "few").write_bytes(to_bytes(C, gs[:10]))
    run(DRIVER, "multi_exp", curve, group, tmp_path / "few", tmp_path / "fs", tmp_path / "out2", expect=10)


@pytest.mark.parametrize("curve,group,shards", [("alt-bn128", "g1", 3), ("alt-bn128", "g2", 2), ("bls12-377", "g1", 2), ("bls12-377", "g2", 4)])
def test_adapter_multi_exp_sharded(curve, group, shards, tmp_path):
    """sharded_bases<ppT, GroupT>::multi_exp over zg_msm_sharded: the bases cut into point ranges, one context (here all on
    device 0; on a node one per device) and one host thread per range, partial points added on the host -- against the
    oracle's multi_exp (libzeth::multi_exp, multi_exp.tcc:25-35), also with fewer scalars than bases."""
    C = CURVES[curve]
    G = C.G1 if group == "g1" else C.G2
    to_bytes = M.g1_vector_to_bytes if group == "g1" else M.g2_vector_to_bytes
    one_to_bytes = M.g1_to_bytes if group == "g1" else M.g2_to_bytes
    rng = random.Random(13 + shards)
    n = 1500
    pts = [G.mul(rng.randrange(1, C.r), G.gen) for _ in range(8)]
    gs = [pts[i % 8] if i % 19 else None for i in range(n + 40)]
    fs = [rng.choice([0, 1, C.r - 1, rng.randrange(C.r)]) for _ in range(n)]
    (tmp_path / "gs").write_bytes(to_bytes(C, gs))
    (tmp_path / "fs").write_bytes(M.assignment_to_bytes(C, fs))
    run(DRIVER, "multi_exp_sharded", curve, group, tmp_path / "gs", tmp_path / "fs", tmp_path / "out", shards)
    assert (tmp_path / "out").read_bytes() == one_to_bytes(C, G.multi_exp(gs[:n], fs))


def test_adapter_proving_key_pool(tmp_path):
    """device_proving_key_pool over two devices' worth of resident keys (here two copies on device 0, one prover each)."""
    C = CURVES["alt-bn128"]
    cs, pk, vk, primary, aux = simple_instance(C, [7, 9])
    (tmp_path / "pk").write_bytes(M.proving_key_to_bytes(C, pk))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    r, s = 0xABCDEF % C.r, 0x123456789 % C.r
    run(DRIVER, "prove_pool", "alt-bn128", tmp_path / "pk", tmp_path / "asg", 1, "%x" % r, "%x" % s, tmp_path / "a", tmp_path / "b")
    want = M.proof_to_bytes(C, M.prove(C, pk, primary, aux, r, s))
    assert (tmp_path / "a").read_bytes() == want and (tmp_path / "b").read_bytes() == want


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_adapter_proofs_in_flight_through_the_pool(curve, tmp_path):
    """BASELINE configs[4] as prover_server would bind it (prover_server.cpp:118-201: concurrent handlers, one shared
    keypair): four threads prove through ONE device_proving_key_pool (device 0, four provers = the uploaded key and three
    zg_pk_clone clones created on demand).  Every proof is byte-equal to the oracle's for its own (r, s); the proofs
    overlap (peak in flight > 1); the stateless generate_proof called from the same four threads at once gives valid
    proofs as well."""
    C = CURVES[curve]
    cs, pk, vk, primary, aux = simple_instance(C, [3, 5, 7])
    (tmp_path / "pk").write_bytes(M.proving_key_to_bytes(C, pk))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    r, s = 0x1F2E3D4C5B6A7988, 0x123456789ABCDEF012345
    threads, per = 4, 6
    out = run(DRIVER, "prove_concurrent", curve, tmp_path / "pk", tmp_path / "asg", 1, "%x" % r, "%x" % s, threads, per, tmp_path / "p")
    words = out.stdout.split()
    peak, done = int(words[words.index("peak_in_flight") + 1]), int(words[words.index("proofs") + 1])
    assert done == threads * per and 2 <= peak <= 4, out.stdout
    # pool.warm_up(): all three provers of a fresh pool created and used once (3 held at once, 3 + 1 proofs), same proof after
    assert "warm_up provers 3 proofs 4 same 1" in out.stdout, out.stdout
    for k in range(threads * per):
        want = M.proof_to_bytes(C, M.prove(C, pk, primary, aux, r + k, s + 2 * k + 1))
        assert (tmp_path / ("p.%d" % k)).read_bytes() == want, "proof %d" % k
    seen = set()
    for t in range(threads):
        raw = (tmp_path / ("p.s%d" % t)).read_bytes()
        assert M.verify(C, vk, primary, M.proof_from_bytes(C, raw))
        seen.add(raw)
    assert len(seen) == threads


def test_generate_setup_swaps_a_and_b_when_beneficial(tmp_path):
    """r1cs_gg_ppzksnark_generator on a system whose B side touches more variables than its A side: the key is built
    for the swapped system (swap_AB_if_beneficial) and its proof verifies under its own verification key."""
    C = BN254
    cs = M.simple_circuit(C, 2)
    flipped = M.R1CS(cs.num_inputs, cs.num_aux, [(b, a, c) for a, b, c in cs.constraints])
    # one of the two orientations is the one the generator swaps
    swapped = [M.swap_AB_if_beneficial(x) is not x for x in (cs, flipped)]
    assert swapped.count(True) == 1
    target = flipped if swapped[1] else cs
    primary, aux = M.simple_circuit_assignment(C, [10, 12])
    (tmp_path / "r1cs").write_bytes(M.r1cs_to_bytes(C, target))
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, primary + aux))
    run(DRIVER, "setup", "alt-bn128", tmp_path / "r1cs", "1234567abcdef", "2345678", "3456789", "456789a0f", tmp_path / "pk", tmp_path / "vk",
        tmp_path / "proof", tmp_path / "asg", len(primary))
    fresh_vk = vk_from_bytes(C, (tmp_path / "vk.fresh").read_bytes())
    assert M.verify(C, fresh_vk, primary, M.proof_from_bytes(C, (tmp_path / "proof").read_bytes()))
    # the from-secrets form does not swap (mpc_test.cpp:340 applies the swap itself before comparing)
    pk, vk = M.generator_from_secrets(C, target, 0x1234567ABCDEF, 0x2345678, 0x3456789, 0x456789A0F)
    assert (tmp_path / "pk").read_bytes() == M.proving_key_to_bytes(C, pk)


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_adapter_domain(curve, tmp_path):
    C = CURVES[curve]
    rng = random.Random(5)
    m = 64
    a = [rng.randrange(C.r) for _ in range(m)]
    (tmp_path / "in").write_bytes(M.assignment_to_bytes(C, a))
    dom = M.Domain(C, m)
    for op, fn in (("fft", dom.FFT), ("ifft", dom.iFFT), ("cosetfft", lambda v: dom.cosetFFT(v, C.fr_gen)),
                   ("icosetfft", lambda v: dom.icosetFFT(v, C.fr_gen))):
        run(DRIVER, "domain", curve, op, tmp_path / "in", tmp_path / op)
        want = list(a)
        fn(want)
        assert M.assignment_from_bytes((tmp_path / op).read_bytes()) == want, op
    # not a power of two: the domain constructor throws (libfqfft DomainSizeException)
    (tmp_path / "odd").write_bytes(M.assignment_to_bytes(C, a[:48]))
    run(DRIVER, "domain", curve, "fft", tmp_path / "odd", tmp_path / "x", expect=10)


def test_evaluator_from_lagrange_on_powersoftau_data(golden_dir, tmp_path):
    """evaluator_from_lagrange_test.cpp:21-66 on the reference's powersoftau_challenge.4.bin (n = 16): for every j,
    iFFT(e_j) dotted with the tau powers equals [L_j(tau)], which is checked here against the definition
    L_j = (1/n) sum_i omega^(-ij) [tau^i] computed with the oracle's group law."""
    C = BN254
    n = 16
    pot, _ = M.pot_load(C, open(os.path.join(golden_dir, "powersoftau_challenge.4.bin"), "rb").read(), n)
    omega_inv = pow(C.omega(n), -1, C.r)
    n_inv = pow(n, -1, C.r)
    for group, G, powers, to_bytes in (("g1", C.G1, pot["tau_g1"][:n], M.g1_vector_to_bytes),
                                       ("g2", C.G2, pot["tau_g2"][:n], M.g2_vector_to_bytes),
                                       ("g1", C.G1, pot["alpha_tau_g1"][:n], M.g1_vector_to_bytes)):
        (tmp_path / "powers").write_bytes(to_bytes(C, powers))
        run(DRIVER, "lagrange", "alt-bn128", group, tmp_path / "powers", tmp_path / "out")
        want = [G.multi_exp(powers, [pow(omega_inv, i * j, C.r) * n_inv % C.r for i in range(n)]) for j in range(n)]
        assert (tmp_path / "out").read_bytes() == to_bytes(C, want)
