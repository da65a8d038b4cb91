"""C++ host module (include/libzeth_gpu/): Zeth's on-disk formats and error convention, without a GPU.

The expected bytes come from the oracle's restatement of the reference formats (oracle/zeth_model.py, stream-format
section); the inputs are the reference's own fixtures: the hand-made constraint system / proving key / assignment
of libzeth/tests/snarks/groth16/groth16_snark_test.cpp:39-171 and the simple circuit of
libzeth/tests/circuits/simple_test.tcc:15-62.  No golden file of these exists in the reference (its tests only
dump them on request), so the byte-level agreement is between two independent restatements -- "parity
unpinned" at the level of whole files, pinned per element by the encodings KATs of tests/test_oracle_kats.py."""
import os
import subprocess

import pytest

import zeth_model as M
from zeth_model import BLS12_377, BN254, R1CS, ProvingKey

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "build", "host_adapter_test")
TOOL = os.path.join(ROOT, "build", "zeth-gpu-tool")
CURVES = {"alt-bn128": BN254, "bls12-377": BLS12_377}


def run(*args, expect=0):
    p = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert p.returncode == expect, "exit %d (wanted %d)\nstdout: %s\nstderr: %s" % (p.returncode, expect, p.stdout, p.stderr)
    return p


def reference_test_data(C):
    """generate_test_data() of groth16_snark_test.cpp:39-171 in the oracle's types."""
    G1, G2, r = C.G1, C.G2, C.r
    P1 = lambda k: G1.mul(k % r, G1.gen)  # noqa: E731
    P2 = lambda k: G2.mul(k % r, G2.gen)  # noqa: E731
    # terms in the order the test adds them
    cs = R1CS(2, 3, [
        ([(1, 2), (3, 3)], [(2, 4), (4, 5)], [(5, 6), (1, 7)]),
        ([(2, 7), (5, 6)], [(1, 5), (3, 4)], [(4, 3), (2, 2)]),
    ])
    pk = ProvingKey()
    pk.alpha_g1, pk.beta_g1, pk.beta_g2, pk.delta_g1, pk.delta_g2 = P1(1), P1(-1), P2(-1), P1(-2), P2(-2)
    pk.A_query = [P1(7), P1(-3), P1(8), P1(-4), P1(9), P1(-5)]
    pk.B_query = [(i, P2(k), P1(k)) for i, k in enumerate([9, -9, 10, -11, 12, -13])]
    pk.B_domain_size = 6
    pk.H_query = [P1(11), P1(-11), P1(12)]
    pk.L_query = [P1(13), P1(-13), P1(14)]
    pk.cs = cs
    assignment = [15, -15 % r, 16, -16 % r, 17, -17 % r]
    proof = (P1(21), P2(-21), P1(22))
    return cs, pk, assignment, proof


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_reference_test_data_bytes(curve, tmp_path):
    C = CURVES[curve]
    (tmp_path / "g1_one").write_bytes(M.g1_vector_to_bytes(C, [C.G1.gen]))
    (tmp_path / "g2_one").write_bytes(M.g2_vector_to_bytes(C, [C.G2.gen]))
    run(DRIVER, "testdata", curve, tmp_path)
    cs, pk, assignment, proof = reference_test_data(C)
    assert (tmp_path / ("groth16_r1cs_" + curve)).read_bytes() == M.r1cs_to_bytes(C, cs)
    assert (tmp_path / ("groth16_proving_key_" + curve)).read_bytes() == M.proving_key_to_bytes(C, pk)
    assert (tmp_path / ("groth16_assignment_" + curve)).read_bytes() == M.assignment_to_bytes(C, assignment)
    assert (tmp_path / ("groth16_proof_" + curve)).read_bytes() == M.proof_to_bytes(C, proof)
    assert (tmp_path / ("groth16_proof_" + curve + ".json")).read_text() == M.proof_to_json(C, proof)
    # sizes stated by SURVEY 8f-1: proof = 256 B on alt-bn128, 384 B on BLS12-377
    assert len(M.proof_to_bytes(C, proof)) == (256 if curve == "alt-bn128" else 384)


def simple_key(C, copies=1):
    cs = M.simple_circuit(C, copies)
    pk, vk = M.generator_from_secrets(C, cs, 0x1234567, 0x2345678, 0x3456789, 0x456789A)
    return cs, pk, vk


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_proving_key_and_assignment_round_trip(curve, tmp_path):
    """proving_key_read_bytes -> proving_key_write_bytes and the (primary, auxiliary) assignment reader/writer
    (common_snark_tests.tcc:64-122, r1cs_variable_assignment_serialization_test.cpp:14-85)."""
    C = CURVES[curve]
    cs, pk, _ = simple_key(C, 4)
    primary, aux = M.simple_circuit_assignment(C, [10, 12, 14, 16])
    pk_bytes = M.proving_key_to_bytes(C, pk)
    asg_bytes = M.assignment_to_bytes(C, primary + aux)
    (tmp_path / "pk").write_bytes(pk_bytes)
    (tmp_path / "asg").write_bytes(asg_bytes)
    out = run(DRIVER, "codec", curve, tmp_path / "pk", tmp_path / "pk2", tmp_path / "asg", tmp_path / "asg2", 1)
    assert "%d A" % (cs.num_variables + 1) in out.stdout and "1+%d assignment" % len(aux) in out.stdout
    assert (tmp_path / "pk2").read_bytes() == pk_bytes
    assert (tmp_path / "asg2").read_bytes() == asg_bytes


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_keypair_file_written_by_the_oracle_is_read_by_the_host_module(curve, tmp_path):
    """keypair_read_bytes / keypair_write_bytes / verification_key_{read,write}_bytes / verification_key_write_json
    (groth16_snark.tcc:58-91, 113-128, 177-191): prover_server loads ~/zeth_setup/keypair.bin with exactly these
    (prover_server.cpp:456-463, 487-490).  The file is written by the oracle's codec and every writer's output is
    compared with it."""
    C = CURVES[curve]
    cs, pk, vk = simple_key(C, 2)
    kp_bytes = M.keypair_to_bytes(C, pk, vk)
    (tmp_path / "kp").write_bytes(kp_bytes)
    out = run(DRIVER, "keypair", curve, tmp_path / "kp", tmp_path / "pk", tmp_path / "vk", tmp_path / "vk.json", tmp_path / "kp2")
    assert "%d A, %d ABC" % (cs.num_variables + 1, cs.num_inputs + 1) in out.stdout
    assert (tmp_path / "pk").read_bytes() == M.proving_key_to_bytes(C, pk)
    assert (tmp_path / "vk").read_bytes() == M.verification_key_to_bytes(C, vk)
    assert (tmp_path / "kp2").read_bytes() == kp_bytes
    assert (tmp_path / "vk.json").read_text() == M.verification_key_to_json(C, vk)
    back = M.verification_key_from_bytes(C, (tmp_path / "vk").read_bytes())
    assert back.ABC_g1 == vk.ABC_g1 and back.alpha_g1 == vk.alpha_g1
    # a verification key that is not well-formed: std::invalid_argument (groth16_snark.tcc:125-127)
    bad = bytearray(kp_bytes)
    bad[-1] ^= 1  # low byte of the last ABC point's Y
    (tmp_path / "bad").write_bytes(bytes(bad))
    assert "verification key (read) not well-formed" in run(
        DRIVER, "keypair", curve, tmp_path / "bad", tmp_path / "o1", tmp_path / "o2", tmp_path / "o3", tmp_path / "o4", expect=10).stderr
    # the file cut inside the verification key: a stream failure
    (tmp_path / "cut").write_bytes(kp_bytes[:-40])
    run(DRIVER, "keypair", curve, tmp_path / "cut", tmp_path / "o1", tmp_path / "o2", tmp_path / "o3", tmp_path / "o4", expect=12)


def test_swap_AB_if_beneficial(tmp_path):
    """r1cs_gg_ppzksnark_generator works on a copy of the system with swap_AB_if_beneficial() applied (the reference's
    callers of the from-secrets form apply it themselves: mpc_test.cpp:42,164,340): a and b change places exactly when
    B touches more variables than A."""
    C = BN254
    # B touches 4 variables, A touches 2: swapped
    dense_b = R1CS(1, 4, [({1: 1}, {2: 1, 3: 1}, {4: 1}), ({1: 2}, {4: 1, 5: 3}, {2: 1}), ({2: 1}, {3: 1}, {5: 1})])
    (tmp_path / "in").write_bytes(M.r1cs_to_bytes(C, dense_b))
    assert run(DRIVER, "swap_ab", "alt-bn128", tmp_path / "in", tmp_path / "out").stdout.strip() == "swapped"
    want = M.swap_AB_if_beneficial(dense_b)
    assert want is not dense_b and (tmp_path / "out").read_bytes() == M.r1cs_to_bytes(C, want)
    # the swapped system is left alone (A now touches more), and so is a tie
    (tmp_path / "in2").write_bytes(M.r1cs_to_bytes(C, want))
    assert run(DRIVER, "swap_ab", "alt-bn128", tmp_path / "in2", tmp_path / "out2").stdout.strip() == "kept"
    assert (tmp_path / "out2").read_bytes() == M.r1cs_to_bytes(C, want)
    tie = R1CS(1, 2, [({1: 1}, {2: 1}, {3: 1})])
    (tmp_path / "in3").write_bytes(M.r1cs_to_bytes(C, tie))
    assert run(DRIVER, "swap_ab", "alt-bn128", tmp_path / "in3", tmp_path / "out3").stdout.strip() == "kept"


def test_identity_is_written_as_zero_one(tmp_path):
    """group_element_utils.tcc:93-113: the identity travels as (0, 1); a key whose B_query holds one survives."""
    C = BN254
    cs, pk, _ = simple_key(C)
    pk.A_query[2] = None
    pk.B_query[0] = (pk.B_query[0][0], None, None)
    pk_bytes = M.proving_key_to_bytes(C, pk)
    assert M.g1_to_bytes(C, None) == bytes(31) + b"\x00" + bytes(31) + b"\x01"
    (tmp_path / "pk").write_bytes(pk_bytes)
    (tmp_path / "asg").write_bytes(M.assignment_to_bytes(C, [1, 2, 3, 4]))
    run(DRIVER, "codec", "alt-bn128", tmp_path / "pk", tmp_path / "pk2", tmp_path / "asg", tmp_path / "asg2", 1)
    assert (tmp_path / "pk2").read_bytes() == pk_bytes


def test_error_convention(tmp_path):
    """std::invalid_argument for a key that is not well-formed (groth16_snark.tcc:145-147), std::length_error for
    an assignment shorter than the primary input (r1cs_variable_assignment_serialization.tcc:34-37), a stream
    failure for a truncated file (libtool/tool_util.cpp:10-17)."""
    C = BN254
    cs, pk, _ = simple_key(C)
    good = M.proving_key_to_bytes(C, pk)
    asg = M.assignment_to_bytes(C, [1, 2, 3, 4])
    (tmp_path / "asg").write_bytes(asg)
    args = lambda pkf, af, p=1: (DRIVER, "codec", "alt-bn128", tmp_path / pkf, tmp_path / "o1", tmp_path / af, tmp_path / "o2", p)  # noqa: E731
    # a point off the curve: flip the low byte of alpha_g1.Y
    bad = bytearray(good)
    bad[63] ^= 1
    (tmp_path / "bad").write_bytes(bytes(bad))
    assert "not well-formed" in run(*args("bad", "asg"), expect=10).stderr
    # truncated key
    (tmp_path / "short").write_bytes(good[: len(good) // 2])
    run(*args("short", "asg"), expect=12)
    # 4 values, 5 primary inputs requested
    (tmp_path / "pk").write_bytes(good)
    assert "smaller than primary input size" in run(*args("pk", "asg", 5), expect=11).stderr
    # a missing file
    run(*args("nonexistent", "asg"), expect=12)


def test_tool_command_line(tmp_path):
    """`prove [pk_file] [assignment_file] [proof_file]` with -c / -s / -p / --profile (prove_cmd.cpp:68-123,
    tool_common.hpp:44-67,103-116): argument errors are reported before any device is touched."""
    assert "prove [pk_file] [assignment_file] [proof_file]" in run(TOOL, "--help").stdout
    assert "pk_file not specified" in run(TOOL, "prove", expect=1).stderr
    assert "proof_file not specified" in run(TOOL, "prove", "a", "b", expect=1).stderr
    assert "unrecognized curve" in run(TOOL, "prove", "-c", "bw6-761", "a", "b", "c", expect=1).stderr
    assert "unrecognized snark" in run(TOOL, "prove", "-s", "plonk", "a", "b", "c", expect=1).stderr
    assert "pghr13" in run(TOOL, "prove", "--snark", "pghr13", "a", "b", "c", expect=1).stderr
    assert "cannot open" in run(TOOL, "prove", tmp_path / "nope", "b", "c", expect=1).stderr
    assert "unknown command" in run(TOOL, "verify", expect=1).stderr
