"""The library's host tail (XYZZ group law on 32-bit-limb Montgomery fields, compiled from the same source as the
device code) against the big-integer model, including the exceptional cases."""
import random

import pytest

import zeth_amd
from zeth_model import BLS12_377, BN254


@pytest.mark.parametrize("cid,C", [(0, BN254), (1, BLS12_377)], ids=["alt-bn128", "bls12-377"])
@pytest.mark.parametrize("gid", [1, 2])
def test_group_ops(cid, C, gid):
    G = C.G1 if gid == 1 else C.G2
    rng = random.Random(5 + cid + gid)
    for _ in range(3):
        k1, k2 = rng.randrange(C.r), rng.randrange(C.r)
        P, Q = G.mul(k1, G.gen), G.mul(k2, G.gen)
        assert zeth_amd.host_group_op(cid, gid, "mul", G.gen, k1) == P
        assert zeth_amd.host_group_op(cid, gid, "add", P, Q) == G.add(P, Q)
        assert zeth_amd.host_group_op(cid, gid, "add", P, P) == G.dbl(P)
        assert zeth_amd.host_group_op(cid, gid, "add", P, G.neg(P)) is None
        assert zeth_amd.host_group_op(cid, gid, "add", None, Q) == Q
        assert zeth_amd.host_group_op(cid, gid, "add", P, None) == P
        assert zeth_amd.host_group_op(cid, gid, "mul", P, 0) is None
        assert zeth_amd.host_group_op(cid, gid, "mul", P, C.r - 1) == G.neg(P)


def test_known_answers_from_reference_tests():
    """KAT-1 (libzeth/tests/core/group_element_utils_test.cpp:23-29): Fr(13) * G1::one() on alt_bn128;
    KAT-2 (libzeth/tests/mpc/groth16/powersoftau_test.cpp:268-269, 302-305): 7^-1 * G1."""
    C = BN254
    got = zeth_amd.host_group_op(0, 1, "mul", C.G1.gen, 13)
    assert got == (0x05E86F8CC8A7A4F10F56093465679F17F8B8C3FDB41469E408B529E030F52F3F,
                   0x2857BD14BBC09767BED8E913D3CCB42B2BC8738F715417DD6F020725D22BCD90)
    inv7 = pow(7, -1, C.r)
    assert zeth_amd.host_group_op(0, 1, "mul", zeth_amd.host_group_op(0, 1, "mul", C.G1.gen, inv7), 7) == C.G1.gen
    assert zeth_amd.host_group_op(0, 2, "mul", zeth_amd.host_group_op(0, 2, "mul", C.G2.gen, inv7), 7) == C.G2.gen
