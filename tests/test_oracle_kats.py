"""Pins oracle/zeth_model.py against every known-answer vector the reference's own tests hold for
this path (SURVEY.md section 8c, KAT-1 .. KAT-6).  The hex strings below are DATA copied from the
cited reference tests; nothing here reads /root/reference at run time."""
import os

import pytest

from zeth_model import (
    BLS12_377,
    BN254,
    Domain,
    VerificationKey,
    g1_to_json,
    g2_to_json,
    pairing_product_is_one,
    pot_load,
    verify,
)

H = lambda s: int(s, 16)  # noqa: E731


def test_kat1_bn254_13G1_and_G2_json():
    # libzeth/tests/core/group_element_utils_test.cpp:23-29,42-52
    C = BN254
    assert g1_to_json(C, C.G1.mul(13, C.G1.gen)) == (
        '["0x05e86f8cc8a7a4f10f56093465679f17f8b8c3fdb41469e408b529e030f52f3f",'
        '"0x2857bd14bbc09767bed8e913d3ccb42b2bc8738f715417dd6f020725d22bcd90"]'
    )
    assert g2_to_json(C, C.G2.gen) == (
        '[["0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2",'
        '"0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"],'
        '["0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b",'
        '"0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"]]'
    )


def test_kat2_bn254_seven_inverse():
    # libzeth/tests/mpc/groth16/powersoftau_test.cpp:268-269,302-305,322-328,378-383
    C = BN254
    inv7 = pow(7, -1, C.r)
    assert inv7 == H("06e9c21069503b73ac9dc0d0edede80d4ee2d80a5a8834a709b290cbfdb6db6e")
    P = C.G1.mul(inv7, C.G1.gen)
    assert P == (
        H("00e0ee2def593392a7a94e9bbbac1d4b104dbe5b6ec573eb04105efeaed342ca"),
        H("0fa5508a1ea35c78b4b7fc8eefa700c2ba6dba17b6747b89f7ddfdd7e38a39d8"),
    )
    # powersoftau Fq2 packing: 64-byte big-endian c1*q + c0
    pack = lambda a: a[1] * C.q + a[0]  # noqa: E731
    gx, gy = C.G2.gen
    assert pack(gx) == H(
        "04d4bf3239f77cee7b47c7245e9281b3e9c1182d6381a87bbf81f9f2a6254b73"
        "1df569cda95e060bee91ba69b3f2d103658a7aea6b10e5bdc761e5715e7ee4bb"
    )
    assert pack(gy) == H(
        "01b4c328f0cbdb4aada63b3d09100d792376b94d07a6004e46054eeec849e8de"
        "9835158a11d28483dd8db236ea49f3630edc9e41944e494c5aacfc36af3b66e7"
    )
    Q = C.G2.mul(inv7, C.G2.gen)
    assert pack(Q[0]) == H(
        "01e7e3104c91291acf280deb0a4cd382378e7a8ef1bd689acb043228e8b43124"
        "b85ceb754819742ae483c87d54b6e7ed757faf6554923b63de6785e5cc97c837"
    )
    assert pack(Q[1]) == H(
        "08f37b607f92d68a476f98faa3d239b199e291263f4e82266aa2c71c2e9b7e5f"
        "b14da6d95df90f1f022fad41d223e50698a91e5d7f74dce6a77d752b8ab81a64"
    )


def test_kat3_powersoftau_challenge_file(golden_dir):
    # testdata/powersoftau_challenge.4.bin, layout powersoftau_utils.tcc:590-622; relations :776-828.
    C = BN254
    data = open(os.path.join(golden_dir, "powersoftau_challenge.4.bin"), "rb").read()
    assert len(data) == 6352
    n = 16
    pot, off = pot_load(C, data, n)
    assert off == len(data)
    G1, G2 = C.G1, C.G2
    for P in pot["tau_g1"] + pot["alpha_tau_g1"] + pot["beta_tau_g1"]:
        assert P is not None and G1.is_on_curve(P)
    for P in pot["tau_g2"] + [pot["beta_g2"]]:
        assert P is not None and G2.is_on_curve(P)
    assert pot["tau_g1"][0] == G1.gen and pot["tau_g2"][0] == G2.gen
    # well-formedness: e(tau^{i+1} G1, G2) == e(tau^i G1, tau G2)   (spot-check i = 0, 1, 29)
    tau_g2 = pot["tau_g2"][1]
    for i in (0, 1, 29):
        assert pairing_product_is_one(C, [(pot["tau_g1"][i + 1], G2.gen), (G1.neg(pot["tau_g1"][i]), tau_g2)])
    # alpha / beta rows follow the same ratio; beta_g2 matches beta_tau_g1[0]
    assert pairing_product_is_one(C, [(pot["alpha_tau_g1"][1], G2.gen), (G1.neg(pot["alpha_tau_g1"][0]), tau_g2)])
    assert pairing_product_is_one(C, [(pot["beta_tau_g1"][0], G2.gen), (G1.neg(G1.gen), pot["beta_g2"])])
    # the challenge's tau is unknown, so this doubles as real data for the evaluator_from_lagrange identity
    # (evaluator_from_lagrange_test.cpp:44-65): sum_j L_j(tau) G1 == G1 since sum_j L_j == 1.
    dom = Domain(C, n)
    acc = None
    for j in range(n):
        coeffs = [0] * n
        coeffs[j] = 1
        dom.iFFT(coeffs)
        acc = G1.add(acc, G1.multi_exp(pot["tau_g1"], coeffs))
    assert acc == G1.gen


KAT4 = {
    "alt-bn128": dict(
        g1_m1=("0x01", "0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd45"),
        g1_8=(
            "0x08b1d51d23480c10f472f5e93b9cfea88238c121fe155af7043937882c306a63",
            "0x299836713dad3fa34e337aa412466015c366af8ec50b9d7bd05aa74642822021",
        ),
        g1_m8y="0x06cc1801a38460866a1ccb126f3af847d41abb02a3662d116bc5e4d095fadd26",
        # JSON order is [c1, c0]
        g2_8=(
            ("0x03589520df85791604b5a2b720a21139aabdb41949d47779484b0db588bfa699",
             "0x18afc7fd8df1c902383c213b6d989f0066b7eca1388be49721792278984d9a29"),
            ("0x2cc25982f4a3b75f57f8f3e966d75e6da8c51776bf0828c7ce3f10171793cd2a",
             "0x17623e9e90176bcdf8454daa96008240b12709ca5d79de805744cfd137609bec"),
        ),
        g2_m8y=("0x03a1f4efec8de8ca605751cd1aa9f9efeebc531aa969a1c56de17bffc0e9301d",
                "0x19020fd4511a345bc00af80beb80d61ce65a60c70af7ec0ce4dbbc45a11c615b"),
        g2_m1y=("0x275dc4a288d1afb3cbb1ac09187524c7db36395df7be3b99e673b13a075a65ec",
                "0x1d9befcd05a5323e6da4d435f3b617cdb3af83285c2df711ef39c01571827f9d"),
    ),
    "bls12-377": dict(
        g1_m1=(
            "0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef",
            "0x001cefdc52b4e1eba6d3b6633bf15a765ca326aa36b6c0b5b1db375b6a5124fa540d200dfb56a6e58785e1aaaa63715b",
        ),
        g1_8=(
            "0x018aff632c0048f5afb5c07fd197a44a127c829be3ff6170c6cebc1154bc72633b45de2ac855e0da30cebfa33672e7f3",
            "0x00efb82d5f13565755df1445db9ed4c4969f09bb31cc610f83eb1490be76ab3817808c52d98074f1a98ff896012a78ab",
        ),
        g1_m8y="0x00be8218b8b1ba93705bf17a910274768383d037cf28b27f9b084d9efb929cc7ff8ad0f1567f8b0edb78c769fed58756",
        g2_8=(
            ("0x019ece89be3c82b561d87e7898ddfc928d2e86fafa21febc222d8ff38a752fe95b25e19b25410205800385fb047ca1cf",
             "0x016251c985d96e59d07cc363290ec7146682da9bde87ed3d64ccad05403992cc9f3704c7f9a1df34116840d220edba27"),
            ("0x012c74d25dd0ca3264d6141945dc309b94be879fc67835ab3db6b1faad807bca3e21e122184678fe4ea38affe654e15f",
             "0x002f302a2d6f3c2a5b53edb53d73c1e3f1771c3b929563c62e74e3330adb0f75b1b164fef915d00aa35b5e668524ef2b"),
        ),
        g2_m8y=("0x0081c573b9f446b86164f1a726c5189f856452533a7cdde3e13cb0350c88cc35d8e97c2217b987023665350019ab1ea2",
                "0x017f0a1bea55d4c06ae7180b2f2d875728abbdb76e5fafc8f07e7efcaf2e388a6559f84536ea2ff5e1ad61997adb10d6"),
        g2_m1y=("0x002933c9ab1da3519734bb7d40c74f7c96f7cd46d372c68a05fbe4f5d29d09ebac0fdae50f6dde73818058280cc85ff1",
                "0x00955cf57c9676d751f0b5431b46efdd5ea49e8f219e8b28d7bbd2176e29c49caa767e18b569d34dd5c5880f834819df"),
    ),
}


@pytest.mark.parametrize("C", [BN254, BLS12_377], ids=lambda c: c.name)
def test_kat4_generator_multiples(C):
    # client/tests/test_pairing.py:11-55,97-141
    k = KAT4[C.name]
    G1, G2 = C.G1, C.G2
    assert G1.mul(C.r - 1, G1.gen) == (H(k["g1_m1"][0]), H(k["g1_m1"][1]))
    p8 = G1.mul(8, G1.gen)
    assert p8 == (H(k["g1_8"][0]), H(k["g1_8"][1]))
    assert G1.mul(C.r - 8, G1.gen) == (p8[0], H(k["g1_m8y"]))
    q8 = G2.mul(8, G2.gen)
    (x1, x0), (y1, y0) = k["g2_8"]
    assert q8 == ((H(x0), H(x1)), (H(y0), H(y1)))
    assert G2.mul(C.r - 8, G2.gen) == (q8[0], (H(k["g2_m8y"][1]), H(k["g2_m8y"][0])))
    assert G2.neg(G2.gen) == (G2.gen[0], (H(k["g2_m1y"][1]), H(k["g2_m1y"][0])))
    # 8 G by repeated addition too (exercises add/dbl, not only the ladder)
    acc = None
    for _ in range(8):
        acc = G1.add(acc, G1.gen)
    assert acc == p8


def _evm_g1(words):
    return (H(words[0] + words[1]), H(words[2] + words[3]))


def _evm_g2(words):
    # binary Fq2 order is c1 then c0 (KAT-5 vs test_pairing.py:131-135)
    v = [H(words[2 * i] + words[2 * i + 1]) for i in range(4)]
    return ((v[1], v[0]), (v[3], v[2]))


def test_kat5_bls12_377_ec_operation_data():
    # client/test_contracts/test_bls12_377_contract.py:21-75,95-121
    C = BLS12_377
    G1, G2 = C.G1, C.G2
    assert C.r - 2 == H("12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a117fffffffffff")
    g1 = {
        2: ["0000000000000000000000000000000000ed453141939e91056edb5a4b5452ed",
            "7e61f7f3dd2a4b7ee90e97c9a2301955880661656781dc90857aed6d6a416390",
            "0000000000000000000000000000000000cfb0b9717bc8e5ae04601813171337",
            "ad99cdae42c561cae80b12f135c64479d6a23f5675ed5ca7e2dd5e8727d7c7ed"],
        3: ["0000000000000000000000000000000001252b781171f507db36291b433a1f91",
            "1a46543890a20ca9712e11f66a5d216e63d817bd8d96cef715abc604dcf6ec2e",
            "00000000000000000000000000000000014a00fa77c727e8987cc438b51bbe01",
            "2c823a19955ae692c54ce572a61f0ea1fe5cd981533df419fd1330d1f6e6d802"],
        4: ["0000000000000000000000000000000000c1090c0750892fa132043a3ec67b5d",
            "fb5aa24b4772bef1724153a759353c27a216259fecad4b19be1f169b2739bd26",
            "00000000000000000000000000000000003f820683c042b943bfb056956d1c8d",
            "bc16939f19bd1131ab15caced9f05b7fd009962b3d13907fde8175d86f72e3a4"],
        6: ["00000000000000000000000000000000010c65c0fb9e6c6ef4cbb27fdc55a07e",
            "474df11c564bd91e3fa162c32b7fc3dabba5fc508cfdd8938fb4a30f7de5ad9c",
            "000000000000000000000000000000000149a58ced619866b242313876fe2df3",
            "188f33b77566a9ddc966ff4d4d5c42d515be862c348f51cc91f1c45a74110ba6"],
        -8: ["00000000000000000000000000000000018aff632c0048f5afb5c07fd197a44a",
             "127c829be3ff6170c6cebc1154bc72633b45de2ac855e0da30cebfa33672e7f3",
             "0000000000000000000000000000000000be8218b8b1ba93705bf17a91027476",
             "8383d037cf28b27f9b084d9efb929cc7ff8ad0f1567f8b0edb78c769fed58756"],
    }
    for k, words in g1.items():
        assert G1.mul(k % C.r, G1.gen) == _evm_g1(words), k
    assert G1.add(_evm_g1(g1[2]), _evm_g1(g1[4])) == _evm_g1(g1[6])  # test_bls12_ecadd
    assert G1.mul(C.r - 2, _evm_g1(g1[4])) == _evm_g1(g1[-8])  # test_bls12_ecmul
    g2_4 = ["00000000000000000000000000000000008deeab95264cf16b9e9b87ce615b9c",
            "8ccff86332b286332c00d4f0f0bb4c9c0fc0923d83231e5e3a3a961c8e266651",
            "0000000000000000000000000000000000a54cee57ec21b53167f636a9bb1d81",
            "691edf259b4164867760b250049bbd722021968ab0b927ee5b0d44c70db106f5",
            "000000000000000000000000000000000113d2333add9290fe17235e413f2d8c",
            "7f448406f586aefaad39284ac6d80fb7f1741fb37f91c5c6299038063a34bb25",
            "000000000000000000000000000000000045e1d96afde76075237a8c3472473f",
            "1c4debf2bf79abf2c8b22fd4ccc5c3da30ee409c293094115bddd5da7de0da8a"]
    g2_8 = ["00000000000000000000000000000000019ece89be3c82b561d87e7898ddfc92",
            "8d2e86fafa21febc222d8ff38a752fe95b25e19b25410205800385fb047ca1cf",
            "00000000000000000000000000000000016251c985d96e59d07cc363290ec714",
            "6682da9bde87ed3d64ccad05403992cc9f3704c7f9a1df34116840d220edba27",
            "00000000000000000000000000000000012c74d25dd0ca3264d6141945dc309b",
            "94be879fc67835ab3db6b1faad807bca3e21e122184678fe4ea38affe654e15f",
            "00000000000000000000000000000000002f302a2d6f3c2a5b53edb53d73c1e3",
            "f1771c3b929563c62e74e3330adb0f75b1b164fef915d00aa35b5e668524ef2b"]
    Q4, Q8 = _evm_g2(g2_4), _evm_g2(g2_8)
    assert G2.mul(4, G2.gen) == Q4 and G2.mul(8, G2.gen) == Q8
    # test_bls12_ecpairing: e([6],[4]) e([3],[8]) e([4],[4]) e([-8],[8]) == 1, and != 1 with the last G2 swapped
    P = {k: _evm_g1(w) for k, w in g1.items()}
    assert pairing_product_is_one(C, [(P[6], Q4), (P[3], Q8), (P[4], Q4), (P[-8], Q8)])
    assert not pairing_product_is_one(C, [(P[6], Q4), (P[3], Q8), (P[4], Q4), (P[-8], Q4)])


def test_kat6_groth16_bls12_377_instance():
    # client/test_contracts/test_groth16_bls12_377_contract.py:15-63,98-104 (JSON Fq2 order [c1, c0])
    C = BLS12_377
    g2 = lambda x1, x0, y1, y0: ((H(x0), H(x1)), (H(y0), H(y1)))  # noqa: E731
    vk = VerificationKey()
    vk.alpha_g1 = (
        H("0x002a586f4738bdc629cfeffae788183586416cf2255ca2dfdca0866fa39a0ff27a75ab2fc8230815a3263da828707bb6"),
        H("0x009797441005cca3c8e1b0ec78e7d5c5da95091ae0d9ca7396fb2245ab99101b34d7c5f0175aafd35ede91dc09f9532d"),
    )
    vk.beta_g2 = g2(
        "0x003d46509bd7f4e68ce267343ee8e555570ef1ab465bfe36756ba76f8059f2714573e6a2833e15a03c5c078c80ed06d5",
        "0x010f8805ac1719b1c76cb7e12f0d548a53f30a0caad7f6f9b0eb1ed676aae1faa6476daa60f60916bfdd1fa63f389728",
        "0x003a0ac1e41533af971fb2faca7e2cf662142c8b329b01efd299bc5ec263617127d5599f36b2038c58dfee7cc009b0ec",
        "0x01595cb8a7e1401cd60c4b132ac611affec72d56c20624c1237fa318fb9c007dee7f92c08db7d910022b0ef1c7f95ff4",
    )
    vk.delta_g2 = g2(
        "0x001e2716cb3ffbfe3cc1e76c6fa548018b3f9b12bfc2aa07640c9071fb7a95b4bdaf0ee15a0a4f6ddddc85834efc9009",
        "0x0098df94239036766b474b3a800d6edbc561e0cc37b251fa94f7cbcaeb01388646016163cd3d6a784a5c63f1879f0cf6",
        "0x008fd1d23577ded8f806c31e5d34b863b534382c83030c4382025cf3a06bdc14534f79861e9d8c17c98c3ad75d4bd46c",
        "0x003b5aeb59778b91048df1827f71209950b0e397cb9430632e2f194d37f51318bff10a37d88c2c86f2c21f71126bc909",
    )
    vk.ABC_g1 = [
        (H("0x01768261e5d9e312768c75fde3c50cd800a3ab94572c55f3482993bc406ea7cac091c0a450c4a2f1a5ad7aa9e3f6eb10"),
         H("0x012b1728cf8b05ebfb588d307f432deebf428c202a8572cec70ab403559b2999abfdbe830ffe51558876936facfc967b")),
        (H("0x00ecfad9e9ebc8a46b3a5664abdcea154d30565f04c90e75e8bdf157a019ddda540d58621a6543745b62e6d856b5a962"),
         H("0x0042421dd9c1f425afd48f2b1814712cc8a5238b96b3db9edc24f80e25e370351c80aa0cd2c79bdf1fc4a4319fea6209")),
    ]
    proof = (
        (H("0x0122d4c743e0b0a9a1bc920d070c536390096eedc60ac427943f01acf8f95011d0d9ee51539c338f5f31b74e9138646c"),
         H("0x0117f63edf1996607168c711e5d6c9add5dbad4df22c344bcb8f5a20026148a2cf1a3b905388ab952e05903944ed1f47")),
        g2(
            "0x019df27cf0dae60f1a6107cd36838b2379af2da7765e9a0257f424db9d58224f45b133d7085bf350885afe7ffcae16b9",
            "0x019cc9fec349ada5065075e7d42355f9bdc413d2b481f62b71ad9cf7af560f555714291ff3ed277ac973d1a83b4fc193",
            "0x009db56d5960ce13e15e93cb3985eb94d8b4ca033917757bf362baccac5462668cd900800f4de4861a5a92a542b81aef",
            "0x019e188316b82cc8a95ae6e7fb27ea5c6abaed28610152d421fc6fa090e406299cbb40e5f25c18339a5bc04c23bd6822",
        ),
        (H("0x00ae39a12b92b09e1ec1e256f7bf0522c067806f1ed46e959f018ec859b3a200506c7a7993243df77cc911ff35c1a0c8"),
         H("0x0016b89bd7d30c7a34d25780ef14fd027e2515c7449afe9d7dbf21f183673aefffb196e5959fcedd0dcf6064f1186d4c")),
    )
    assert verify(C, vk, [7], proof)
    assert not verify(C, vk, [8], proof)
