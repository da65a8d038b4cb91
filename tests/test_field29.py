"""The reduced-radix field arithmetic of the accumulation and transform kernels (zeth_amd/csrc/field29.cuh, accum29.cuh),
checked WITHOUT a GPU: the library's host build of the same templates against its 32-bit-limb arithmetic (which
tests/test_host_tail.py and the KATs tie to the oracle), through zg_selftest_reduced_radix.  Covers unpack/pack, lazy
add/sub/neg/carry, products, squares, the fused two-product reduction, the stored-form shortcuts of the transforms, and
the mixed addition with every exceptional case, in each field that uses them (libff Fp_model arithmetic: SURVEY 8a7)."""
import ctypes

import pytest

import zeth_amd.lib as zl

FIELDS = {0: "alt_bn128 Fq + G1 mixed addition", 1: "BLS12-377 Fq + G1 mixed addition", 2: "alt_bn128 Fr", 3: "BLS12-377 Fr",
          4: "alt_bn128 Fq2 + G2 mixed addition", 5: "BLS12-377 Fq2 + G2 mixed addition"}


@pytest.mark.parametrize("field", sorted(FIELDS))
def test_reduced_radix_arithmetic_matches_32_bit_limbs(field):
    L = zl.load_library()
    L.zg_selftest_reduced_radix.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int)]
    L.zg_selftest_reduced_radix.restype = ctypes.c_int
    bad = ctypes.c_int(-1)
    for seed in (1, 0x9E3779B97F4A7C15):
        assert L.zg_selftest_reduced_radix(field, 4000, seed, ctypes.byref(bad)) == 0
        assert bad.value == 0, FIELDS[field]


def test_selftest_rejects_unknown_field():
    L = zl.load_library()
    L.zg_selftest_reduced_radix.restype = ctypes.c_int
    bad = ctypes.c_int(0)
    assert L.zg_selftest_reduced_radix(ctypes.c_int(9), ctypes.c_int(1), ctypes.c_uint64(1), ctypes.byref(bad)) == -1


QUAD_GROUPS = {0: "alt_bn128 G1", 1: "BLS12-377 G1", 2: "alt_bn128 G2", 3: "BLS12-377 G2"}


@pytest.mark.parametrize("group", sorted(QUAD_GROUPS))
def test_quad_lane_point_arithmetic_matches_one_lane_formulas(group):
    """curve_quad.cuh / curve_quad29.cuh (the additions of k_combine_* and k_reduce*) on the host, four threads standing
    in for the four lanes of a quad: distinct points, P + P, P - P, the identity on either side, running sums, doublings
    and the round trip through the kernels' buffer form, against curve.cuh's xyzz_add / xyzz_dbl."""
    L = zl.load_library()
    L.zg_selftest_quad_lanes.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.zg_selftest_quad_lanes.restype = ctypes.c_int
    bad = ctypes.c_int(-1)
    assert L.zg_selftest_quad_lanes(group, ctypes.byref(bad)) == 0
    assert bad.value == 0, QUAD_GROUPS[group]
    assert L.zg_selftest_quad_lanes(9, ctypes.byref(bad)) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("group", [0, 1, 2, 3])
def test_quad_lane_point_arithmetic_on_the_device(group):
    """The same templates as compiled for gfx950 (G1: 32-bit limbs and the reduced radix; G2: the Fq2 instantiations on
    32-bit limbs): 68 additions and doublings with every exceptional case, compared on the host with the one-lane
    formulas -- every instantiation that ships is checked on the device it ships for."""
    L = zl.load_library()
    L.zg_selftest_quad_lanes_device.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.zg_selftest_quad_lanes_device.restype = ctypes.c_int
    bad = ctypes.c_int(-1)
    assert L.zg_selftest_quad_lanes_device(group, ctypes.byref(bad)) == 0
    assert bad.value == 0, QUAD_GROUPS[group]
