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
