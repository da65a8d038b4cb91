"""The C-ABI library loads without a GPU, exports every symbol include/zeth_gpu.h declares, and refuses to run
without a device (no CPU fallback in the product path)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "zeth_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zg_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    import zeth_amd

    L = zeth_amd.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n


def test_no_device_is_a_loud_error_not_a_fallback():
    import zeth_amd

    L = zeth_amd.load_library()
    count = ctypes.c_int()
    hip = ctypes.CDLL("libamdhip64.so")
    has_gpu = hip.hipGetDeviceCount(ctypes.byref(count)) == 0 and count.value > 0
    if has_gpu:
        zeth_amd.Context(0).close()
        return
    h = ctypes.c_void_p()
    assert L.zg_ctx_create(0, ctypes.byref(h)) == -3
    assert b"no CPU fallback" in L.zg_strerror(-3)
    try:
        zeth_amd.Context(0)
    except zeth_amd.ZethGpuError as e:
        assert e.status == -3
    else:
        raise AssertionError("Context() must fail without a GPU")


def test_sizes():
    import zeth_amd

    L = zeth_amd.load_library()
    v = [ctypes.c_size_t() for _ in range(4)]
    assert L.zg_sizes(0, *[ctypes.byref(x) for x in v]) == 0
    assert [x.value for x in v] == [32, 32, 64, 128]
    assert L.zg_sizes(1, *[ctypes.byref(x) for x in v]) == 0
    assert [x.value for x in v] == [32, 48, 96, 192]
    assert L.zg_sizes(7, *[ctypes.byref(x) for x in v]) == -1
