"""ctypes binding of include/zeth_gpu.h (the same C ABI a libzeth/gpu C++ adapter links against)."""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

ALT_BN128, BLS12_377 = 0, 1
G1, G2 = 1, 2
FORM_PLAIN, FORM_MONT = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1

_FQ_LIMBS = {ALT_BN128: 8, BLS12_377: 12}
_LIB = None


class ZethGpuError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__("zeth_gpu: %s (status %d)" % (what, status))
        self.status = status


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzeth_gpu.so")


def load_library():
    """Load zeth_amd/libzeth_gpu.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `make` (or __graft_entry__.build()) first" % path)
    L = ctypes.CDLL(path)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.zg_strerror.restype = ctypes.c_char_p
    L.zg_strerror.argtypes = [ci]
    L.zg_sizes.argtypes = [ci] + [ctypes.POINTER(sz)] * 4
    L.zg_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.zg_ctx_destroy.argtypes = [vp]
    L.zg_ctx_destroy.restype = None
    L.zg_ctx_set_profiling.argtypes = [vp, ci]
    L.zg_ctx_last_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), ci, ctypes.POINTER(ci)]
    L.zg_bases_create.argtypes = [vp, ci, ci, vp, sz, ci, ci, ctypes.POINTER(vp)]
    L.zg_bases_generate.argtypes = [vp, ci, ci, sz, vp, vp, ctypes.POINTER(vp)]
    L.zg_bases_count.argtypes = [vp, ctypes.POINTER(sz)]
    L.zg_bases_read.argtypes = [vp, sz, sz, ci, vp]
    L.zg_bases_free.argtypes = [vp]
    L.zg_bases_free.restype = None
    L.zg_msm.argtypes = [vp, sz, vp, sz, ci, ci, ci, vp]
    L.zg_msm_ex.argtypes = [vp, sz, vp, sz, ci, ci, ci, vp, ci]
    L.zg_host_group_op.argtypes = [ci, ci, ci, vp, vp, ci, vp]
    _LIB = L
    return L


def _check(status: int):
    if status != 0:
        raise ZethGpuError(status, load_library().zg_strerror(status).decode())


# ---- integer <-> little-endian limb bytes ----
def int_to_le(x: int, nbytes: int) -> bytes:
    return int(x).to_bytes(nbytes, "little")


def le_to_int(b: bytes) -> int:
    return int.from_bytes(b, "little")


def fq_bytes(curve: int) -> int:
    return 4 * _FQ_LIMBS[curve]


def affine_bytes(curve: int, group: int) -> int:
    return fq_bytes(curve) * (2 if group == G1 else 4)


def encode_point(curve: int, group: int, P) -> bytes:
    """Affine point of the oracle's shape ((x, y) ints, G2 coordinates (c0, c1) tuples, None = identity)."""
    n = fq_bytes(curve)
    if P is None:
        return bytes(affine_bytes(curve, group))
    if group == G1:
        return int_to_le(P[0], n) + int_to_le(P[1], n)
    (x0, x1), (y0, y1) = P
    return int_to_le(x0, n) + int_to_le(x1, n) + int_to_le(y0, n) + int_to_le(y1, n)


def decode_point(curve: int, group: int, b: bytes):
    n = fq_bytes(curve)
    v = [le_to_int(b[i * n : (i + 1) * n]) for i in range(len(b) // n)]
    if not any(v):
        return None
    if group == G1:
        return (v[0], v[1])
    return ((v[0], v[1]), (v[2], v[3]))


def encode_scalars(scalars: Sequence[int]) -> bytes:
    return b"".join(int_to_le(s, 32) for s in scalars)


class Context:
    """One HIP device: stream + workspace (zg_ctx)."""

    def __init__(self, device: int = 0):
        self._L = load_library()
        h = ctypes.c_void_p()
        _check(self._L.zg_ctx_create(device, ctypes.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            self._L.zg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_profiling(self, enabled: bool):
        _check(self._L.zg_ctx_set_profiling(self._h, int(enabled)))

    def last_timings(self) -> List[Tuple[str, float]]:
        names = (ctypes.c_char_p * 64)()
        ms = (ctypes.c_float * 64)()
        n = ctypes.c_int()
        _check(self._L.zg_ctx_last_timings(self._h, names, ms, 64, ctypes.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]


class Bases:
    """Resident base points (the `gs` of libzeth::multi_exp, libzeth/core/multi_exp.hpp:13-22)."""

    def __init__(self, ctx: Context, curve: int, group: int, handle, n: int):
        self.ctx, self.curve, self.group, self._h, self.n = ctx, curve, group, handle, n
        self._L = ctx._L

    @classmethod
    def from_points(cls, ctx: Context, curve: int, group: int, points: Sequence) -> "Bases":
        buf = b"".join(encode_point(curve, group, P) for P in points)
        return cls.from_bytes(ctx, curve, group, buf, len(points), FORM_PLAIN)

    @classmethod
    def from_bytes(cls, ctx, curve, group, buf: bytes, n: int, form: int = FORM_PLAIN) -> "Bases":
        assert len(buf) == n * affine_bytes(curve, group)
        h = ctypes.c_void_p()
        cbuf = ctypes.create_string_buffer(buf, len(buf)) if n else None
        _check(ctx._L.zg_bases_create(ctx._h, curve, group, cbuf, n, form, MEM_HOST, ctypes.byref(h)))
        return cls(ctx, curve, group, h, n)

    @classmethod
    def generate(cls, ctx, curve, group, n: int, t0: int, d: int) -> "Bases":
        """P_i = (t0 + i d) G."""
        h = ctypes.c_void_p()
        _check(ctx._L.zg_bases_generate(ctx._h, curve, group, n, int_to_le(t0, 32), int_to_le(d, 32), ctypes.byref(h)))
        return cls(ctx, curve, group, h, n)

    def read(self, first: int, count: int) -> list:
        ab = affine_bytes(self.curve, self.group)
        out = ctypes.create_string_buffer(max(1, count * ab))
        _check(self._L.zg_bases_read(self._h, first, count, FORM_PLAIN, out))
        raw = out.raw
        return [decode_point(self.curve, self.group, raw[i * ab : (i + 1) * ab]) for i in range(count)]

    def msm(self, scalars, first: int = 0, window_bits: int = 0, n: Optional[int] = None, form: int = FORM_PLAIN):
        """sum_i scalars[i] * bases[first + i] -> affine point (None = identity).
        `scalars`: sequence of ints, bytes (n x 32 B little-endian), or a device pointer int with n given."""
        ab = affine_bytes(self.curve, self.group)
        out = ctypes.create_string_buffer(ab)
        if isinstance(scalars, int):
            assert n is not None
            _check(self._L.zg_msm_ex(self._h, first, ctypes.c_void_p(scalars), n, form, MEM_DEVICE, FORM_PLAIN, out, window_bits))
        else:
            buf = scalars if isinstance(scalars, (bytes, bytearray)) else encode_scalars(scalars)
            cnt = len(buf) // 32
            cbuf = ctypes.create_string_buffer(bytes(buf), len(buf)) if cnt else None
            _check(self._L.zg_msm_ex(self._h, first, cbuf, cnt, form, MEM_HOST, FORM_PLAIN, out, window_bits))
        return decode_point(self.curve, self.group, out.raw)

    def free(self):
        if self._h:
            self._L.zg_bases_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def host_group_op(curve: int, group: int, op: str, a, b):
    """Host-tail group law: op 'add' (a + b) or 'mul' (b * a with b an int)."""
    L = load_library()
    ab = affine_bytes(curve, group)
    out = ctypes.create_string_buffer(ab)
    abuf = ctypes.create_string_buffer(encode_point(curve, group, a), ab)
    if op == "add":
        bbuf = ctypes.create_string_buffer(encode_point(curve, group, b), ab)
        code = 0
    else:
        bbuf = ctypes.create_string_buffer(int_to_le(b, 32), 32)
        code = 1
    _check(L.zg_host_group_op(curve, group, code, abuf, bbuf, FORM_PLAIN, out))
    return decode_point(curve, group, out.raw)
