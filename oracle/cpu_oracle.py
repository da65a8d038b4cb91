"""ctypes wrapper of oracle/liboracle.so (zeth_cpu.cpp) -- TEST INFRASTRUCTURE ONLY.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes
import os
from typing import List, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CsrView(ctypes.Structure):
    _fields_ = [("row_ptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("val", ctypes.c_void_p)]


class PkView(ctypes.Structure):
    _fields_ = [
        ("alpha_g1", ctypes.c_void_p), ("beta_g1", ctypes.c_void_p), ("beta_g2", ctypes.c_void_p),
        ("delta_g1", ctypes.c_void_p), ("delta_g2", ctypes.c_void_p),
        ("A_query", ctypes.c_void_p), ("A_len", ctypes.c_size_t),
        ("B_idx", ctypes.c_void_p), ("B_g2", ctypes.c_void_p), ("B_g1", ctypes.c_void_p), ("B_len", ctypes.c_size_t),
        ("H_query", ctypes.c_void_p), ("H_len", ctypes.c_size_t),
        ("L_query", ctypes.c_void_p), ("L_len", ctypes.c_size_t),
    ]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise ImportError("%s missing: run `make -C oracle`" % path)
        L = ctypes.CDLL(path)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        dp = ctypes.POINTER(ctypes.c_double)
        L.oz_msm.argtypes = [ci, ci, vp, vp, sz, ci, ci, vp, dp]
        L.oz_fft.argtypes = [ci, vp, sz, ci, ci, dp]
        L.oz_witness_map.argtypes = [ci, sz, sz, sz, ctypes.POINTER(CsrView), vp, vp, sz]
        L.oz_prove.argtypes = [ci, ctypes.POINTER(PkView), sz, sz, sz, ctypes.POINTER(CsrView), vp, vp, vp, ci, vp, dp]
        L.oz_set_threads.argtypes = [ci]
        _LIB = L
    return _LIB


def max_threads() -> int:
    return lib().oz_max_threads()


def _le(x: int, n: int) -> bytes:
    return int(x).to_bytes(n, "little")


def fq_bytes(curve: int) -> int:
    return 32 if curve == 0 else 48


def enc_point(curve: int, group: int, P) -> bytes:
    n = fq_bytes(curve)
    if P is None:
        return bytes(n * (2 if group == 1 else 4))
    if group == 1:
        return _le(P[0], n) + _le(P[1], n)
    (x0, x1), (y0, y1) = P
    return _le(x0, n) + _le(x1, n) + _le(y0, n) + _le(y1, n)


def dec_point(curve: int, group: int, b: bytes):
    n = fq_bytes(curve)
    v = [int.from_bytes(b[i * n:(i + 1) * n], "little") for i in range(len(b) // n)]
    if not any(v):
        return None
    return (v[0], v[1]) if group == 1 else ((v[0], v[1]), (v[2], v[3]))


def msm_bytes(curve: int, group: int, bases: bytes, scalars: bytes, n: int, method: int = 0, chunks: int = 1):
    """Returns (affine point, core seconds)."""
    ab = fq_bytes(curve) * (2 if group == 1 else 4)
    out = ctypes.create_string_buffer(ab)
    sec = ctypes.c_double()
    rc = lib().oz_msm(curve, group, bases, scalars, n, method, chunks, out, ctypes.byref(sec))
    assert rc == 0
    return dec_point(curve, group, out.raw), sec.value


def msm(curve: int, group: int, points: Sequence, scalars: Sequence[int], method: int = 0, chunks: int = 1):
    bases = b"".join(enc_point(curve, group, P) for P in points[: len(scalars)])
    sc = b"".join(_le(s, 32) for s in scalars)
    return msm_bytes(curve, group, bases, sc, len(scalars), method, chunks)[0]


def fft(curve: int, values: Sequence[int], inverse: bool, coset: bool) -> List[int]:
    n = len(values)
    buf = ctypes.create_string_buffer(b"".join(_le(v, 32) for v in values), 32 * n)
    rc = lib().oz_fft(curve, buf, n, int(inverse), int(coset), None)
    assert rc == 0
    raw = buf.raw
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(n)]


def fft_bytes(curve: int, data: bytearray, n: int, inverse: bool, coset: bool) -> float:
    """In place on a writable buffer of n x 32 B; returns core seconds."""
    sec = ctypes.c_double()
    cbuf = (ctypes.c_char * len(data)).from_buffer(data)
    rc = lib().oz_fft(curve, cbuf, n, int(inverse), int(coset), ctypes.byref(sec))
    assert rc == 0
    return sec.value


class Csr:
    """CSR encoding of one R1CS matrix (list of {var: coeff} rows)."""

    def __init__(self, rows, r: int):
        import array

        rp = array.array("Q", [0])
        cols = array.array("I")
        vals = bytearray()
        for lc in rows:
            for idx, c in sorted(lc.items()):
                cols.append(idx)
                vals += _le(c % r, 32)
            rp.append(len(cols))
        self.row_ptr = (ctypes.c_uint64 * len(rp))(*rp)
        self.col = (ctypes.c_uint32 * max(1, len(cols)))(*cols)
        self.val = ctypes.create_string_buffer(bytes(vals), max(1, len(vals)))
        self.nnz = len(cols)

    def view(self) -> CsrView:
        return CsrView(ctypes.cast(self.row_ptr, ctypes.c_void_p), ctypes.cast(self.col, ctypes.c_void_p),
                       ctypes.cast(self.val, ctypes.c_void_p))


def witness_map(curve: int, C, cs, primary, aux) -> List[int]:
    mats = [Csr([con[k] for con in cs.constraints], C.r) for k in range(3)]
    views = (CsrView * 3)(*[m.view() for m in mats])
    full = list(primary) + list(aux)
    fb = ctypes.create_string_buffer(b"".join(_le(v, 32) for v in full), max(1, 32 * len(full)))
    m = cs.domain_size()
    out = ctypes.create_string_buffer(32 * (m + 1))
    rc = lib().oz_witness_map(curve, cs.num_inputs, cs.num_variables, cs.num_constraints, views, fb, out, m)
    assert rc == 0
    raw = out.raw
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(m + 1)]


def prove(curve: int, C, pk, primary, aux, r_rand: int, s_rand: int, chunks: int = 1):
    cs = pk.cs
    mats = [Csr([con[k] for con in cs.constraints], C.r) for k in range(3)]
    views = (CsrView * 3)(*[m.view() for m in mats])
    full = list(primary) + list(aux)
    fb = ctypes.create_string_buffer(b"".join(_le(v, 32) for v in full), max(1, 32 * len(full)))
    keep = []

    def buf(b: bytes):
        cb = ctypes.create_string_buffer(b, max(1, len(b)))
        keep.append(cb)
        return ctypes.cast(cb, ctypes.c_void_p)

    g1 = lambda pts: buf(b"".join(enc_point(curve, 1, P) for P in pts))  # noqa: E731
    g2 = lambda pts: buf(b"".join(enc_point(curve, 2, P) for P in pts))  # noqa: E731
    bidx = (ctypes.c_uint32 * max(1, len(pk.B_query)))(*[e[0] for e in pk.B_query])
    view = PkView(g1([pk.alpha_g1]), g1([pk.beta_g1]), g2([pk.beta_g2]), g1([pk.delta_g1]), g2([pk.delta_g2]),
                  g1(pk.A_query), len(pk.A_query),
                  ctypes.cast(bidx, ctypes.c_void_p), g2([e[1] for e in pk.B_query]), g1([e[2] for e in pk.B_query]), len(pk.B_query),
                  g1(pk.H_query), len(pk.H_query), g1(pk.L_query), len(pk.L_query))
    n = fq_bytes(curve)
    out = ctypes.create_string_buffer(8 * n)
    sec = (ctypes.c_double * 3)()
    rc = lib().oz_prove(curve, ctypes.byref(view), cs.num_inputs, cs.num_variables, cs.num_constraints, views, fb,
                        _le(r_rand, 32), _le(s_rand, 32), chunks, out, sec)
    assert rc == 0
    raw = out.raw
    return dec_point(curve, 1, raw[: 2 * n]), dec_point(curve, 2, raw[2 * n: 6 * n]), dec_point(curve, 1, raw[6 * n:])
