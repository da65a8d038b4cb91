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


class R1csMatrix(ctypes.Structure):
    _fields_ = [("row_ptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("val", ctypes.c_void_p)]


class PkView(ctypes.Structure):
    _fields_ = [
        ("num_inputs", ctypes.c_size_t), ("num_variables", ctypes.c_size_t), ("num_constraints", ctypes.c_size_t),
        ("a", R1csMatrix), ("b", R1csMatrix), ("c", R1csMatrix),
        ("alpha_g1", ctypes.c_void_p), ("beta_g1", ctypes.c_void_p), ("beta_g2", ctypes.c_void_p),
        ("delta_g1", ctypes.c_void_p), ("delta_g2", ctypes.c_void_p),
        ("A_query", ctypes.c_void_p), ("A_len", ctypes.c_size_t),
        ("B_indices", ctypes.c_void_p), ("B_g2", ctypes.c_void_p), ("B_g1", ctypes.c_void_p), ("B_len", ctypes.c_size_t),
        ("H_query", ctypes.c_void_p), ("H_len", ctypes.c_size_t),
        ("L_query", ctypes.c_void_p), ("L_len", ctypes.c_size_t),
        ("form", ctypes.c_int),
    ]


class ZethGpuError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__("zeth_gpu: %s (status %d)" % (what, status))
        self.status = status


def library_path() -> str:
    override = os.environ.get("ZETH_GPU_LIBRARY")  # developer knob: A/B a differently built library
    if override:
        return override
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
    L.zg_ctx_synchronize.argtypes = [vp]
    L.zg_ctx_set_profiling.argtypes = [vp, ci]
    L.zg_ctx_last_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), ci, ctypes.POINTER(ci)]
    L.zg_dev_malloc.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.zg_dev_free.argtypes = [vp, vp]
    L.zg_dev_upload.argtypes = [vp, vp, vp, sz]
    L.zg_dev_download.argtypes = [vp, vp, vp, sz]
    L.zg_scalars_random.argtypes = [vp, ci, vp, sz, ctypes.c_uint64]
    L.zg_bases_create.argtypes = [vp, ci, ci, vp, sz, ci, ci, ctypes.POINTER(vp)]
    L.zg_bases_generate.argtypes = [vp, ci, ci, sz, vp, vp, ctypes.POINTER(vp)]
    L.zg_bases_batch_exp.argtypes = [vp, ci, ci, vp, sz, ctypes.POINTER(vp)]
    L.zg_bases_count.argtypes = [vp, ctypes.POINTER(sz)]
    L.zg_bases_read.argtypes = [vp, sz, sz, ci, vp]
    L.zg_bases_precompute.argtypes = [vp, ci]
    L.zg_bases_table_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.zg_bases_free.argtypes = [vp]
    L.zg_bases_free.restype = None
    L.zg_msm.argtypes = [vp, sz, vp, sz, ci, ci, ci, vp]
    L.zg_msm_ex.argtypes = [vp, sz, vp, sz, ci, ci, ci, vp, ci]
    L.zg_ntt.argtypes = [vp, ci, vp, ctypes.c_uint, ci, ci, ci, ci]
    L.zg_msm_sharded.argtypes = [ctypes.POINTER(vp), sz, ctypes.POINTER(vp), ctypes.POINTER(sz), ci, ci, ci, vp]
    L.zg_runtime_info.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(ci)]
    L.zg_ntt_noalloc.argtypes = [vp, ci, vp, ctypes.c_uint, ci, ci, ci]
    L.zg_pk_create.argtypes = [vp, ci, ctypes.POINTER(PkView), ctypes.POINTER(vp)]
    L.zg_pk_free.argtypes = [vp]
    L.zg_pk_clone.argtypes = [vp, ctypes.POINTER(vp)]
    L.zg_pk_free.restype = None
    L.zg_pk_domain_size.argtypes = [vp, ctypes.POINTER(sz)]
    L.zg_prove.argtypes = [vp, vp, sz, vp, sz, ci, vp, vp, ci, vp]
    L.zg_witness_map.argtypes = [vp, vp, sz, vp, sz, ci, ci, vp]
    L.zg_host_group_op.argtypes = [ci, ci, ci, vp, vp, ci, vp]
    L.zg_host_group_sum.argtypes = [ci, ci, vp, sz, ci, vp]
    L.zg_host_points_check.argtypes = [ci, ci, vp, sz, ci, ctypes.POINTER(sz)]
    L.zg_group_fft.argtypes = [vp, ci, ci, vp, ctypes.c_uint, ci, ci]
    L.zg_group_scale.argtypes = [vp, ci, ci, vp, sz, vp, ci]
    L.zg_group_sparse_combine.argtypes = [vp, ci, ci, vp, sz, vp, vp, vp, sz, ci, vp]
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

    def synchronize(self):
        _check(self._L.zg_ctx_synchronize(self._h))

    def set_profiling(self, enabled: bool):
        _check(self._L.zg_ctx_set_profiling(self._h, int(enabled)))

    def last_timings(self) -> List[Tuple[str, float]]:
        names = (ctypes.c_char_p * 64)()
        ms = (ctypes.c_float * 64)()
        n = ctypes.c_int()
        _check(self._L.zg_ctx_last_timings(self._h, names, ms, 64, ctypes.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]


class DeviceBuffer:
    """Device memory owned by the library (zg_dev_malloc): keeps scalars resident in HBM."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx, self.nbytes = ctx, nbytes
        p = ctypes.c_void_p()
        _check(ctx._L.zg_dev_malloc(ctx._h, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_bytes(cls, ctx: Context, data: bytes) -> "DeviceBuffer":
        b = cls(ctx, len(data))
        b.upload(data)
        return b

    @classmethod
    def random_scalars(cls, ctx: Context, curve: int, n: int, seed: int) -> "DeviceBuffer":
        b = cls(ctx, 32 * n)
        _check(ctx._L.zg_scalars_random(ctx._h, curve, ctypes.c_void_p(b.ptr), n, seed))
        return b

    def upload(self, data: bytes, offset: int = 0):
        assert offset + len(data) <= self.nbytes
        _check(self.ctx._L.zg_dev_upload(self.ctx._h, ctypes.c_void_p(self.ptr + offset), data, len(data)))

    def download(self, nbytes: Optional[int] = None, offset: int = 0) -> bytes:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = ctypes.create_string_buffer(max(1, nbytes))
        _check(self.ctx._L.zg_dev_download(self.ctx._h, out, ctypes.c_void_p(self.ptr + offset), nbytes))
        return out.raw[:nbytes]

    def free(self):
        if self.ptr:
            self.ctx._L.zg_dev_free(self.ctx._h, ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


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

    @classmethod
    def batch_exp(cls, ctx, curve, group, scalars) -> "Bases":
        """P_i = s_i G (libff::batch_exp shape); scalars: ints or n x 32 B little-endian bytes."""
        buf = scalars if isinstance(scalars, (bytes, bytearray)) else encode_scalars(scalars)
        n = len(buf) // 32
        h = ctypes.c_void_p()
        _check(ctx._L.zg_bases_batch_exp(ctx._h, curve, group, bytes(buf), n, ctypes.byref(h)))
        return cls(ctx, curve, group, h, n)

    def precompute(self, window_bits: int = 0) -> "Bases":
        """Fixed-base table (zg_bases_precompute): the multiples 2^(j c) P_i stay in HBM and every later msm() sorts
        all digits into one bucket set.  Returns self."""
        _check(self._L.zg_bases_precompute(self._h, window_bits))
        return self

    def table_info(self) -> Tuple[int, int]:
        """(digit width c, levels D); (0, 0) without a table."""
        c, d = ctypes.c_int(0), ctypes.c_int(0)
        _check(self._L.zg_bases_table_info(self._h, ctypes.byref(c), ctypes.byref(d)))
        return c.value, d.value

    def read_bytes(self, first: int, count: int, form: int = FORM_PLAIN) -> bytes:
        ab = affine_bytes(self.curve, self.group)
        out = ctypes.create_string_buffer(max(1, count * ab))
        _check(self._L.zg_bases_read(self._h, first, count, form, out))
        return out.raw[: count * ab]

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
        if isinstance(scalars, DeviceBuffer):
            n = scalars.nbytes // 32 if n is None else n
            scalars = scalars.ptr
        if isinstance(scalars, int):
            assert n is not None
            _check(self._L.zg_msm_ex(self._h, first, ctypes.c_void_p(scalars), n, form, MEM_DEVICE, FORM_PLAIN, out, window_bits))
        else:
            buf = scalars if isinstance(scalars, (bytes, bytearray)) else encode_scalars(scalars)
            cnt = len(buf) // 32
            cbuf = ctypes.create_string_buffer(bytes(buf), len(buf)) if cnt else None
            _check(self._L.zg_msm_ex(self._h, first, cbuf, cnt, form, MEM_HOST, FORM_PLAIN, out, window_bits))
        return decode_point(self.curve, self.group, out.raw)

    def msm_bytes(self, scalars: "DeviceBuffer", n: int, first: int = 0, form: int = FORM_PLAIN) -> bytes:
        """zg_msm on device-resident scalars; the affine result as raw bytes (plain form): the fast path of the
        sharded benchmark, no integer decoding."""
        out = ctypes.create_string_buffer(affine_bytes(self.curve, self.group))
        _check(self._L.zg_msm(self._h, first, ctypes.c_void_p(scalars.ptr), n, form, MEM_DEVICE, FORM_PLAIN, out))
        return out.raw

    def free(self):
        if self._h:
            self._L.zg_bases_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def runtime_info() -> dict:
    """zg_runtime_info: which libamdhip64 / libhsa-runtime64 images serve this process."""
    import json

    buf = ctypes.create_string_buffer(8192)
    mixed = ctypes.c_int(0)
    _check(load_library().zg_runtime_info(buf, len(buf), ctypes.byref(mixed)))
    return json.loads(buf.value.decode())


def msm_sharded(shards: Sequence["Bases"], scalars: Sequence["DeviceBuffer"], counts: Sequence[int]):
    """zg_msm_sharded over resident-base shards (one context each) with device-resident scalar slices -> affine point."""
    n = len(shards)
    L = shards[0]._L
    hs = (ctypes.c_void_p * n)(*[b._h for b in shards])
    ps = (ctypes.c_void_p * n)(*[ctypes.c_void_p(s.ptr) for s in scalars])
    cs = (ctypes.c_size_t * n)(*counts)
    out = ctypes.create_string_buffer(affine_bytes(shards[0].curve, shards[0].group))
    _check(L.zg_msm_sharded(hs, n, ps, cs, FORM_PLAIN, MEM_DEVICE, FORM_PLAIN, out))
    return decode_point(shards[0].curve, shards[0].group, out.raw)


class EvaluationDomain:
    """Mirror of libfqfft::basic_radix2_domain<Fr>(m) as Zeth uses it (evaluator_from_lagrange.tcc:43,
    powersoftau_utils.tcc:849-851): FFT / iFFT / cosetFFT / icosetFFT in place, natural order."""

    def __init__(self, ctx: Context, curve: int, m: int):
        if m < 1 or m & (m - 1):
            raise ValueError("basic_radix2_domain: m must be a power of two")
        self.ctx, self.curve, self.m, self.log_m = ctx, curve, m, m.bit_length() - 1

    def _run(self, values, inverse: bool, coset: bool):
        if isinstance(values, DeviceBuffer):
            assert values.nbytes == 32 * self.m
            _check(self.ctx._L.zg_ntt(self.ctx._h, self.curve, ctypes.c_void_p(values.ptr), self.log_m, int(inverse),
                                      int(coset), FORM_PLAIN, MEM_DEVICE))
            return values
        assert len(values) == self.m
        buf = ctypes.create_string_buffer(encode_scalars(values), 32 * self.m)
        _check(self.ctx._L.zg_ntt(self.ctx._h, self.curve, buf, self.log_m, int(inverse), int(coset), FORM_PLAIN, MEM_HOST))
        raw = buf.raw
        return [le_to_int(raw[32 * i : 32 * i + 32]) for i in range(self.m)]

    def run_noalloc(self, values: "DeviceBuffer", inverse: bool, coset: bool) -> "DeviceBuffer":
        """The same transform through zg_ntt_noalloc (static device scratch, m <= 2^16, device memory only)."""
        assert values.nbytes == 32 * self.m
        _check(self.ctx._L.zg_ntt_noalloc(self.ctx._h, self.curve, ctypes.c_void_p(values.ptr), self.log_m, int(inverse),
                                          int(coset), FORM_PLAIN))
        return values

    def FFT(self, a):
        return self._run(a, False, False)

    def iFFT(self, a):
        return self._run(a, True, False)

    def cosetFFT(self, a):
        return self._run(a, False, True)

    def icosetFFT(self, a):
        return self._run(a, True, True)


def csr_from_rows(rows, r: int):
    """rows: list of {variable index: coefficient} (index 0 = constant one) -> (row_ptr, col, val) ctypes buffers."""
    import array

    rp = array.array("Q", [0])
    cols = array.array("I")
    vals = bytearray()
    for lc in rows:
        for idx, c in sorted(lc.items()):
            cols.append(idx)
            vals += int_to_le(c % r, 32)
        rp.append(len(cols))
    return ((ctypes.c_uint64 * len(rp))(*rp), (ctypes.c_uint32 * max(1, len(cols)))(*cols),
            ctypes.create_string_buffer(bytes(vals), max(1, len(vals))))


class ProvingKey:
    """Device-resident r1cs_gg_ppzksnark_proving_key (groth16_snark.hpp:18-21)."""

    def __init__(self, ctx: Context, curve: int, handle, keep):
        self.ctx, self.curve, self._h, self._keep = ctx, curve, handle, keep
        m = ctypes.c_size_t()
        _check(ctx._L.zg_pk_domain_size(handle, ctypes.byref(m)))
        self.domain_size = m.value

    @classmethod
    def from_model(cls, ctx: Context, curve: int, pk, r: int) -> "ProvingKey":
        """pk: an object with the fields of the oracle's ProvingKey (alpha_g1 ... L_query, cs)."""
        cs = pk.cs
        keep = []

        def buf(b: bytes):
            cb = ctypes.create_string_buffer(b, max(1, len(b)))
            keep.append(cb)
            return ctypes.cast(cb, ctypes.c_void_p)

        def g1(pts):
            return buf(b"".join(encode_point(curve, G1, P) for P in pts))

        def g2(pts):
            return buf(b"".join(encode_point(curve, G2, P) for P in pts))

        mats = []
        for k in range(3):
            rp, col, val = csr_from_rows([con[k] for con in cs.constraints], r)
            keep += [rp, col, val]
            mats.append(R1csMatrix(ctypes.cast(rp, ctypes.c_void_p), ctypes.cast(col, ctypes.c_void_p), ctypes.cast(val, ctypes.c_void_p)))
        bidx = (ctypes.c_uint32 * max(1, len(pk.B_query)))(*[e[0] for e in pk.B_query])
        keep.append(bidx)
        view = PkView(cs.num_inputs, cs.num_variables, cs.num_constraints, mats[0], mats[1], mats[2],
                      g1([pk.alpha_g1]), g1([pk.beta_g1]), g2([pk.beta_g2]), g1([pk.delta_g1]), g2([pk.delta_g2]),
                      g1(pk.A_query), len(pk.A_query),
                      ctypes.cast(bidx, ctypes.c_void_p), g2([e[1] for e in pk.B_query]), g1([e[2] for e in pk.B_query]), len(pk.B_query),
                      g1(pk.H_query), len(pk.H_query), g1(pk.L_query), len(pk.L_query), FORM_PLAIN)
        h = ctypes.c_void_p()
        _check(ctx._L.zg_pk_create(ctx._h, curve, ctypes.byref(view), ctypes.byref(h)))
        return cls(ctx, curve, h, None)

    def clone(self) -> "ProvingKey":
        """A second prover over the same resident key for another thread (zg_pk_clone); free it before `self`."""
        h = ctypes.c_void_p()
        _check(self.ctx._L.zg_pk_clone(self._h, ctypes.byref(h)))
        return ProvingKey(self.ctx, self.curve, h, self)

    def free(self):
        if self._h:
            self.ctx._L.zg_pk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Groth16Snark:
    """Mirror of libzeth::groth16_snark<ppT> for the proving path (groth16_snark.hpp:32-40)."""

    def __init__(self, ctx: Context, curve: int):
        self.ctx, self.curve = ctx, curve

    def generate_proof(self, pk: ProvingKey, primary_input: Sequence[int], auxiliary_input: Sequence[int],
                       r: Optional[int] = None, s: Optional[int] = None, order: Optional[int] = None):
        """groth16_snark::generate_proof(pk, primary, auxiliary) -> (g_A, g_B, g_C).  (r, s) default to OS randomness
        as libsnark's Fr::random_element() does; pass them for reproducible proofs."""
        if r is None or s is None:
            import secrets

            assert order is not None, "pass the group order to draw (r, s)"
            r, s = secrets.randbelow(order), secrets.randbelow(order)
        n = fq_bytes(self.curve)
        out = ctypes.create_string_buffer(8 * n)
        pb = ctypes.create_string_buffer(encode_scalars(primary_input), max(1, 32 * len(primary_input)))
        ab = ctypes.create_string_buffer(encode_scalars(auxiliary_input), max(1, 32 * len(auxiliary_input)))
        _check(self.ctx._L.zg_prove(pk._h, pb, len(primary_input), ab, len(auxiliary_input), FORM_PLAIN,
                                    int_to_le(r, 32), int_to_le(s, 32), FORM_PLAIN, out))
        raw = out.raw
        return (decode_point(self.curve, G1, raw[: 2 * n]), decode_point(self.curve, G2, raw[2 * n : 6 * n]),
                decode_point(self.curve, G1, raw[6 * n :]))

    def witness_map(self, pk: ProvingKey, primary_input: Sequence[int], auxiliary_input: Sequence[int]) -> List[int]:
        """libsnark::r1cs_to_qap_witness_map(...).coefficients_for_H (domain size + 1 values)."""
        m = pk.domain_size
        out = ctypes.create_string_buffer(32 * (m + 1))
        pb = ctypes.create_string_buffer(encode_scalars(primary_input), max(1, 32 * len(primary_input)))
        ab = ctypes.create_string_buffer(encode_scalars(auxiliary_input), max(1, 32 * len(auxiliary_input)))
        _check(self.ctx._L.zg_witness_map(pk._h, pb, len(primary_input), ab, len(auxiliary_input), FORM_PLAIN, FORM_PLAIN, out))
        raw = out.raw
        return [le_to_int(raw[32 * i : 32 * i + 32]) for i in range(m + 1)]


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


def _decode_points(curve: int, group: int, raw: bytes, n: int) -> list:
    ab = affine_bytes(curve, group)
    return [decode_point(curve, group, raw[i * ab : (i + 1) * ab]) for i in range(n)]


def group_fft(ctx: Context, curve: int, group: int, points: Sequence, inverse: bool = False) -> list:
    """Radix-2 transform over group elements (libfqfft::_basic_radix2_FFT<Fr, Gr>); inverse=True also scales by 1/n,
    i.e. compute_lagrange_from_powers (powersoftau_utils.tcc:125-134)."""
    n = len(points)
    if n < 1 or n & (n - 1):
        raise ValueError("group_fft: the number of points must be a power of two")
    buf = ctypes.create_string_buffer(b"".join(encode_point(curve, group, P) for P in points), n * affine_bytes(curve, group))
    _check(ctx._L.zg_group_fft(ctx._h, curve, group, buf, n.bit_length() - 1, int(inverse), FORM_PLAIN))
    return _decode_points(curve, group, buf.raw, n)


def group_scale(ctx: Context, curve: int, group: int, points: Sequence, k: int) -> list:
    """[k * P for P in points] (phase2.tcc:447-470)."""
    n = len(points)
    buf = ctypes.create_string_buffer(b"".join(encode_point(curve, group, P) for P in points), max(1, n * affine_bytes(curve, group)))
    _check(ctx._L.zg_group_scale(ctx._h, curve, group, buf, n, int_to_le(k, 32), FORM_PLAIN))
    return _decode_points(curve, group, buf.raw, n)


def group_sparse_combine(ctx: Context, curve: int, group: int, points: Sequence, rows: Sequence, r: int) -> list:
    """rows: list of {point index: coefficient}; -> [sum coeff * points[idx] for each row] (mpc_utils.tcc:187-242)."""
    rp, col, val = csr_from_rows(rows, r)
    n = len(points)
    pbuf = ctypes.create_string_buffer(b"".join(encode_point(curve, group, P) for P in points), max(1, n * affine_bytes(curve, group)))
    out = ctypes.create_string_buffer(max(1, len(rows) * affine_bytes(curve, group)))
    _check(ctx._L.zg_group_sparse_combine(ctx._h, curve, group, pbuf, n, rp, col, val, len(rows), FORM_PLAIN, out))
    return _decode_points(curve, group, out.raw, len(rows))


def host_group_sum(curve: int, group: int, raw: bytes, n: int) -> bytes:
    """Sum of n affine points given as raw plain-form bytes (zg_host_group_sum) -> raw bytes."""
    ab = affine_bytes(curve, group)
    assert len(raw) == n * ab
    out = ctypes.create_string_buffer(ab)
    _check(load_library().zg_host_group_sum(curve, group, raw, n, FORM_PLAIN, out))
    return out.raw
