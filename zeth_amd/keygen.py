"""Key generation from explicit secrets, at scale, on the device.

Host-side mirror of groth16_snark<ppT>::generate_setup (libzeth/snarks/groth16/groth16_snark.tcc:18-25) in the
deterministic form the reference's tests use, r1cs_gg_ppzksnark_generator_from_secrets(cs, tau, alpha, beta, delta,
G1::one(), G2::one(), true) (libzeth/tests/mpc/groth16/mpc_test.cpp:344-365); the key structure is the one those
tests assert (mpc_test.cpp:282-331):
    A_query[j] = [A_j(tau)]_1 (j <= num_variables, with the input-consistency terms), B_query[j] = [B_j(tau)]_{2,1}
    (zero entries skipped), H_query[i] = [tau^i Z(tau) / delta]_1 (i < m - 1),
    L_query[j] = [(beta A_j + alpha B_j + C_j)(tau) / delta]_1 (j > num_inputs), vk.ABC_g1 likewise without / delta.
The scalar side (QAP evaluation at tau) is plain integer arithmetic here; the group side is one fixed-base batch
exponentiation per query on the GPU (zg_bases_batch_exp).  It exists to fabricate REAL keys for parity tests and
benchmarks at JoinSplit scale -- SURVEY.md 8f-4 lists a full GPU generate_setup as a later row."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import lib as _lib


@dataclass
class CurveInfo:
    """The numeric constants the host side needs (values are pinned by tests against the oracle's)."""

    curve: int
    r: int
    q: int
    fr_gen: int
    fr_s: int
    fr_root: int
    g1: tuple
    g2: tuple


ALT_BN128_INFO = CurveInfo(
    curve=_lib.ALT_BN128,
    r=0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    q=0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
    fr_gen=5, fr_s=28,
    fr_root=19103219067921713944291392827692070036145651957329286315305642004821462161904,
    g1=(1, 2),
    g2=((0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
         0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
        (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
         0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B)),
)
BLS12_377_INFO = CurveInfo(
    curve=_lib.BLS12_377,
    r=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    q=0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
    fr_gen=22, fr_s=47,
    fr_root=8065159656716812877374967518403273466521432693661810619979959746626482506078,
    g1=(0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
        0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6),
    g2=((0x00B997FEF930828FE1B9E6A1707B8AA508A3DBFD7FE2246499C709226A0A6FEF49F85B3A375363F4F8F6EA3FBD159F8A,
         0x00D6AC33B84947D9845F81A57A136BFA326E915FABC8CD6A57FF133B42D00F62E4E1AF460228CD5184DEAE976FA62596),
        (0x0118DD509B2E9A13744A507D515A595DBB7E3B63DF568866473790184BDF83636C94DF2B7A962CB2AF4337F07CB7E622,
         0x0185067C6CA76D992F064A432BD9F9BE832B0CAC2D824D0518F77D39E76C3E146AFB825F2092218D038867D7F337A010)),
)
CURVE_INFO = {_lib.ALT_BN128: ALT_BN128_INFO, _lib.BLS12_377: BLS12_377_INFO}


@dataclass
class CsrR1cs:
    """R1CS as three CSR matrices with integer coefficients: row_ptr[k] (nc + 1), col[k], val[k] (Python ints mod r)."""

    num_inputs: int
    num_variables: int  # primary + auxiliary, without the constant
    num_constraints: int
    row_ptr: List[np.ndarray] = field(default_factory=list)  # 3 x uint64
    col: List[np.ndarray] = field(default_factory=list)  # 3 x uint32
    val: List[list] = field(default_factory=list)  # 3 x list[int]

    def domain_size(self) -> int:
        need = self.num_constraints + self.num_inputs + 1
        m = 2
        while m < need:
            m *= 2
        return m

    @classmethod
    def from_rows(cls, num_inputs: int, num_aux: int, constraints, r: int) -> "CsrR1cs":
        """constraints: list of (a, b, c) dicts {variable index: coefficient}."""
        out = cls(num_inputs, num_inputs + num_aux, len(constraints))
        for k in range(3):
            rp = [0]
            cols: List[int] = []
            vals: List[int] = []
            for con in constraints:
                for idx, c in sorted(con[k].items()):
                    cols.append(idx)
                    vals.append(c % r)
                rp.append(len(cols))
            out.row_ptr.append(np.array(rp, dtype=np.uint64))
            out.col.append(np.array(cols, dtype=np.uint32))
            out.val.append(vals)
        return out

    def is_satisfied(self, full: Sequence[int], r: int) -> bool:
        w = [1] + [int(x) for x in full]
        ev = []
        for k in range(3):
            rp, col, val = self.row_ptr[k], self.col[k], self.val[k]
            ev.append([sum(val[t] * w[col[t]] for t in range(int(rp[i]), int(rp[i + 1]))) % r
                       for i in range(self.num_constraints)])
        return all(a * b % r == c for a, b, c in zip(*ev))


def _batch_inverse(xs: List[int], r: int) -> List[int]:
    pref = [1] * (len(xs) + 1)
    for i, x in enumerate(xs):
        pref[i + 1] = pref[i] * x % r
    inv = pow(pref[-1], -1, r)
    out = [0] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        out[i] = inv * pref[i] % r
        inv = inv * xs[i] % r
    return out


def lagrange_at(info: CurveInfo, m: int, t: int) -> List[int]:
    """L_i(t), i < m, over the m-th roots of unity (t outside the domain)."""
    r = info.r
    logm = m.bit_length() - 1
    omega = pow(info.fr_root, 1 << (info.fr_s - logm), r)
    Z = (pow(t, m, r) - 1) % r
    assert Z != 0, "tau must not be a domain element"
    w = [1] * m
    for i in range(1, m):
        w[i] = w[i - 1] * omega % r
    inv = _batch_inverse([(t - wi) % r for wi in w], r)
    c = Z * pow(m, -1, r) % r
    return [c * wi % r * iv % r for wi, iv in zip(w, inv)]


@dataclass
class VerificationKey:
    alpha_g1: tuple
    beta_g2: tuple
    delta_g2: tuple
    ABC_g1: list


def _mont_bytes(values: Sequence[int], mod: int, nbytes: int) -> bytes:
    R = 1 << (8 * nbytes)
    cache = {}
    out = bytearray()
    for v in values:
        b = cache.get(v)
        if b is None:
            b = (v * R % mod).to_bytes(nbytes, "little")
            if len(cache) < 4096:
                cache[v] = b
        out += b
    return bytes(out)


def _point_mont_bytes(info: CurveInfo, group: int, P) -> bytes:
    n = _lib.fq_bytes(info.curve)
    if P is None:
        return bytes(n * (2 if group == _lib.G1 else 4))
    coords = list(P) if group == _lib.G1 else [P[0][0], P[0][1], P[1][0], P[1][1]]
    return _mont_bytes(coords, info.q, n)


def generate_setup_from_secrets(ctx, curve: int, cs: CsrR1cs, t: int, alpha: int, beta: int, delta: int,
                                return_scalars: bool = False):
    """-> (zeth_amd.ProvingKey resident on the device, VerificationKey).  All secrets are explicit integers."""
    info = CURVE_INFO[curve]
    r = info.r
    m = cs.domain_size()
    nv, ni, nc = cs.num_variables, cs.num_inputs, cs.num_constraints
    u = lagrange_at(info, m, t)
    poly = [[0] * (nv + 1) for _ in range(3)]
    for k in range(3):
        rp, col, val, acc = cs.row_ptr[k], cs.col[k].tolist(), cs.val[k], poly[k]
        rpl = rp.tolist()
        for i in range(nc):
            ui = u[i]
            for tt in range(rpl[i], rpl[i + 1]):
                acc[col[tt]] += ui * val[tt]
    At, Bt, Ct = poly
    for i in range(ni + 1):
        At[i] += u[nc + i]
    At = [x % r for x in At]
    Bt = [x % r for x in Bt]
    Ct = [x % r for x in Ct]
    Zt = (pow(t, m, r) - 1) % r
    dinv = pow(delta, -1, r)
    h_sc = [0] * (m - 1)
    acc = Zt * dinv % r
    for i in range(m - 1):
        h_sc[i] = acc
        acc = acc * t % r
    abc = [(beta * a + alpha * b + c) % r for a, b, c in zip(At, Bt, Ct)]
    l_sc = [x * dinv % r for x in abc[ni + 1:]]
    b_idx = [i for i, x in enumerate(Bt) if x]
    b_sc = [Bt[i] for i in b_idx]

    G1, G2 = _lib.G1, _lib.G2
    keep = []

    plain = {}

    def query(group, scalars, name):
        bases = _lib.Bases.batch_exp(ctx, curve, group, scalars)
        raw = bases.read_bytes(0, bases.n, _lib.FORM_MONT)
        if return_scalars:
            plain[name] = bases.read_bytes(0, bases.n, _lib.FORM_PLAIN)
        bases.free()
        cb = ctypes.create_string_buffer(raw, max(1, len(raw)))
        keep.append(cb)
        return ctypes.cast(cb, ctypes.c_void_p)

    def buf(b: bytes):
        cb = ctypes.create_string_buffer(b, max(1, len(b)))
        keep.append(cb)
        return ctypes.cast(cb, ctypes.c_void_p)

    def gmul(group, k):
        return _lib.host_group_op(curve, group, "mul", info.g1 if group == G1 else info.g2, k % r)

    alpha_g1, beta_g1, delta_g1 = gmul(G1, alpha), gmul(G1, beta), gmul(G1, delta)
    beta_g2, delta_g2 = gmul(G2, beta), gmul(G2, delta)
    mats = []
    for k in range(3):
        rp = np.ascontiguousarray(cs.row_ptr[k], dtype=np.uint64)
        col = np.ascontiguousarray(cs.col[k], dtype=np.uint32)
        keep += [rp, col]
        mats.append(_lib.R1csMatrix(rp.ctypes.data if rp.size else None, col.ctypes.data if col.size else None,
                                    buf(_mont_bytes(cs.val[k], r, 32))))
    bidx = np.array(b_idx, dtype=np.uint32)
    keep.append(bidx)
    view = _lib.PkView(
        ni, nv, nc, mats[0], mats[1], mats[2],
        buf(_point_mont_bytes(info, G1, alpha_g1)), buf(_point_mont_bytes(info, G1, beta_g1)),
        buf(_point_mont_bytes(info, G2, beta_g2)), buf(_point_mont_bytes(info, G1, delta_g1)),
        buf(_point_mont_bytes(info, G2, delta_g2)),
        query(G1, At, "A"), nv + 1,
        bidx.ctypes.data if bidx.size else None, query(G2, b_sc, "B2"), query(G1, b_sc, "B1"), len(b_idx),
        query(G1, h_sc, "H"), m - 1,
        query(G1, l_sc, "L"), nv - ni,
        _lib.FORM_MONT,
    )
    h = ctypes.c_void_p()
    _lib._check(ctx._L.zg_pk_create(ctx._h, curve, ctypes.byref(view), ctypes.byref(h)))
    pk = _lib.ProvingKey(ctx, curve, h, None)
    vk = VerificationKey(alpha_g1, beta_g2, delta_g2, [gmul(G1, x) for x in abc[: ni + 1]])
    if return_scalars:
        return pk, vk, {"b_idx": b_idx, "plain": plain, "points": (alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2)}
    return pk, vk
