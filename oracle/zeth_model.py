"""Big-integer model of Zeth's Groth16 prover hot path (TEST INFRASTRUCTURE ONLY).

This file is the *checker*, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  Plain Python ints, no dependencies.

What it restates (the arithmetic lives in clearmatics/libsnark + nested libff/libfqfft,
an un-vendored submodule that is EMPTY in /root/reference -- pinned version unrecoverable,
see SURVEY.md section 8c):

  * curve constants             pinned by client/tests/test_pairing.py:11-55,97-141 (KAT-4)
  * G1/G2 group law             pinned by libzeth/tests/core/group_element_utils_test.cpp:23-52 (KAT-1),
                                libzeth/tests/mpc/groth16/powersoftau_test.cpp:268-383 (KAT-2),
                                client/test_contracts/test_bls12_377_contract.py:21-75 (KAT-5)
  * multi_exp                   libzeth/core/multi_exp.tcc:13-35 (sum_i f_i * g_i over the first len(fs) bases)
  * radix-2 domain iFFT/FFT     libzeth/core/evaluator_from_lagrange.tcc:43, powersoftau_utils.tcc:125-134
  * r1cs_to_qap_witness_map, r1cs_gg_ppzksnark_prover / generator_from_secrets / verifier
                                call sites groth16_snark.tcc:23-24,44-45,54-55; key structure pinned by
                                libzeth/tests/mpc/groth16/mpc_test.cpp:282-331 (KAT-7); verifier equation pinned by
                                client/test_contracts/test_groth16_bls12_377_contract.py:15-63 (KAT-6) and
                                zeth_contracts/contracts/LibGroth16AltBN128.sol:160-161.

Parity status: group law, encodings and the verifier equation are pinned by the KATs above
(tests/test_oracle_kats.py).  MSM / NTT / proof *outputs* have no reference-authored vectors
("parity unpinned" in SURVEY 8c): they are pinned indirectly -- MSM by the group law, the NTT by
the DFT definition with libff's root of unity, the proof by the KAT-6-pinned verifier.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------------------
# Fq2 = Fq[u]/(u^2 - beta), elements are (c0, c1) tuples
# --------------------------------------------------------------------------------------


class Fq2Ops:
    def __init__(self, q: int, beta: int):
        self.q = q
        self.beta = beta % q

    def add(self, a, b):
        return ((a[0] + b[0]) % self.q, (a[1] + b[1]) % self.q)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.q, (a[1] - b[1]) % self.q)

    def neg(self, a):
        return ((-a[0]) % self.q, (-a[1]) % self.q)

    def mul(self, a, b):
        q = self.q
        return ((a[0] * b[0] + self.beta * a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)

    def sqr(self, a):
        return self.mul(a, a)

    def inv(self, a):
        q = self.q
        n = (a[0] * a[0] - self.beta * a[1] * a[1]) % q
        ni = pow(n, -1, q)
        return (a[0] * ni % q, (-a[1]) * ni % q)

    def muls(self, a, k: int):
        return (a[0] * k % self.q, a[1] * k % self.q)

    zero = (0, 0)
    one = (1, 0)

    def is_zero(self, a):
        return a[0] == 0 and a[1] == 0


class FqOps:
    def __init__(self, q: int):
        self.q = q

    def add(self, a, b):
        return (a + b) % self.q

    def sub(self, a, b):
        return (a - b) % self.q

    def neg(self, a):
        return (-a) % self.q

    def mul(self, a, b):
        return a * b % self.q

    def sqr(self, a):
        return a * a % self.q

    def inv(self, a):
        return pow(a, -1, self.q)

    def muls(self, a, k: int):
        return a * k % self.q

    zero = 0
    one = 1

    def is_zero(self, a):
        return a == 0


# --------------------------------------------------------------------------------------
# Short-Weierstrass group y^2 = x^3 + b (a = 0 for every group used here).
# Affine points are (x, y) tuples, the identity is None.
# --------------------------------------------------------------------------------------


class Group:
    def __init__(self, F, b, gen, order: int, name: str):
        self.F = F
        self.b = b
        self.gen = gen
        self.order = order
        self.name = name

    def is_on_curve(self, P) -> bool:
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sqr(y) == F.add(F.mul(F.sqr(x), x), self.b)

    def neg(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]))

    def add(self, P, Q):
        F = self.F
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if P[1] == Q[1]:
                return self.dbl(P)
            return None
        lam = F.mul(F.sub(Q[1], P[1]), F.inv(F.sub(Q[0], P[0])))
        x3 = F.sub(F.sub(F.sqr(lam), P[0]), Q[0])
        y3 = F.sub(F.mul(lam, F.sub(P[0], x3)), P[1])
        return (x3, y3)

    def dbl(self, P):
        F = self.F
        if P is None or F.is_zero(P[1]):
            return None
        lam = F.mul(F.muls(F.sqr(P[0]), 3), F.inv(F.muls(P[1], 2)))
        x3 = F.sub(F.sqr(lam), F.muls(P[0], 2))
        y3 = F.sub(F.mul(lam, F.sub(P[0], x3)), P[1])
        return (x3, y3)

    # Jacobian internals for speed (X, Y, Z); identity has Z == 0
    def _jdbl(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z):
            return P
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.muls(F.sub(F.sub(F.sqr(F.add(X, B)), A), C), 2)
        E = F.muls(A, 3)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.muls(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.muls(C, 8))
        Z3 = F.muls(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def _jadd_affine(self, P, Q):
        """Jacobian P + affine Q (Q not None)."""
        F = self.F
        X1, Y1, Z1 = P
        if F.is_zero(Z1):
            return (Q[0], Q[1], F.one)
        Z1Z1 = F.sqr(Z1)
        U2 = F.mul(Q[0], Z1Z1)
        S2 = F.mul(F.mul(Q[1], Z1), Z1Z1)
        if U2 == X1:
            if S2 == Y1:
                return self._jdbl(P)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, X1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        R = F.sub(S2, Y1)
        V = F.mul(X1, HH)
        X3 = F.sub(F.sub(F.sqr(R), HHH), F.muls(V, 2))
        Y3 = F.sub(F.mul(R, F.sub(V, X3)), F.mul(Y1, HHH))
        Z3 = F.mul(Z1, H)
        return (X3, Y3, Z3)

    def _to_affine(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def mul(self, k: int, P):
        """k * P (k reduced mod the group order, like libff's Fr * G)."""
        if P is None:
            return None
        k %= self.order
        if k == 0:
            return None
        F = self.F
        acc = (F.one, F.one, F.zero)
        for bit in bin(k)[2:]:
            acc = self._jdbl(acc)
            if bit == "1":
                acc = self._jadd_affine(acc, P)
        return self._to_affine(acc)

    def multi_exp(self, gs: Sequence, fs: Sequence[int]):
        """libzeth::multi_exp (multi_exp.tcc:25-35): uses the first len(fs) bases."""
        assert len(gs) >= len(fs)
        acc = None
        for g, f in zip(gs, fs):
            f %= self.order
            if f == 0 or g is None:
                continue
            acc = self.add(acc, g if f == 1 else self.mul(f, g))
        return acc


# --------------------------------------------------------------------------------------
# Pairing-friendly curve bundle
# --------------------------------------------------------------------------------------


class Curve:
    def __init__(self, name, q, r, g1, b1, beta, g2, xi, fr_gen, fr_s, fr_root, w12):
        self.name = name
        self.q = q
        self.r = r
        self.Fq = FqOps(q)
        self.Fq2 = Fq2Ops(q, beta)
        self.beta = beta % q
        self.xi = xi  # Fq2 element, twist: y^2 = x^3 + b / xi  (D-type, as libff)
        self.G1 = Group(self.Fq, b1 % q, g1, r, name + ".G1")
        b2 = self.Fq2.muls(self.Fq2.inv(xi), b1)
        self.G2 = Group(self.Fq2, b2, g2, r, name + ".G2")
        # Fr two-adicity data (libff Fr::multiplicative_generator / s / root_of_unity)
        self.fr_gen = fr_gen
        self.fr_s = fr_s
        self.fr_root = fr_root
        assert pow(fr_gen, (r - 1) >> fr_s, r) == fr_root
        assert pow(fr_root, 1 << (fr_s - 1), r) == r - 1
        # Fq12 = Fq[w]/(w^12 - w12[6] w^6 - w12[0]) and u = u_poly(w)
        self.w12 = w12
        self.fq_bytes = (q.bit_length() + 7) // 8
        self.fr_bytes = (r.bit_length() + 7) // 8
        self.fq_limbs32 = (q.bit_length() + 31) // 32
        self.fr_limbs32 = (r.bit_length() + 31) // 32

    # ---- Fr domain helpers (libfqfft basic_radix2_domain) ----
    def omega(self, m: int) -> int:
        """libfqfft get_root_of_unity(m): root_of_unity^(2^(s - log2 m))."""
        logm = m.bit_length() - 1
        assert 1 << logm == m and logm <= self.fr_s
        return pow(self.fr_root, 1 << (self.fr_s - logm), self.r)


BN254 = Curve(
    "alt-bn128",
    q=0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
    r=0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    g1=(1, 2),
    b1=3,
    beta=-1,
    g2=(
        (
            0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
            0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
        ),
        (
            0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
            0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B,
        ),
    ),
    xi=(9, 1),
    fr_gen=5,
    fr_s=28,
    fr_root=19103219067921713944291392827692070036145651957329286315305642004821462161904,
    # u = w^6 - 9, u^2 = -1  =>  w^12 = 18 w^6 - 82
    w12={0: -82, 6: 18, "u": {0: -9, 6: 1}},
)

BLS12_377 = Curve(
    "bls12-377",
    q=0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
    r=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    g1=(
        0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
        0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
    ),
    b1=1,
    beta=-5,
    g2=(
        (
            0x00B997FEF930828FE1B9E6A1707B8AA508A3DBFD7FE2246499C709226A0A6FEF49F85B3A375363F4F8F6EA3FBD159F8A,
            0x00D6AC33B84947D9845F81A57A136BFA326E915FABC8CD6A57FF133B42D00F62E4E1AF460228CD5184DEAE976FA62596,
        ),
        (
            0x0118DD509B2E9A13744A507D515A595DBB7E3B63DF568866473790184BDF83636C94DF2B7A962CB2AF4337F07CB7E622,
            0x0185067C6CA76D992F064A432BD9F9BE832B0CAC2D824D0518F77D39E76C3E146AFB825F2092218D038867D7F337A010,
        ),
    ),
    xi=(0, 1),
    fr_gen=22,
    fr_s=47,
    fr_root=8065159656716812877374967518403273466521432693661810619979959746626482506078,
    # u = w^6, u^2 = -5  =>  w^12 = -5
    w12={0: -5, 6: 0, "u": {0: 0, 6: 1}},
)

CURVES = {"alt-bn128": BN254, "bls12-377": BLS12_377}


# --------------------------------------------------------------------------------------
# Fq12 as Fq[w]/(w^12 - c6 w^6 - c0): generic, slow, only used by the verifier
# --------------------------------------------------------------------------------------


class Fq12Ops:
    def __init__(self, curve: Curve):
        self.q = curve.q
        self.c0 = curve.w12[0] % curve.q
        self.c6 = curve.w12[6] % curve.q
        self.u = curve.w12["u"]

    def one(self):
        return [1] + [0] * 11

    def from_fq2(self, a, shift: int = 0):
        """Embed (c0 + c1 u) * w^shift with u = u0 + u6 w^6 (shift < 6)."""
        q = self.q
        out = [0] * 12
        out[shift] = (a[0] + a[1] * self.u[0]) % q
        out[shift + 6] = (a[1] * self.u[6]) % q
        return out

    def mul(self, a, b):
        q = self.q
        t = [0] * 23
        for i, ai in enumerate(a):
            if ai == 0:
                continue
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
        for k in range(22, 11, -1):
            v = t[k] % q
            if v:
                t[k - 6] += v * self.c6
                t[k - 12] += v * self.c0
        return [x % q for x in t[:12]]

    def pow(self, a, e: int):
        result = self.one()
        for bit in bin(e)[2:]:
            result = self.mul(result, result)
            if bit == "1":
                result = self.mul(result, a)
        return result


def miller_tate(curve: Curve, P, Q):
    """f_{r,P}(Q) with P in G1 (affine over Fq), Q in G2 (affine on the twist), untwisted into
    E(Fq12) as (x' w^2, y' w^3).  Vertical lines lie in Fq6 and die in the final exponentiation."""
    K = Fq12Ops(curve)
    q = curve.q
    if P is None or Q is None:
        return K.one()
    xq = K.from_fq2(Q[0], 2)
    yq = K.from_fq2(Q[1], 3)

    def line(T, lam):
        # (yQ - yT) - lam (xQ - xT)
        out = [(yq[i] - lam * xq[i]) % q for i in range(12)]
        out[0] = (out[0] - T[1] + lam * T[0]) % q
        return out

    f = K.one()
    T = P
    bits = bin(curve.r)[3:]
    for idx, bit in enumerate(bits):
        lam = 3 * T[0] * T[0] * pow(2 * T[1], -1, q) % q
        f = K.mul(K.mul(f, f), line(T, lam))
        x3 = (lam * lam - 2 * T[0]) % q
        T = (x3, (lam * (T[0] - x3) - T[1]) % q)
        if bit == "1":
            if T[0] == P[0]:
                # T == -P at the very last step (r P = O): vertical line, eliminated
                assert idx == len(bits) - 1 and (T[1] + P[1]) % q == 0
                T = None
                continue
            lam = (P[1] - T[1]) * pow(P[0] - T[0], -1, q) % q
            f = K.mul(f, line(T, lam))
            x3 = (lam * lam - T[0] - P[0]) % q
            T = (x3, (lam * (T[0] - x3) - T[1]) % q)
    return f


def pairing_product_is_one(curve: Curve, pairs) -> bool:
    """prod_i e(P_i, Q_i) == 1 in GT (reduced Tate pairing; one shared final exponentiation)."""
    K = Fq12Ops(curve)
    f = K.one()
    for P, Q in pairs:
        f = K.mul(f, miller_tate(curve, P, Q))
    e = (curve.q ** 12 - 1) // curve.r
    return K.pow(f, e) == K.one()


# --------------------------------------------------------------------------------------
# radix-2 evaluation domain (libfqfft basic_radix2_domain semantics)
# --------------------------------------------------------------------------------------


def _bitrev(i: int, logn: int) -> int:
    return int(bin(i)[2:].zfill(logn)[::-1], 2) if logn else 0


def fft_inplace(a: List[int], omega: int, r: int) -> None:
    """out[k] = sum_j a[j] omega^(jk): bit-reverse then log n DIT stages (libfqfft _basic_radix2_FFT)."""
    n = len(a)
    logn = n.bit_length() - 1
    assert 1 << logn == n
    for k in range(n):
        rk = _bitrev(k, logn)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(logn):
        w_m = pow(omega, n // (2 * m), r)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = w * a[k + j + m] % r
                a[k + j + m] = (a[k + j] - t) % r
                a[k + j] = (a[k + j] + t) % r
                w = w * w_m % r
        m *= 2


def dft_naive(a: Sequence[int], omega: int, r: int) -> List[int]:
    n = len(a)
    return [sum(a[j] * pow(omega, j * k, r) for j in range(n)) % r for k in range(n)]


class Domain:
    """basic_radix2_domain<Fr>(m): FFT, iFFT, cosetFFT, icosetFFT, divide_by_Z_on_coset."""

    def __init__(self, curve: Curve, m: int):
        self.curve = curve
        self.r = curve.r
        self.m = m
        self.omega = curve.omega(m)

    def FFT(self, a):
        assert len(a) == self.m
        fft_inplace(a, self.omega, self.r)

    def iFFT(self, a):
        assert len(a) == self.m
        fft_inplace(a, pow(self.omega, -1, self.r), self.r)
        minv = pow(self.m, -1, self.r)
        for i in range(self.m):
            a[i] = a[i] * minv % self.r

    def cosetFFT(self, a, g):
        u = 1
        for i in range(self.m):
            a[i] = a[i] * u % self.r
            u = u * g % self.r
        self.FFT(a)

    def icosetFFT(self, a, g):
        self.iFFT(a)
        gi = pow(g, -1, self.r)
        u = 1
        for i in range(self.m):
            a[i] = a[i] * u % self.r
            u = u * gi % self.r

    def compute_vanishing_polynomial(self, t):
        return (pow(t, self.m, self.r) - 1) % self.r

    def divide_by_Z_on_coset(self, a, g):
        zi = pow(self.compute_vanishing_polynomial(g), -1, self.r)
        for i in range(self.m):
            a[i] = a[i] * zi % self.r

    def get_domain_element(self, i):
        return pow(self.omega, i, self.r)

    def evaluate_all_lagrange_polynomials(self, t):
        """L_i(t) for i < m (the domain is the m-th roots of unity)."""
        r, m = self.r, self.m
        if pow(t, m, r) == 1:
            out = [0] * m
            w = 1
            for i in range(m):
                if w == t % r:
                    out[i] = 1
                    return out
                w = w * self.omega % r
        Z = self.compute_vanishing_polynomial(t)
        minv = pow(m, -1, r)
        out = []
        w = 1
        for i in range(m):
            out.append(Z * minv % r * w % r * pow(t - w, -1, r) % r)
            w = w * self.omega % r
        return out


# --------------------------------------------------------------------------------------
# R1CS, QAP witness map, Groth16 (r1cs_gg_ppzksnark as used by groth16_snark.tcc:18-56)
# --------------------------------------------------------------------------------------

LinComb = Dict[int, int]  # variable index (0 = constant one) -> coefficient


class R1CS:
    def __init__(self, num_inputs: int, num_aux: int, constraints: List[Tuple[LinComb, LinComb, LinComb]]):
        self.num_inputs = num_inputs
        self.num_aux = num_aux
        self.constraints = constraints

    @property
    def num_variables(self):
        return self.num_inputs + self.num_aux

    @property
    def num_constraints(self):
        return len(self.constraints)

    def domain_size(self) -> int:
        """force_pow_2_domain=true (groth16_snark.tcc:43-45): next pow2 >= num_constraints + num_inputs + 1."""
        need = self.num_constraints + self.num_inputs + 1
        m = 1
        while m < need:
            m *= 2
        return max(m, 2)

    def is_satisfied(self, full: Sequence[int], r: int) -> bool:
        w = [1] + list(full)
        ev = lambda lc: sum(c * w[i] for i, c in lc.items()) % r  # noqa: E731
        return all(ev(a) * ev(b) % r == ev(c) for a, b, c in self.constraints)


def r1cs_to_qap_witness_map(curve: Curve, cs: R1CS, primary: Sequence[int], aux: Sequence[int]) -> List[int]:
    """coefficients_for_H (length m+1; d1=d2=d3=0 as the gg prover passes)."""
    r = curve.r
    m = cs.domain_size()
    dom = Domain(curve, m)
    full = list(primary) + list(aux)
    w = [1] + full
    ev = lambda lc: sum(c * w[i] for i, c in lc.items()) % r  # noqa: E731
    aA = [0] * m
    aB = [0] * m
    aC = [0] * m
    nc = cs.num_constraints
    # input-consistency rows: A_i gets an extra Lagrange term at nc + i for i <= num_inputs
    for i in range(cs.num_inputs + 1):
        aA[nc + i] = w[i]
    for i, (a, b, c) in enumerate(cs.constraints):
        aA[i] = (aA[i] + ev(a)) % r
        aB[i] = (aB[i] + ev(b)) % r
        aC[i] = (aC[i] + ev(c)) % r
    g = curve.fr_gen
    dom.iFFT(aA)
    dom.iFFT(aB)
    dom.cosetFFT(aA, g)
    dom.cosetFFT(aB, g)
    dom.iFFT(aC)
    dom.cosetFFT(aC, g)
    H = [(aA[i] * aB[i] - aC[i]) % r for i in range(m)]
    dom.divide_by_Z_on_coset(H, g)
    dom.icosetFFT(H, g)
    return H + [0]


class ProvingKey:
    def __init__(self):
        self.alpha_g1 = None
        self.beta_g1 = None
        self.beta_g2 = None
        self.delta_g1 = None
        self.delta_g2 = None
        self.A_query: List = []  # num_variables + 1 G1 points
        self.B_query: List[Tuple[int, object, object]] = []  # sparse (index, G2, G1), zero entries skipped
        self.B_domain_size = 0
        self.H_query: List = []  # m - 1
        self.L_query: List = []  # num_variables - num_inputs
        self.cs: Optional[R1CS] = None


class VerificationKey:
    def __init__(self):
        self.alpha_g1 = None
        self.beta_g2 = None
        self.delta_g2 = None
        self.ABC_g1: List = []  # num_inputs + 1


def qap_instance_evaluation(curve: Curve, cs: R1CS, t: int):
    """At, Bt, Ct (num_variables+1 each), Ht (m+1 powers), Zt; mirrors r1cs_to_qap_instance_map_with_evaluation."""
    r = curve.r
    m = cs.domain_size()
    dom = Domain(curve, m)
    u = dom.evaluate_all_lagrange_polynomials(t)
    nv = cs.num_variables
    At = [0] * (nv + 1)
    Bt = [0] * (nv + 1)
    Ct = [0] * (nv + 1)
    nc = cs.num_constraints
    for i in range(cs.num_inputs + 1):
        At[i] = u[nc + i]
    for i, (a, b, c) in enumerate(cs.constraints):
        for idx, coeff in a.items():
            At[idx] = (At[idx] + u[i] * coeff) % r
        for idx, coeff in b.items():
            Bt[idx] = (Bt[idx] + u[i] * coeff) % r
        for idx, coeff in c.items():
            Ct[idx] = (Ct[idx] + u[i] * coeff) % r
    Ht = [pow(t, i, r) for i in range(m + 1)]
    Zt = dom.compute_vanishing_polynomial(t)
    return At, Bt, Ct, Ht, Zt


def generator_from_secrets(curve: Curve, cs: R1CS, t: int, alpha: int, beta: int, delta: int):
    """r1cs_gg_ppzksnark_generator_from_secrets(cs, t, alpha, beta, delta, G1::one(), G2::one(), true);
    key structure as asserted by mpc_test.cpp:282-331."""
    r = curve.r
    G1, G2 = curve.G1, curve.G2
    At, Bt, Ct, Ht, Zt = qap_instance_evaluation(curve, cs, t)
    m = cs.domain_size()
    nv, ni = cs.num_variables, cs.num_inputs
    dinv = pow(delta, -1, r)
    pk = ProvingKey()
    pk.cs = cs
    pk.alpha_g1 = G1.mul(alpha, G1.gen)
    pk.beta_g1 = G1.mul(beta, G1.gen)
    pk.beta_g2 = G2.mul(beta, G2.gen)
    pk.delta_g1 = G1.mul(delta, G1.gen)
    pk.delta_g2 = G2.mul(delta, G2.gen)
    pk.A_query = [G1.mul(x, G1.gen) for x in At]
    pk.B_domain_size = nv + 1
    pk.B_query = [(i, G2.mul(x, G2.gen), G1.mul(x, G1.gen)) for i, x in enumerate(Bt) if x % r]
    pk.H_query = [G1.mul(Ht[i] * Zt % r * dinv, G1.gen) for i in range(m - 1)]
    abc = [(beta * At[i] + alpha * Bt[i] + Ct[i]) % r for i in range(nv + 1)]
    pk.L_query = [G1.mul(abc[i] * dinv, G1.gen) for i in range(ni + 1, nv + 1)]
    vk = VerificationKey()
    vk.alpha_g1 = pk.alpha_g1
    vk.beta_g2 = pk.beta_g2
    vk.delta_g2 = pk.delta_g2
    vk.ABC_g1 = [G1.mul(abc[i], G1.gen) for i in range(ni + 1)]
    return pk, vk


def prove(curve: Curve, pk: ProvingKey, primary: Sequence[int], aux: Sequence[int], r_rand: int, s_rand: int):
    """r1cs_gg_ppzksnark_prover(pk, primary, aux, force_pow_2_domain=true) with (r, s) injected."""
    G1, G2 = curve.G1, curve.G2
    cs = pk.cs
    assert len(primary) == cs.num_inputs and len(aux) == cs.num_aux
    H = r1cs_to_qap_witness_map(curve, cs, primary, aux)
    m = cs.domain_size()
    w = [1] + list(primary) + list(aux)
    At = G1.multi_exp(pk.A_query, w)
    Bt_g2 = None
    Bt_g1 = None
    for idx, g2p, g1p in pk.B_query:
        f = w[idx] % curve.r
        if f:
            Bt_g2 = G2.add(Bt_g2, G2.mul(f, g2p))
            Bt_g1 = G1.add(Bt_g1, G1.mul(f, g1p))
    Ht = G1.multi_exp(pk.H_query, H[: m - 1])
    Lt = G1.multi_exp(pk.L_query, w[cs.num_inputs + 1 :])
    g_A = G1.add(G1.add(pk.alpha_g1, At), G1.mul(r_rand, pk.delta_g1))
    g1_B = G1.add(G1.add(pk.beta_g1, Bt_g1), G1.mul(s_rand, pk.delta_g1))
    g_B = G2.add(G2.add(pk.beta_g2, Bt_g2), G2.mul(s_rand, pk.delta_g2))
    g_C = G1.add(Ht, Lt)
    g_C = G1.add(g_C, G1.mul(s_rand, g_A))
    g_C = G1.add(g_C, G1.mul(r_rand, g1_B))
    g_C = G1.add(g_C, G1.neg(G1.mul(r_rand * s_rand % curve.r, pk.delta_g1)))
    return g_A, g_B, g_C


def verify(curve: Curve, vk: VerificationKey, primary: Sequence[int], proof) -> bool:
    """e(A,B) = e(alpha,beta) e(sum x_i ABC_i, G2::one()) e(C,delta)  (LibGroth16AltBN128.sol:160-161)."""
    G1, G2 = curve.G1, curve.G2
    g_A, g_B, g_C = proof
    if len(primary) + 1 != len(vk.ABC_g1):
        return False
    if not (G1.is_on_curve(g_A) and G2.is_on_curve(g_B) and G1.is_on_curve(g_C)):
        return False
    acc = vk.ABC_g1[0]
    for x, P in zip(primary, vk.ABC_g1[1:]):
        acc = G1.add(acc, G1.mul(x, P))
    return pairing_product_is_one(
        curve,
        [
            (g_A, g_B),
            (G1.neg(vk.alpha_g1), vk.beta_g2),
            (G1.neg(acc), G2.gen),
            (G1.neg(g_C), vk.delta_g2),
        ],
    )


# --------------------------------------------------------------------------------------
# Encodings (libzeth/core/{field,group}_element_utils.tcc): big-endian plain field bytes,
# affine X||Y, Fq2 written c1 then c0 (KAT-5), identity as (0, 1).
# --------------------------------------------------------------------------------------


def fq_to_bytes(curve: Curve, x: int) -> bytes:
    return x.to_bytes(curve.fq_bytes, "big")


def g1_to_bytes(curve: Curve, P) -> bytes:
    if P is None:
        P = (0, 1)
    return fq_to_bytes(curve, P[0]) + fq_to_bytes(curve, P[1])


def g2_to_bytes(curve: Curve, P) -> bytes:
    if P is None:
        P = ((0, 0), (1, 0))
    (x0, x1), (y0, y1) = P
    return fq_to_bytes(curve, x1) + fq_to_bytes(curve, x0) + fq_to_bytes(curve, y1) + fq_to_bytes(curve, y0)


def g1_from_bytes(curve: Curve, b: bytes):
    n = curve.fq_bytes
    x = int.from_bytes(b[:n], "big")
    y = int.from_bytes(b[n : 2 * n], "big")
    return None if (x, y) == (0, 1) else (x, y)


def g2_from_bytes(curve: Curve, b: bytes):
    n = curve.fq_bytes
    v = [int.from_bytes(b[i * n : (i + 1) * n], "big") for i in range(4)]
    P = ((v[1], v[0]), (v[3], v[2]))
    return None if P == ((0, 0), (1, 0)) else P


def g1_to_json(curve: Curve, P) -> str:
    n = curve.fq_bytes
    if P is None:
        P = (0, 1)
    return '["0x%s","0x%s"]' % (P[0].to_bytes(n, "big").hex(), P[1].to_bytes(n, "big").hex())


def g2_to_json(curve: Curve, P) -> str:
    n = curve.fq_bytes
    h = lambda v: v.to_bytes(n, "big").hex()  # noqa: E731
    (x0, x1), (y0, y1) = P
    return '[["0x%s","0x%s"],["0x%s","0x%s"]]' % (h(x1), h(x0), h(y1), h(y0))


# Stream formats (libzeth/serialization/stream_utils.tcc, r1cs_serialization.tcc:199-275,
# r1cs_variable_assignment_serialization.tcc:15-77, snarks/groth16/groth16_snark.tcc:93-175): size_t and
# var_index_t are native 8-byte little-endian, the per-linear-combination term count is a uint32.
def fr_to_bytes(x: int) -> bytes:
    return x.to_bytes(32, "big")


def _u64(x: int) -> bytes:
    return x.to_bytes(8, "little")


def g1_vector_to_bytes(curve: Curve, pts) -> bytes:
    return _u64(len(pts)) + b"".join(g1_to_bytes(curve, P) for P in pts)


def g2_vector_to_bytes(curve: Curve, pts) -> bytes:
    return _u64(len(pts)) + b"".join(g2_to_bytes(curve, P) for P in pts)


def linear_combination_to_bytes(curve: Curve, lc) -> bytes:
    """lc: dict or list of (index, coeff) pairs, written in stored order."""
    terms = list(lc.items()) if isinstance(lc, dict) else list(lc)
    out = len(terms).to_bytes(4, "little")
    for idx, coeff in terms:
        out += _u64(idx) + fr_to_bytes(coeff % curve.r)
    return out


def r1cs_to_bytes(curve: Curve, cs: "R1CS") -> bytes:
    out = _u64(cs.num_inputs) + _u64(cs.num_aux) + _u64(cs.num_constraints)
    for a, b, c in cs.constraints:
        out += linear_combination_to_bytes(curve, a) + linear_combination_to_bytes(curve, b) + linear_combination_to_bytes(curve, c)
    return out


def assignment_to_bytes(curve: Curve, values) -> bytes:
    return _u64(len(values)) + b"".join(fr_to_bytes(v % curve.r) for v in values)


def assignment_from_bytes(b: bytes):
    n = int.from_bytes(b[:8], "little")
    assert len(b) == 8 + 32 * n
    return [int.from_bytes(b[8 + 32 * i : 40 + 32 * i], "big") for i in range(n)]


def proving_key_to_bytes(curve: Curve, pk: "ProvingKey") -> bytes:
    out = g1_to_bytes(curve, pk.alpha_g1) + g1_to_bytes(curve, pk.beta_g1) + g2_to_bytes(curve, pk.beta_g2)
    out += g1_to_bytes(curve, pk.delta_g1) + g2_to_bytes(curve, pk.delta_g2)
    out += g1_vector_to_bytes(curve, pk.A_query)
    parts = [_u64(pk.B_domain_size), _u64(len(pk.B_query))]
    for idx, g2p, g1p in pk.B_query:  # knowledge_commitment: g (G2) then h (G1)
        parts.append(_u64(idx) + g2_to_bytes(curve, g2p) + g1_to_bytes(curve, g1p))
    out += b"".join(parts)
    out += g1_vector_to_bytes(curve, pk.H_query) + g1_vector_to_bytes(curve, pk.L_query)
    return out + r1cs_to_bytes(curve, pk.cs)


def verification_key_to_bytes(curve: Curve, vk: "VerificationKey") -> bytes:
    """groth16_snark.tcc:76-91: alpha_g1, beta_g2, delta_g2, then ABC_g1 as libsnark's accumulation_vector: the first
    element, then the rest as a sparse vector (domain size, entry count, (index, value) pairs; indices 0..n-1) --
    stream_utils.tcc:110-139."""
    rest = vk.ABC_g1[1:]
    out = g1_to_bytes(curve, vk.alpha_g1) + g2_to_bytes(curve, vk.beta_g2) + g2_to_bytes(curve, vk.delta_g2)
    out += g1_to_bytes(curve, vk.ABC_g1[0]) + _u64(len(rest)) + _u64(len(rest))
    for i, P in enumerate(rest):
        out += _u64(i) + g1_to_bytes(curve, P)
    return out


def verification_key_from_bytes(curve: Curve, raw: bytes) -> "VerificationKey":
    """groth16_snark.tcc:113-128 (the well-formedness check is the caller's)."""
    n1, n2 = 2 * curve.fq_bytes, 4 * curve.fq_bytes
    vk = VerificationKey()
    vk.alpha_g1 = g1_from_bytes(curve, raw[:n1])
    vk.beta_g2 = g2_from_bytes(curve, raw[n1 : n1 + n2])
    vk.delta_g2 = g2_from_bytes(curve, raw[n1 + n2 : n1 + 2 * n2])
    pos = n1 + 2 * n2
    vk.ABC_g1 = [g1_from_bytes(curve, raw[pos : pos + n1])]
    count = int.from_bytes(raw[pos + n1 + 8 : pos + n1 + 16], "little")
    pos += n1 + 16
    for i in range(count):
        assert int.from_bytes(raw[pos : pos + 8], "little") == i
        vk.ABC_g1.append(g1_from_bytes(curve, raw[pos + 8 : pos + 8 + n1]))
        pos += 8 + n1
    assert pos == len(raw)
    return vk


def verification_key_to_json(curve: Curve, vk: "VerificationKey") -> str:
    """groth16_snark.tcc:58-74."""
    out = '{\n  "alpha": %s,\n  "beta": %s,\n  "delta": %s,\n  "ABC": [\n    %s' % (
        g1_to_json(curve, vk.alpha_g1), g2_to_json(curve, vk.beta_g2), g2_to_json(curve, vk.delta_g2), g1_to_json(curve, vk.ABC_g1[0]))
    for P in vk.ABC_g1[1:]:
        out += ",\n    " + g1_to_json(curve, P)
    return out + "\n  ]\n}"


def keypair_to_bytes(curve: Curve, pk: "ProvingKey", vk: "VerificationKey") -> bytes:
    """groth16_snark.tcc:177-183: the proving key, then the verification key (prover_server's keypair.bin)."""
    return proving_key_to_bytes(curve, pk) + verification_key_to_bytes(curve, vk)


def swap_AB_if_beneficial(cs: "R1CS") -> "R1CS":
    """libsnark r1cs_constraint_system::swap_AB_if_beneficial (restated from the published libsnark source; used by the
    reference at libzeth/tests/mpc/groth16/mpc_test.cpp:42,164,340): swap a and b everywhere when B touches more
    variables than A."""
    touched_a, touched_b = set(), set()
    for a, b, _ in cs.constraints:
        touched_a.update(dict(a).keys() if not isinstance(a, dict) else a.keys())
        touched_b.update(dict(b).keys() if not isinstance(b, dict) else b.keys())
    if len(touched_b) <= len(touched_a):
        return cs
    return R1CS(cs.num_inputs, cs.num_aux, [(b, a, c) for a, b, c in cs.constraints])


def proof_to_bytes(curve: Curve, proof) -> bytes:
    g_A, g_B, g_C = proof
    return g1_to_bytes(curve, g_A) + g2_to_bytes(curve, g_B) + g1_to_bytes(curve, g_C)


def proof_from_bytes(curve: Curve, b: bytes):
    n = curve.fq_bytes
    assert len(b) == 8 * n
    return g1_from_bytes(curve, b[: 2 * n]), g2_from_bytes(curve, b[2 * n : 6 * n]), g1_from_bytes(curve, b[6 * n :])


def proof_to_json(curve: Curve, proof) -> str:
    g_A, g_B, g_C = proof
    return '{\n  "a": %s,\n  "b": %s,\n  "c": %s\n}' % (g1_to_json(curve, g_A), g2_to_json(curve, g_B), g1_to_json(curve, g_C))


# powersoftau encodings (powersoftau_utils.tcc:485-505, 590-622): 0x04 marker, Fq2 packed as c1*q + c0
def pot_read_g1(curve: Curve, b: bytes, off: int):
    n = curve.fq_bytes
    if b[off] == 0:
        return None, off + 1 + 2 * n
    assert b[off] == 4
    x = int.from_bytes(b[off + 1 : off + 1 + n], "big")
    y = int.from_bytes(b[off + 1 + n : off + 1 + 2 * n], "big")
    return (x, y), off + 1 + 2 * n


def pot_read_g2(curve: Curve, b: bytes, off: int):
    n = 2 * curve.fq_bytes
    if b[off] == 0:
        return None, off + 1 + 2 * n
    assert b[off] == 4
    xv = int.from_bytes(b[off + 1 : off + 1 + n], "big")
    yv = int.from_bytes(b[off + 1 + n : off + 1 + 2 * n], "big")
    q = curve.q
    return ((xv % q, xv // q), (yv % q, yv // q)), off + 1 + 2 * n


def pot_load(curve: Curve, data: bytes, n: int):
    """powersoftau_load layout: 64-byte hash, 2n-1 tau G1, n tau G2, n alpha-tau G1, n beta-tau G1, beta G2."""
    off = 64
    tau_g1 = []
    for _ in range(2 * n - 1):
        P, off = pot_read_g1(curve, data, off)
        tau_g1.append(P)
    tau_g2 = []
    for _ in range(n):
        P, off = pot_read_g2(curve, data, off)
        tau_g2.append(P)
    alpha_g1 = []
    for _ in range(n):
        P, off = pot_read_g1(curve, data, off)
        alpha_g1.append(P)
    beta_g1 = []
    for _ in range(n):
        P, off = pot_read_g1(curve, data, off)
        beta_g1.append(P)
    beta_g2, off = pot_read_g2(curve, data, off)
    return {"tau_g1": tau_g1, "tau_g2": tau_g2, "alpha_tau_g1": alpha_g1, "beta_tau_g1": beta_g1, "beta_g2": beta_g2}, off


# --------------------------------------------------------------------------------------
# Fixture circuits from the reference tests
# --------------------------------------------------------------------------------------


def simple_circuit(curve: Curve, copies: int = 1) -> R1CS:
    """x^3 + 4x^2 + 2x + 5 = y  (libzeth/tests/circuits/simple_test.tcc:15-62): per copy the variables
    y, x, g1, g2 are allocated in that order and 3 constraints added; set_input_sizes(1) means only the FIRST
    y is a primary input, exactly as the 4x stacked circuit of common_snark_tests.tcc:22-62."""
    cons = []
    for k in range(copies):
        y, x, g1, g2 = 1 + 4 * k, 2 + 4 * k, 3 + 4 * k, 4 + 4 * k
        cons.append(({x: 1}, {x: 1}, {g1: 1}))
        cons.append(({g1: 1}, {x: 1}, {g2: 1}))
        cons.append(({g2: 1, g1: 4, x: 2, 0: 5}, {0: 1}, {y: 1}))
    return R1CS(1, 4 * copies - 1, cons)


def simple_circuit_assignment(curve: Curve, xs: Sequence[int]):
    """simple_test.tcc:49-62 applied the way common_snark_tests.tcc:46-54 does: the first y is primary,
    every later y is pushed onto the auxiliary vector ahead of its x, g1, g2."""
    r = curve.r
    primary: List[int] = []
    aux: List[int] = []
    for k, x in enumerate(xs):
        g1 = x * x % r
        g2 = g1 * x % r
        y = (g2 + 4 * g1 + 2 * x + 5) % r
        (primary if k == 0 else aux).append(y)
        aux += [x % r, g1, g2]
    return primary, aux
