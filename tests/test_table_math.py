"""Host-side restatement of the arithmetic identities the fixed-base table mode relies on (zeth_amd/csrc/msm.cuh
recode_scalar / msm_make_table_plan, msm_host.cuh msm_finish), checked over the integers -- no GPU needed:
  * signed c-bit digits: s = sum_j d_j 2^(jc), d_j in [-2^(c-1), 2^(c-1)), D = ceil((bits + 2) / c) digits;
  * bucket b (weight b + 1) = group b mod W, key b // W:  sum_b (b+1) B_b = W (sum_g R_g - sum_g S_g) + sum_g (g+1) S_g
    with R_g = sum_k (k+1) B_{g,k} and S_g = sum_k B_{g,k};
  * hence sum_i s_i P_i = that expression when entry (i, j) adds sign(d_ij) * 2^(jc) * P_i to bucket |d_ij| - 1."""
import random

from zeth_model import BLS12_377, BN254


def recode(s, c, D):
    K = sum(1 << (j * c + c - 1) for j in range(D))
    t = s + K
    assert t < 1 << (D * c)
    return [((t >> (j * c)) & ((1 << c) - 1)) - (1 << (c - 1)) for j in range(D)]


def table_digits(bits, c):
    return (bits + 2 + c - 1) // c


def test_signed_digits_reconstruct_the_scalar():
    rng = random.Random(7)
    for C in (BN254, BLS12_377):
        bits = C.r.bit_length()
        for c in (10, 12, 17, 20, 21, 22):
            D = table_digits(bits, c)
            for s in [0, 1, 2, C.r - 1, C.r - 2, (1 << (bits - 1)) - 1, 1 << (bits - 1)] + [rng.randrange(C.r) for _ in range(200)]:
                d = recode(s, c, D)
                assert len(d) == D and all(-(1 << (c - 1)) <= x < (1 << (c - 1)) for x in d)
                assert sum(x << (j * c) for j, x in enumerate(d)) == s


def test_level_counts_match_the_library_policy():
    # BN254 (254-bit r): 12 digits at c = 22, 13 at c = 20, 16 at c = 17; BLS12-377 (253-bit r): 12, 13, 15
    assert BN254.r.bit_length() == 254 and BLS12_377.r.bit_length() == 253
    assert [table_digits(254, c) for c in (22, 20, 17)] == [12, 13, 16]
    assert [table_digits(253, c) for c in (22, 20, 17)] == [12, 13, 15]


def test_group_key_weights_and_whole_msm_over_integers():
    """Points are modelled by integers (P_i = p_i, group law = addition mod a prime), which keeps every step linear."""
    rng = random.Random(11)
    q = (1 << 61) - 1
    for c, gbits in ((10, 0), (17, 1), (20, 4), (22, 6)):
        bits = 254
        D = table_digits(bits, c)
        W = 1 << gbits
        n = 300
        pts = [rng.randrange(q) for _ in range(n)]
        scalars = [rng.choice([0, 1, rng.randrange(1 << 20), rng.randrange(BN254.r)]) for _ in range(n)]
        buckets = {}
        for i, s in enumerate(scalars):
            for j, d in enumerate(recode(s, c, D)):
                if d:
                    b = abs(d) - 1
                    term = (pts[i] << (j * c)) % q  # table level j
                    buckets[b] = (buckets.get(b, 0) + (term if d > 0 else -term)) % q
        R = [0] * W
        S = [0] * W
        for b, v in buckets.items():
            g, k = b & (W - 1), b >> gbits
            R[g] = (R[g] + (k + 1) * v) % q
            S[g] = (S[g] + v) % q
        total = (W * (sum(R) - sum(S)) + sum((g + 1) * S[g] for g in range(W))) % q
        assert total == sum((b + 1) * v for b, v in buckets.items()) % q
        assert total == sum(s * p for s, p in zip(scalars, pts)) % q
