"""Synthetic "JoinSplit-shaped" R1CS: the real Zeth circuit needs libsnark's gadgetlib1, which is not
available, so benchmarks use a satisfiable system with the same SHAPE -- ~4.3e5 constraints, ~5.5e5 variables, one
public input, ~90 % boolean wires (the BLAKE2s bit wires, libzeth/circuits/circuit_types.hpp:24), short linear
combinations, a MiMC-like chain of field-valued wires (libzeth/circuits/mimc/mimc_selector.hpp:25-32) and 32-bit
packing constraints -- domain size 2^19.  It is NOT the Zeth circuit; results say so."""
from __future__ import annotations

import numpy as np

from .keygen import CsrR1cs


def joinsplit_shaped(r: int, seed: int = 1, scale: float = 1.0, constraints: int = 0):
    """-> (CsrR1cs, primary, aux).  scale < 1 shrinks every count proportionally (tests); `constraints` asks for an
    exact constraint count instead (BASELINE.json configs[3]: 2^22 - 2, so that the domain is 2^22)."""
    rng = np.random.default_rng(seed)
    if constraints:
        scale = (constraints - 1) / 436_000.0
    n_bool_checked = int(230_000 * scale)    # free bits with a booleanity constraint x (1 - x) = 0
    n_bool_free = int(110_000 * scale)       # bits whose booleanity is implied elsewhere
    n_and = int(160_000 * scale)             # z = x AND y  (x * y = z)
    n_chain = int(43_000 * scale)            # s' = (s + k)^2, field valued
    n_pack = max(1, int(3_000 * scale))      # sum 2^i b_i = word
    if constraints:
        n_bool_checked += constraints - (n_bool_checked + n_and + n_chain + n_pack + 1)
    n_free = n_bool_checked + n_bool_free
    # variable indices (0 is the constant one, 1 the public input)
    v_bits = 2 + np.arange(n_free)
    v_and = v_bits[-1] + 1 + np.arange(n_and) if n_and else np.zeros(0, dtype=np.int64)
    base = 2 + n_free + n_and
    v_chain = base + np.arange(n_chain)
    v_pack = base + n_chain + np.arange(n_pack)
    nv = 1 + n_free + n_and + n_chain + n_pack
    w = [0] * (nv + 1)
    w[0] = 1
    bits = rng.integers(0, 2, n_free)
    for i, b in zip(v_bits.tolist(), bits.tolist()):
        w[i] = b
    ax = rng.integers(0, n_free, n_and)
    ay = rng.integers(0, n_free, n_and)
    for z, x, y in zip(v_and.tolist(), ax.tolist(), ay.tolist()):
        w[z] = bits[x] & bits[y]
    ks = [int.from_bytes(rng.bytes(31), "little") % r for _ in range(n_chain)]
    s_prev = int.from_bytes(rng.bytes(31), "little") % r
    chain_prev = []
    # the chain starts from a packed word so that it depends on the bit wires
    pack_bits = rng.integers(0, n_free, (n_pack, 32))
    for k in range(n_pack):
        w[int(v_pack[k])] = sum(int(bits[j]) << i for i, j in enumerate(pack_bits[k].tolist()))
    prev_var = int(v_pack[0])
    s_prev = w[prev_var]
    for k in range(n_chain):
        chain_prev.append(prev_var)
        s_prev = (s_prev + ks[k]) * (s_prev + ks[k]) % r
        w[int(v_chain[k])] = s_prev
        prev_var = int(v_chain[k])
    # public input = last chain value + sum of the packed words (a stand-in for "hash of the public data")
    n_out = min(n_pack, 64)  # the public input binds the chain output and the first packed words
    w[1] = (s_prev + sum(w[int(v)] for v in v_pack.tolist()[:n_out])) % r

    rows = [[], [], []]  # per matrix: list of (cols, vals)
    neg1 = r - 1

    def add(a, b, c):
        rows[0].append(a)
        rows[1].append(b)
        rows[2].append(c)

    for i in v_bits[:n_bool_checked].tolist():
        add(([i], [1]), ([0, i], [1, neg1]), ([], []))
    for z, x, y in zip(v_and.tolist(), ax.tolist(), ay.tolist()):
        add(([2 + x], [1]), ([2 + y], [1]), ([z], [1]))
    for k in range(n_chain):
        lc = ([0, chain_prev[k]], [ks[k], 1])
        add(lc, lc, ([int(v_chain[k])], [1]))
    pw = [1 << i for i in range(32)]
    for k in range(n_pack):
        # duplicate bit indices in one row are merged (libsnark keeps linear combinations sorted and unique)
        acc = {}
        for i, j in enumerate(pack_bits[k].tolist()):
            acc[2 + j] = acc.get(2 + j, 0) + pw[i]
        cols = sorted(acc)
        add((cols, [acc[c] for c in cols]), ([0], [1]), ([int(v_pack[k])], [1]))
    add(([prev_var] + [int(v) for v in v_pack.tolist()[:n_out]], [1] * (1 + n_out)), ([0], [1]), ([1], [1]))
    nc = len(rows[0])
    cs = CsrR1cs(1, nv, nc)
    for k in range(3):
        rp = np.zeros(nc + 1, dtype=np.uint64)
        cols, vals = [], []
        for i, (c, v) in enumerate(rows[k]):
            order = sorted(range(len(c)), key=lambda t: c[t])
            cols.extend(c[t] for t in order)
            vals.extend(v[t] % r for t in order)
            rp[i + 1] = len(cols)
        cs.row_ptr.append(rp)
        cs.col.append(np.array(cols, dtype=np.uint32))
        cs.val.append(vals)
    return cs, [w[1]], w[2:]
