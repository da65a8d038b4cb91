"""Host-side mirror of the library's synthetic benchmark inputs: reproduces zg_scalars_random in
numpy and evaluates the closed-form expected scalar for bases P_i = (t0 + i d) G, so that an MSM of any size can be
checked with O(n) integer work.  Checker-side helper: nothing here is on the measured path."""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def random_scalars_host(n: int, r: int, seed: int) -> np.ndarray:
    """n x 4 little-endian uint64 words, identical to zg_scalars_random(curve, n, seed)."""
    bits = r.bit_length()
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    p = [np.uint64((r >> (64 * k)) & 0xFFFFFFFFFFFFFFFF) for k in range(4)]
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n, dtype=np.uint64)
    t = 0
    with np.errstate(over="ignore"):
        while todo.size:
            ctr = ((np.uint64(seed) + todo) << np.uint64(8)) + np.uint64(t)
            w = np.stack([_splitmix64(ctr * np.uint64(4) + np.uint64(k)) for k in range(4)], axis=1)
            w[:, 3] &= top_mask
            lt = np.zeros(todo.size, dtype=bool)
            decided = np.zeros(todo.size, dtype=bool)
            for k in (3, 2, 1, 0):
                ne = (w[:, k] != p[k]) & ~decided
                lt |= ne & (w[:, k] < p[k])
                decided |= ne
            out[todo[lt].astype(np.int64)] = w[lt]
            todo = todo[~lt]
            t += 1
            assert t < 256
    return out


def scalars_to_ints(words: np.ndarray):
    return [int(a) | (int(b) << 64) | (int(c) << 128) | (int(d) << 192) for a, b, c, d in words.tolist()]


def closed_form_scalar(words: np.ndarray, t0: int, d: int, r: int) -> int:
    """sum_i a_i (t0 + i d) mod r with exact integer column sums (16-bit pieces x 12-bit index halves)."""
    n = words.shape[0]
    idx = np.arange(n, dtype=np.uint64)
    ilo, ihi = idx & np.uint64(0xFFF), idx >> np.uint64(12)
    assert n <= 1 << 24 or int(ihi.max()) < 1 << 20
    s_a = 0
    s_ia = 0
    for k in range(4):
        col = words[:, k]
        for h in range(4):
            piece = (col >> np.uint64(16 * h)) & np.uint64(0xFFFF)
            sh = 64 * k + 16 * h
            s_a += int(piece.sum(dtype=np.uint64)) << sh
            s_ia += (int((piece * ilo).sum(dtype=np.uint64)) + (int((piece * ihi).sum(dtype=np.uint64)) << 12)) << sh
    return (t0 * s_a + d * s_ia) % r
