#!/usr/bin/env python3
"""Digit width of the fixed-base table by number of bases: BN254 G1, 2^lo..2^hi points, widths a..b.  One JSON line per case
(tools/bench_sweep.py msm_case); the choice is msm_table_pick_c (msm.cuh)."""
import sys

import bench_sweep as B
import zeth_amd

lo, hi, ca, cb = (int(x) for x in sys.argv[1:5])
ctx = zeth_amd.Context(0)
ctx.set_profiling(True)
for lg in range(lo, hi + 1):
    for c in range(ca, cb + 1):
        B.msm_case(ctx, 0, 1, lg, reps=5, check=False, table=c)
