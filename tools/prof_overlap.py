#!/usr/bin/env python3
"""Device occupancy over time from a rocprofv3 kernel trace (rocpd sqlite): python tools/prof_overlap.py DB [T0_FRAC T1_FRAC]
Over the chosen part of the trace (default: the last 40 %): wall time, time with at least one kernel running, mean number
of kernels running, and the kernels by summed duration."""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
lo, hi = rows[0][1], max(r[2] for r in rows)
a, b = lo + (hi - lo) * f0, lo + (hi - lo) * f1
sel = [(n, max(s, a), min(e, b)) for n, s, e in rows if e > a and s < b]
ev = []
for n, s, e in sel:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = 0.0
area = 0.0
depth = 0
prev = a
for t, d in ev:
    if depth > 0:
        busy += t - prev
    area += depth * (t - prev)
    depth += d
    prev = t
wall = b - a
print("window %.3f ms: busy %.3f ms (%.1f %%), mean kernels running %.2f, kernels %d" % (wall / 1e6, busy / 1e6, 100 * busy / wall, area / wall, len(sel)))
agg = defaultdict(float)
for n, s, e in sel:
    short = re.sub(r"[<(].*", "", re.sub(r"^void ", "", n.replace("zg::", "")))
    if "Fp2" in n:
        short += " Fq2"
    agg[short] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-26s %8.3f ms  (%.2f of the window)" % (k, v / 1e6, v / wall))
