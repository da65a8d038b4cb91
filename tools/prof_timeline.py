"""Timeline of the last proof in a rocprofv3 kernel trace (rocpd sqlite): python tools/prof_timeline.py DB [GAP_US]
Kernels are grouped by queue; times are relative to the first kernel after the last idle gap of GAP_US or more."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
# last idle gap
busy_end = 0
cut = 0
for i, r in enumerate(rows):
    if busy_end and r[1] - busy_end > gap_us * 1e3:
        cut = i
    busy_end = max(busy_end, r[2])
sel = rows[cut:]
t0 = sel[0][1]
print("columns:", cols)
print("kernels in the last burst: %d, span %.3f ms, summed kernel time %.3f ms" % (len(sel), (max(r[2] for r in sel) - t0) / 1e6, sum(r[2] - r[1] for r in sel) / 1e6))
for r in sel:
    short = re.sub(r"[<(].*", "", re.sub(r"^void ", "", re.sub(r"zg::", "", r[0])))[:30]
    typ = "Fq2" if "Fp2" in r[0] else ""
    print("%8.1f us  +%7.1f us  q=%-4s %s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", short, typ))
