#!/usr/bin/env python3
"""VALU wave-instructions per kernel of ONE proof: python tools/pmc_valu_by_kernel.py DIR
DIR holds the *_counter_collection.csv of
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -- python3 tools/bench_prove.py ...
One proof = the dispatches from the second-last k_build_w up to the last one."""
import csv
import glob
import re
import sys
from collections import defaultdict

import os

# the newest run only: a directory that was profiled into twice holds two files whose dispatch ids collide
files = sorted(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
disp = {}
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
bw = [i for i in ids if "k_build_w" in disp[i]["name"]]
# one proof's worth of dispatches whatever the order in which the lanes' threads queued theirs: from the second-last
# k_build_w up to the last one
sel = [i for i in ids if bw[-2] <= i < bw[-1]]


def short(n):
    n = re.sub(r"^void ", "", n).replace("zg::", "")
    base = re.sub(r"[<(].*", "", n)
    return base + (" Fq2" if "Fp2" in n else "")


acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i in sel:
    d = disp[i]
    a = acc[short(d["name"])]
    a[0] += 1
    a[1] += d.get("SQ_INSTS_VALU", 0.0)
    a[2] += (d["t1"] - d["t0"]) / 1e3
    a[3] += d.get("SQ_WAVES", 0.0)
tot = sum(a[1] for a in acc.values())
print("total VALU wave-instructions per proof: %.1fM  (at 4 cycles each on 1024 SIMDs at 2.4 GHz: %.2f ms)" % (tot / 1e6, tot * 4 / 1024 / 2.4e9 * 1e3))
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-28s launches %2d  VALU %8.2fM (%4.1f%%)  duration when serialised %7.1f us  waves %d" % (k, a[0], a[1] / 1e6, 100 * a[1] / tot, a[2], a[3]))
