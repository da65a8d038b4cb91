#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 --pmc results: python tools/pmc_table.py DIR [NAME_FILTER]
Reads every *_counter_collection.csv under DIR; prints, per kernel name, the mean of each counter over its dispatches
(and the mean duration from the timestamps)."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"zg::", "", name)
        short = re.sub(r"\(.*", "", name)[:70]
        if flt and flt not in short:
            continue
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[short][(f, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k in sorted(acc):
    d = list(dur[k].values())
    print("%s   dispatches/pass %d   mean %.1f us" % (k, len(d), sum(d) / len(d)))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-28s %16.0f" % (c, sum(v) / len(v)))
