#!/usr/bin/env python3
"""profiles/rN_pmc_ntt.json from the transform PMC passes of tools/final_measure.sh (CSV output of rocprofv3):
    python tools/pmc_ntt_summary.py gpurun_out/final > profiles/r2_pmc_ntt.json
Per transform size: FETCH_SIZE and WRITE_SIZE of pass A and pass B, mean over the dispatches of the run (all four
variants), in bytes (counter unit: KB = 1024 B as in tools/pmc_summary.py).  Pass B reads rows as wide coalesced streams,
for which FETCH_SIZE reports one half (MI355X_MICROARCH.md, HBM section): fetch_corrected doubles it.  Pass A reads
columns in runs of 32-128 bytes, a pattern the guide does not calibrate: its raw value is given as it is."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for lg in (19, 22):
    per = defaultdict(lambda: defaultdict(list))
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in glob.glob("%s/pmc_ntt%d_%s/**/*_counter_collection.csv" % (root, lg, kind), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                name = r["Kernel_Name"]
                if "k_ntt_pass_a" in name:
                    per["pass_a"][counter].append(float(r["Counter_Value"]) * 1024.0)
                elif "k_ntt_pass_b" in name:
                    per["pass_b"][counter].append(float(r["Counter_Value"]) * 1024.0)
    if not per:
        continue
    m = 1 << lg
    mean = lambda v: sum(v) / len(v) if v else None  # noqa: E731
    a_f, a_w = mean(per["pass_a"]["FETCH_SIZE"]), mean(per["pass_a"]["WRITE_SIZE"])
    b_f, b_w = mean(per["pass_b"]["FETCH_SIZE"]), mean(per["pass_b"]["WRITE_SIZE"])
    e = {"algorithmic_bytes": 128 * m,
         "pass_a": {"FETCH_SIZE_bytes": a_f, "WRITE_SIZE_bytes": a_w, "dispatches": len(per["pass_a"]["FETCH_SIZE"])},
         "pass_b": {"FETCH_SIZE_bytes": b_f, "fetch_corrected_bytes": 2 * b_f if b_f else None, "WRITE_SIZE_bytes": b_w,
                    "dispatches": len(per["pass_b"]["FETCH_SIZE"])}}
    if None not in (a_f, a_w, b_f, b_w):
        e["bytes_per_transform"] = a_f + a_w + 2 * b_f + b_w
        e["over_algorithmic"] = e["bytes_per_transform"] / (128.0 * m)
        # pass A's raw FETCH_SIZE is also about half of the bytes it reads: upper estimate with the same factor of two
        e["bytes_per_transform_upper"] = 2 * a_f + a_w + 2 * b_f + b_w
        e["over_algorithmic_upper"] = e["bytes_per_transform_upper"] / (128.0 * m)
    e["note"] = ("two passes, each reads and writes the vector once: 128 m bytes; pass A also reads its twiddle and coset tables "
                 "(2 m/N2-entry and N2-entry tables, L2-resident) and, in runs shorter than a 128-byte line, shares lines with "
                 "neighbouring tiles")
    out[str(lg)] = e
print(json.dumps(out, indent=1))
