"""HBM bytes per launch from two rocprofv3 counter passes (rocpd sqlite):
    python tools/pmc_summary.py FETCH_DB WRITE_DB [CALIB_DB [DIGITS]] > profiles/rN_pmc_hbm_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in KB.  The calibration database (tools/calib_gather.hip under --pmc FETCH_SIZE)
says how FETCH_SIZE relates to bytes for the two access patterns here: exact for random 64-byte gathers, one half for
wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section)."""
import json
import re
import sqlite3
import sys

N, ENTRY_BYTES = 1 << 24, 68  # k_accum's algorithmic bytes per (point, digit): a 64 B affine base + a 4 B sorted entry
DIGITS = int(sys.argv[4]) if len(sys.argv) > 4 else 16  # digits per scalar: 16 windowed, 12 with the fixed-base table


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name"
    for name, calls, avg in db.execute(q, (counter,)):
        short = re.sub(r"^void ", "", re.sub(r"zg::", "", name))
        short = re.sub(r"[<(].*", "", short)
        out[short] = (calls, avg * 1024.0)
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around `python3 bench.py "
    "--no-cpu-baseline --no-prove --steps 3 --warmup 1` (BN254 G1 MSM, 2^24 points); values are per launch, in bytes "
    "(counter unit: KB).",
    "kernels": {},
}
if len(sys.argv) > 3:
    cal = per_kernel(sys.argv[3], "FETCH_SIZE")
    res["calibration"] = {k: {"expected_bytes": 1 << 30, "FETCH_SIZE_bytes": v[1]} for k, v in cal.items()}
    res["calibration"]["conclusion"] = (
        "FETCH_SIZE is exact for k_accum's pattern (one random 64-byte record per lane, four 16-byte loads) and reports "
        "one half of a wide coalesced streaming read: streaming kernels carry fetch_corrected = 2 x raw, k_accum and the "
        "sort's gathers raw."
    )
for k in sorted(set(fetch) | set(write)):
    e = {}
    if k in fetch:
        e["FETCH_SIZE_bytes_per_launch"] = fetch[k][1]
        e["launches"] = fetch[k][0]
    if k in write:
        e["WRITE_SIZE_bytes_per_launch"] = write[k][1]
        e.setdefault("launches", write[k][0])
    res["kernels"][k] = e
acc = res["kernels"].get("k_accum")
if acc:
    alg = N * DIGITS * ENTRY_BYTES
    acc["digits_per_scalar"] = DIGITS
    acc["algorithmic_bytes_per_launch"] = alg
    acc["traffic_over_algorithmic"] = (acc["FETCH_SIZE_bytes_per_launch"] + acc.get("WRITE_SIZE_bytes_per_launch", 0)) / alg
print(json.dumps(res, indent=1))
