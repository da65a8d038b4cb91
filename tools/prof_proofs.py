"""Per-proof summary of a rocprofv3 kernel trace (rocpd sqlite) of tools/bench_prove.py: python tools/prof_proofs.py DB
A proof starts at its k_build_w.  For every proof: the device span, when each queue's last kernel ended, when the H
chain's transforms ended, and the start/end of every accumulation kernel -- to tell fast proofs from slow ones."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = db.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()


def short(n):
    s = re.sub(r"[<(].*", "", re.sub(r"^void ", "", n.replace("zg::", "")))
    return s + ("/Fq2" if "Fp2" in n else "")


starts = [i for i, r in enumerate(rows) if "k_build_w" in r[0]]
for pi, s in enumerate(starts):
    e = starts[pi + 1] if pi + 1 < len(starts) else len(rows)
    sel = rows[s:e]
    t0 = sel[0][1]
    span = (max(r[2] for r in sel) - t0) / 1e3
    qend = {}
    for r in sel:
        qend[r[3]] = max(qend.get(r[3], 0), (r[2] - t0) / 1e3)
    ntt_end = max([(r[2] - t0) / 1e3 for r in sel if "k_ntt" in r[0]] or [0])
    acc = ["%s q%s %.0f-%.0f" % (short(r[0]), r[3], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3) for r in sel if "k_accum" in r[0]]
    red = ["%s q%s %.0f-%.0f" % (short(r[0]), r[3], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3) for r in sel if "k_reduce1" in r[0]]
    print("proof %2d  span %7.1f us  transforms end %7.1f  queue ends %s" % (pi, span, ntt_end, " ".join("q%s:%.0f" % (q, v) for q, v in sorted(qend.items()))))
    print("          accum  ", "; ".join(acc))
    print("          reduce1", "; ".join(red))
