"""Per-kernel totals from a rocprofv3 results database (rocpd sqlite): python tools/prof_summary.py DB [DIVISOR]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel time / %g: %.3f ms" % (div, tot / 1e6 / div))
print("%-34s %-4s %9s %12s %10s %10s %10s" % ("kernel", "", "calls", "total ms", "avg us", "min us", "max us"))
for n, c, s, a, mn, mx in rows:
    short = re.sub(r"zg::", "", n)
    short = re.sub(r"^void ", "", short)
    short = re.sub(r"[<(].*", "", short)[:34]
    typ = "Fq2" if "Fp2" in n else ("bls" if "Bls377" in n else "")
    print("%-34s %-4s %9.1f %12.3f %10.1f %10.1f %10.1f" % (short, typ, c / div, s / 1e6 / div, a / 1e3, mn / 1e3, mx / 1e3))
