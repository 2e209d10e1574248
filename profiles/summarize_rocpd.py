"""Turn a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table.
usage: python profiles/summarize_rocpd.py <results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
    "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["%-90s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns",
                                               "max_ns", "pct")]
for name, calls, total, avg, mn, mx in rows:
    lines.append("%-90s %8d %14d %12.0f %12d %12d %6.2f%%" % (name[:90], calls, total, avg, mn, mx,
                                                            100.0 * total / tot))
text = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print(text)
