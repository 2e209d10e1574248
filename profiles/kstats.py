"""per-kernel totals of a rocprofv3 --kernel-trace database: python profiles/kstats.py DB [calls]
(calls = number of factor() calls in the run, to print per-call figures)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
calls = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(
    f"select s.kernel_name, count(*), sum(d.end-d.start) from {kd} d join {ks} s "
    f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("%-44s %8s %12s %10s %6s" % ("kernel", "launches", "us per call", "avg us", "%"))
for name, n, t in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    short = name.split("hipk")[-1][:44] if "hipk" in name else name[:44]
    print("%-44s %8.1f %12.1f %10.1f %6.1f" % (short, n / calls, t / 1e3 / calls, t / 1e3 / n, 100.0 * t / tot))
