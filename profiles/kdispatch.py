"""per-dispatch list of one kernel in a rocprofv3 --kernel-trace database, last factor() call:
python profiles/kdispatch.py DB NAME_SUBSTRING [last_n]  ->  start offset, duration, grid, workgroup"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
gx = [c for c in cols if c.lower() in ("grid_size_x", "grid_x")][0]
gy = [c for c in cols if c.lower() in ("grid_size_y", "grid_y")][0]
wx = [c for c in cols if c.lower() in ("workgroup_size_x", "workgroup_x")][0]
rows = cur.execute(
    f"select s.kernel_name, d.start, d.end, d.{gx}, d.{gy}, d.{wx} from {kd} d join {ks} s "
    f"on d.kernel_id=s.id order by d.start").fetchall()
sel = [r for r in rows if sys.argv[2] in r[0]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(sel)
sel = sel[-n:]
t0 = sel[0][1]
print("%10s %10s %10s %8s %8s" % ("start us", "dur us", "wgs x", "y", "us/kwg"))
for name, st, en, x, y, w in sel:
    wgs = x // w
    print("%10.1f %10.1f %10d %8d %8.2f" % ((st - t0) / 1e3, (en - st) / 1e3, wgs, y, (en - st) / 1e3 / max(wgs * y, 1) * 1e3))
