"""launch sequence of the last factor() in a rocprofv3 --kernel-trace database: python profiles/kseq.py DB FIRST_KERNEL_SUBSTRING
-> per launch: start offset, duration, gap to the previous end (same queue order), kernel, grid"""
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
first = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
# the last factor() starts at the last occurrence of the marker kernel that follows a long pause
starts = [i for i in first if i == 0 or rows[i][1] - rows[i - 1][2] > 200000]
rows = rows[(starts[-1] if starts else first[-1]):]
t0, prev = rows[0][1], rows[0][1]
tot = {}
for name, st, en, x, y, w in rows:
    short = name.split("hipk")[-1][:28] if "hipk" in name else name[:28]
    if "Cijk" in name or "rocclr" in name or "elementwise" in name:
        continue
    print("%9.1f %8.1f %7.1f  %-28s %6d x %d" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3, short, x // w, y))
    prev = max(prev, en)
    k = tot.setdefault(short, [0, 0.0])
    k[0] += 1
    k[1] += (en - st) / 1e3
print("total span %.1f us" % ((prev - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s %4d launches %9.1f us" % (k, v[0], v[1]))
