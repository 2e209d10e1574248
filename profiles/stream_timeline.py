"""Per-stream view of one factor() from a rocprofv3 --kernel-trace run: busy time of every stream
in the dense phase, the main stream's gaps (by size), and the bulk launches of each side stream."""
import glob
import re
import sqlite3
import sys

f = (glob.glob(sys.argv[1] + '/*/*.db') + glob.glob(sys.argv[1] + '/*.db'))[0]
db = sqlite3.connect(f)
rows = db.execute("select name,start,end,stream_id,grid_x from kernels order by start").fetchall()
short = lambda n: (re.search(r'hipk::(\w+)', n) or [None, n[:20]])[1]
# the headline workload's factor() calls start with the elimFactorTiny launch of the largest grid;
# take the last one that is followed by another (a timed step, not the profiled ones at the end)
idx = [i for i, r in enumerate(rows) if 'elimFactor' in r[0]]
gmax = max(rows[i][4] for i in idx)
heads = [i for i in idx if rows[i][4] == gmax]
pick = int(sys.argv[2]) if len(sys.argv) > 2 else min(4, len(heads) - 2)
seg = [r for r in rows[heads[pick]:heads[pick + 1]]
       if any(t in r[0] for t in ("elimFactor", "elimGather", "chainStep", "updateTile", "potrfPanel", "trsmPanel", "tailFactor"))]
t0 = seg[0][1]
tg = [r for r in seg if 'elimGather' in r[0]][-1][2]
tend = max(r[2] for r in seg)
print("factor %.3f ms: elimination %.3f ms, dense phase %.3f ms" % ((tend - t0) / 1e6, (tg - t0) / 1e6, (tend - tg) / 1e6))
# launches of one kernel on different streams overlap: time during which at least one is running
for kname in ("updateTileBulk", "chainStep"):
    ev = sorted((r[1], r[2]) for r in seg if kname in r[0])
    if not ev:
        continue
    union, end = 0, -1
    for a, b in ev:
        if b <= end:
            continue
        union += b - max(a, end)
        end = b
    print("%s: %d launches, %.3f ms of launch time, %.3f ms with at least one running" % (
        kname, len(ev), sum(b - a for a, b in ev) / 1e6, union / 1e6))
streams = sorted(set(r[3] for r in seg))
main = None
for s in streams:
    ks = [r for r in seg if r[3] == s and r[1] >= tg]
    if not ks:
        continue
    names = sorted(set(short(r[0]) for r in ks))
    busy = sum(r[2] - r[1] for r in ks)
    print("stream %d: %3d kernels, busy %.3f ms (%.0f %% of the dense phase)  %s" % (
        s, len(ks), busy / 1e6, 100.0 * busy / (tend - tg), ",".join(names)[:70]))
    if any('chainStep' in r[0] for r in ks):
        main = ks
gaps = sorted(((main[i + 1][1] - main[i][2]) / 1e3, (main[i][2] - t0) / 1e3) for i in range(len(main) - 1))
tot = sum(g[0] for g in gaps)
print("main stream: %d gaps, %.1f us in all; <3 us: %d (%.1f us)  3-10 us: %d (%.1f us)  >10 us: %d (%.1f us)" % (
    len(gaps), tot,
    sum(1 for g in gaps if g[0] < 3), sum(g[0] for g in gaps if g[0] < 3),
    sum(1 for g in gaps if 3 <= g[0] < 10), sum(g[0] for g in gaps if 3 <= g[0] < 10),
    sum(1 for g in gaps if g[0] >= 10), sum(g[0] for g in gaps if g[0] >= 10)))
print("largest gaps (us @ time):", ["%.0f@%.0f" % g for g in gaps[-12:]])
d = [(r[2] - r[1]) / 1e3 for r in main if 'chainStep' in r[0]]
print("chainStep: %d launches, mean %.1f us, first 20 mean %.1f, last 20 mean %.1f" % (
    len(d), sum(d) / len(d), sum(d[:20]) / 20, sum(d[-20:]) / 20))
