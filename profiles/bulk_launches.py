"""Per side-stream (bulk) launch of one factor: grid size, duration, tiles per microsecond, and how
the launches line up with the chain (rocprofv3 --kernel-trace output directory as argument)."""
import glob
import re
import sqlite3
import sys

f = (glob.glob(sys.argv[1] + '/*/*.db') + glob.glob(sys.argv[1] + '/*.db'))[0]
db = sqlite3.connect(f)
rows = db.execute("select name,start,end,stream_id,grid_x from kernels order by start").fetchall()
short = lambda n: (re.search(r'hipk::(\w+)', n) or [None, n[:20]])[1]
idx = [i for i, r in enumerate(rows) if 'elimFactor' in r[0]]
i0, i1 = idx[-2], idx[-1]
seg = rows[i0:i1]
t0 = seg[0][1]
bulk = [r for r in seg if 'updateTileBulk' in r[0]]
chain = [r for r in seg if 'chainStep' in r[0]]
print("bulk launches %d, total %.1f us; chain launches %d, total %.1f us" % (
    len(bulk), sum(r[2] - r[1] for r in bulk) / 1e3, len(chain), sum(r[2] - r[1] for r in chain) / 1e3))
print("  start_us   dur_us   tiles  tiles/768  us_per_round  gap_before_us")
prev = None
for r in bulk:
    tiles = r[4] // 256
    rounds = tiles / 768.0
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    print("%9.1f %8.1f %7d %9.2f %12.1f %10.1f" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, tiles, rounds,
                                                 (r[2] - r[1]) / 1e3 / max(rounds, 1e-9), gap))
    prev = r[2]
tot_tiles = sum(r[4] // 256 for r in bulk)
print("tiles %d, mean us per 768 tiles %.1f" % (tot_tiles, sum(r[2] - r[1] for r in bulk) / 1e3 / (tot_tiles / 768.0)))
