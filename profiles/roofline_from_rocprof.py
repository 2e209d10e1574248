"""rocprofv3 kernel trace of `python bench.py` -> (1) the per-kernel stats table of the run, (2) the
same table restricted to the headline workload's factor() calls (the default bench line also runs
the extra workloads -- 64 x GRID batched, BAL-1723 fp32 -- whose launches carry the same kernel
names), (3) profiles/rocprof_roofline.json: per kernel class of the headline workload, average
launch duration and the roofline fraction recomputed from it with the algorithmic work bench.py
reports (kernel_rates[*].algorithmic_flops / algorithmic_bytes).
usage: python profiles/roofline_from_rocprof.py <rocprof dir or .db> <bench.json> <tag>
The headline factor() calls are recognised by their first kernel: elimFactorTiny with the largest
grid of the run; a call ends where the next one begins (the last one: at the last chain kernel)."""
import glob
import json
import os
import re
import sqlite3
import sys

src, bench_json, tag = sys.argv[1:4]
f = src if src.endswith(".db") else (glob.glob(src + '/*/*.db') + glob.glob(src + '/*.db'))[0]
db = sqlite3.connect(f)
rows = db.execute("select name,start,end,stream_id,grid_x from kernels order by start").fetchall()
here = os.path.dirname(os.path.abspath(__file__))
short = lambda n: (re.search(r'hipk::(\w+)', n) or [None, n[:40]])[1]


def table(rs, title):
    agg = {}
    for n, s, e, _, _ in rs:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values()) or 1
    out = ["# " + title, "%-28s %7s %14s %11s %10s %10s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append("%-28s %7d %14d %11.0f %10d %10d %6.2f%%" % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))
    return "\n".join(out) + "\n", agg


full, _ = table(rows, "all kernels of the run")
heads = [i for i, r in enumerate(rows) if 'elimFactorTiny' in r[0]]
gmax = max(rows[i][4] for i in heads)
heads = [i for i in heads if rows[i][4] == gmax]
bench = json.load(open(bench_json))
# the warm-up and timed steps only: the two event-profiled factors that follow (bench.py's
# kernel_ms / kernel_ms_isolated) carry event records between the launches, one of them serialised
heads = heads[:bench["steps"] + bench["warmup"]] + heads[bench["steps"] + bench["warmup"]:][:0]
stop_at = None
# a headline factor() = from its elimFactorTiny to the kernel before the next elimFactor* / foreign
# kernel; solve kernels and copies in between are excluded by name
FACTOR = ("elimFactor", "elimGather", "chainStep", "updateTile", "potrfPanel", "trsmPanel", "tailFactor")
sel = []
all_heads = [i for i, r in enumerate(rows) if 'elimFactor' in r[0]]
for k, i in enumerate(heads):
    j = i
    later = [h for h in all_heads if h > i]
    nxt = later[0] if later else len(rows)
    while j < nxt:
        n = rows[j][0]
        if j > i and 'elimFactor' in n:
            break
        if any(t in n for t in FACTOR):
            sel.append(rows[j])
        elif 'solve' in n or 'addMv' in n:
            break
        j += 1
ncalls = len(heads)
head_txt, agg = table(sel, "kernels of the %d headline factor() calls only" % ncalls)
open(os.path.join(here, "%s_kernel_stats.txt" % tag), "w").write(full + "\n" + head_txt)
print(head_txt)

CLASS = {"update": ["updateTileBulk", "updateTile"], "chain_update": ["chainStep", "tailFactor", "updateTileDirectPotrf", "updateTileDirect"],
         "elim_update": ["elimGatherMfma", "elimGather", "elimGatherTiny", "elimUpdate"],
         "elim_factor": ["elimFactorTiny", "elimFactorSmall"], "trsm": ["trsmPanel", "trsmPanelPotrf", "trsmPanelDirect", "trsmPanelDirectPlus"],
         "potrf": ["potrfPanel", "potrfPanelDirect"]}
out = {}
for cls, ent in bench.get("kernel_rates", {}).items():
    ks = [k for k in CLASS[cls] if k in agg]
    if not ks:
        continue
    tot_ns = sum(agg[k][1] for k in ks) / ncalls     # per factor()
    launches = sum(agg[k][0] for k in ks) / ncalls
    work = ent.get("algorithmic_flops", ent.get("algorithmic_bytes"))
    mfma = "algorithmic_flops" in ent
    rate = work / (tot_ns * 1e-9) / (1e12 if mfma else 1e9)
    peak = 78.6 if mfma else 8000.0
    # launches of a class on different streams overlap: time with at least one of them running
    ev = sorted((r[1], r[2]) for r in sel if short(r[0]) in ks)
    union, end = 0, -1
    for a, b in ev:
        if b > end:
            union += b - max(a, end)
            end = b
    busy_ns = union / ncalls
    out[cls] = {"kernels": ks, "launches_per_factor": launches, "total_us_per_factor": round(tot_ns / 1e3, 1),
                "avg_ns": round(tot_ns / launches, 0), "rate": round(rate, 3), "unit": "TFLOP/s" if mfma else "GB/s",
                "frac": round(rate / peak, 4), "busy_us_per_factor": round(busy_ns / 1e3, 1),
                "frac_busy": round(work / (busy_ns * 1e-9) / (1e12 if mfma else 1e9) / peak, 4),
                "stats_file": "profiles/%s_kernel_stats.txt" % tag}
path = os.path.join(here, "rocprof_roofline.json")
try:
    doc = json.load(open(path))
except (OSError, ValueError):
    doc = {}
doc[bench["config"]["workload"]] = out
sys.path.insert(0, os.path.dirname(here))
from baspacho_amd import _lib  # noqa: E402
doc["_kernel_source_sha16"] = _lib.kernel_source_sha16()   # bench.py ignores the file when this is stale
doc["_how"] = ("frac = algorithmic work of the class (bench.py kernel_rates) / (calls x avg_ns of its kernels in the "
               "headline-only table of the stats file) / peak (78.6 TFLOP/s fp64 MFMA, 8000 GB/s HBM); frac_busy = the same "
               "over the time during which at least one launch of the class is running (its launches overlap when "
               "they run on two auxiliary streams)")
json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
