"""Per-launch HBM-side traffic of every kernel from the FETCH_SIZE / WRITE_SIZE passes of
profiles/collect_pmc.sh -> profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
usage: python profiles/make_pmc_traffic.py <workload> <fetch.db> <write.db> <factor calls in the run> <label>
Totals are stored per factor() call: the launch count of a kernel class differs between the
lookahead schedule (profiled by rocprofv3) and bench.py's event-timed sequential schedule."""
import json
import os
import re
import sqlite3
import sys

workload, fetch_db, write_db, nfactor, label = sys.argv[1:6]
nfactor = float(nfactor)


def per_launch(path, counter):
    db = sqlite3.connect(path)
    cur = db.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for kname, total, n in db.execute(
            "select %s, sum(value), count(distinct dispatch_id) from counters_collection "
            "where counter_name = ? group by %s" % (name_col, name_col), (counter,)):
        m = re.search(r"hipk::(\w+)", kname)
        if m:
            out[m.group(1)] = (total / nfactor, n / nfactor)
    return out


f = per_launch(fetch_db, "FETCH_SIZE")
w = per_launch(write_db, "WRITE_SIZE")
here = os.path.dirname(os.path.abspath(__file__))
path = os.path.join(here, "pmc_traffic.json")
try:
    doc = json.load(open(path))
except (OSError, ValueError):
    doc = {}
doc["_source"] = label
sys.path.insert(0, os.path.dirname(here))
from baspacho_amd import _lib  # noqa: E402
doc["_kernel_source_sha16"] = _lib.kernel_source_sha16()   # bench.py ignores the file when this is stale
doc["_note"] = ("KB per factor() call as rocprofv3 reports them; bench.py doubles FETCH_SIZE (gfx950 "
                "tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE is "
                "uncalibrated; Infinity-Cache hits are included")
doc[workload] = {k: {"fetch_KB_per_factor": round(f[k][0], 1),
                     "write_KB_per_factor": round(w.get(k, (0.0, 0))[0], 1),
                     "launches_per_factor": f[k][1]} for k in sorted(f)}
json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(doc[workload], indent=1))
