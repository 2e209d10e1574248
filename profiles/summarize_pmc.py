"""Per-kernel sums of the PMC counters in rocprofv3 (rocpd sqlite) outputs.
usage: python profiles/summarize_pmc.py <results.db> [<results.db> ...]"""
import re
import sqlite3
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute("select %s, counter_name, sum(value), count(distinct dispatch_id) "
                      "from counters_collection group by %s, counter_name" % (name_col, name_col)).fetchall()
    agg = defaultdict(dict)
    calls = {}
    for kname, cname, val, n in rows:
        m = re.search(r"hipk::(\w+)", kname)
        k = m.group(1) if m else kname[:40]
        agg[k][cname] = agg[k].get(cname, 0) + val
        calls[k] = max(calls.get(k, 0), n)
    print("==", path)
    for k in sorted(agg, key=lambda k: -max(agg[k].values())):
        print("  %-22s calls %5d  " % (k, calls[k]) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(agg[k].items())))
