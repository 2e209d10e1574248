"""Per-stream timeline of ONE factor() call from a rocprofv3 kernel trace (rocpd sqlite):
usage: python profiles/timeline_rocpd.py <trace dir> <first kernel> <last kernel>  (0 0 = totals only)"""
import sqlite3, re, glob, sys
f=glob.glob(sys.argv[1]+'/*/*.db')[0]
db=sqlite3.connect(f)
rows=db.execute("select name,start,end,stream_id,queue_id,grid_x,lds_size from kernels order by start").fetchall()
def short(n):
    m=re.search(r'hipk::(\w+)',n); return m.group(1) if m else n[:20]
idx=[i for i,r in enumerate(rows) if 'elimFactor' in r[0]]
i0=idx[min(4, len(idx)-2)]; i1=idx[min(5, len(idx)-1)]
t0=rows[i0][1]
seg=rows[i0:i1]
print('iteration span ms', (max(r[2] for r in seg)-t0)/1e6)
a,b=int(sys.argv[2]),int(sys.argv[3])
for r in seg[a:b]:
    print("%-16s s%d grid%-8d lds%-6d start %9.1f dur %7.1f"%(short(r[0]),r[3],r[5],r[6],(r[1]-t0)/1e3,(r[2]-r[1])/1e3))
# aggregate by kernel name on main stream
import collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in seg:
    k=(short(r[0]), r[3]); agg[k][0]+=1; agg[k][1]+=(r[2]-r[1])/1e3
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): print(k, v[0], round(v[1],1), 'avg', round(v[1]/v[0],1))
