import sqlite3, re, glob, sys
f=(glob.glob(sys.argv[1]+'/*/*.db')+glob.glob(sys.argv[1]+'/*.db'))[0]
db=sqlite3.connect(f)
rows=db.execute("select name,start,end,stream_id,queue_id,grid_x,lds_size from kernels order by start").fetchall()
def short(n):
    m=re.search(r'hipk::(\w+)',n); return m.group(1) if m else n[:20]
idx=[i for i,r in enumerate(rows) if 'elimFactor' in r[0]]
i0=idx[min(4, len(idx)-2)]; i1=idx[min(5, len(idx)-1)]
seg=rows[i0:i1]
t0=seg[0][1]
# dense phase: after elimGather ends
tg=[r for r in seg if 'elimGather' in r[0]][-1][2]
tend=max(r[2] for r in seg)
print('iteration ms', (tend-t0)/1e6, 'dense ms', (tend-tg)/1e6)
streams=sorted(set(r[3] for r in seg))
for s in streams:
    ks=[r for r in seg if r[3]==s and r[1]>=tg]
    busy=sum(r[2]-r[1] for r in ks)
    print('stream',s,'kernels',len(ks),'busy ms',busy/1e6,'frac',busy/(tend-tg))
# gaps on main stream (largest)
main=[r for r in seg if r[3]==streams[0] and r[1]>=tg]
gaps=[(main[i+1][1]-main[i][2], short(main[i][0]), short(main[i+1][0]), (main[i][2]-t0)/1e3) for i in range(len(main)-1)]
import collections
agg=collections.defaultdict(lambda:[0,0.0])
for g in gaps: agg[(g[1],g[2])][0]+=1; agg[(g[1],g[2])][1]+=g[0]/1e3
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): print('gap',k,v[0],'total us',round(v[1],1),'avg',round(v[1]/v[0],2))
agg=collections.defaultdict(lambda:[0,0.0])
for r in seg:
    k=(short(r[0]), r[3]); agg[k][0]+=1; agg[k][1]+=(r[2]-r[1])/1e3
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): print(k, v[0], round(v[1],1), 'avg', round(v[1]/v[0],1))
a,b=int(sys.argv[2]),int(sys.argv[3])
for r in seg[a:b]:
    print("%-22s s%d grid%-8d start %9.1f dur %7.1f"%(short(r[0]),r[3],r[5],(r[1]-t0)/1e3,(r[2]-r[1])/1e3))
# per outer block: time of the block-last trsm launch, the wait before it, and the block period
prev=None
print('block  t_us   wait_before_trsmPlus  period_us  side_busy_in_period')
side=[r for r in seg if r[3]!=streams[0]]
bi=0
for i,r in enumerate(main):
    if 'trsmPanelDirectPlus' in r[0]:
        wait=(r[1]-main[i-1][2])/1e3
        t=(r[1]-t0)/1e3
        if prev is not None:
            sb=sum(min(x[2],r[1])-max(x[1],prev) for x in side if x[2]>prev and x[1]<r[1])/1e3
            print('%3d %8.1f %8.1f %10.1f %8.1f'%(bi,t,wait,(r[1]-prev)/1e3,sb))
        prev=r[1]; bi+=1
