cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile"
for cfg in "0 8" "1 8" "2 2" "2 4" "2 16"; do
  set -- $cfg
  export BSP_GATHER_ROWS=$1 BSP_GATHER_GROUP=$2
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d gpurun_out/gx_$1_$2 -o r -- $CMD > gpurun_out/gx_$1_$2.log 2>&1
  python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/gx_$1_$2/*.db')[0])
rows=db.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%elimGatherMfma%' group by counter_name").fetchall() if True else []
cur=db.execute("select * from counters_collection limit 1"); cols=[d[0] for d in cur.description]
nc='kernel_name' if 'kernel_name' in cols else 'name'
rows=db.execute("select counter_name, sum(value) from counters_collection where %s like '%%elimGatherMfma%%' group by counter_name"%nc).fetchall()
t=db.execute("select avg(end-start) from kernels where name like '%elimGatherMfma%'").fetchone()[0]
d=dict(rows)
print("mode $1 group $2: time %.3f ms, hit %.3g miss %.3g (hit rate %.1f%%), rdreq %.3g"%(t/1e6, d.get('TCC_HIT_sum',0)/2, d.get('TCC_MISS_sum',0)/2, 100*d.get('TCC_HIT_sum',0)/max(1,d.get('TCC_HIT_sum',0)+d.get('TCC_MISS_sum',0)), d.get('TCC_EA0_RDREQ_sum',0)/2))
PY
done
