# round 6, final sources: share of the next block's chain time handed out as OPTIONAL lookahead units (bulkAhead, product 0.8)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline --no-profile --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])"; }
for rep in 1 2; do for w in bal871 flat50k; do
  echo "$w  0.8 (product): $(run $w)  1.2: $(BSP_BULK_AHEAD=1.2 run $w)  1.6: $(BSP_BULK_AHEAD=1.6 run $w)  2.5: $(BSP_BULK_AHEAD=2.5 run $w)  0.5: $(BSP_BULK_AHEAD=0.5 run $w)"
done; done
