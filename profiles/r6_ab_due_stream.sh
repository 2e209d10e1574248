cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --workload $1 --no-extras --no-cpu-baseline --no-profile --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for w in flat50k bal871; do
  echo "$w product:          $(run $w)"
  echo "$w BSP_DUE_STREAM=0: $(BSP_DUE_STREAM=0 run $w)"
done; done
