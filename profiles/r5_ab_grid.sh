# A/B on the GRID workloads (single / batch 8 / batch 64), alternating: bash profiles/r5_ab_grid.sh TAG "ENV_A" "ENV_B" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    line="$cfg rep $rep:"
    for b in 0 8 64; do
      a=""; [ $b -gt 0 ] && a="--batch $b"
      ms=$(env $cfg timeout 300 python bench.py --workload grid82 $a --no-cpu-baseline --no-extras --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f (%.0e)' % (d['ms_per_step'], d.get('residual_probe') or -1))")
      line="$line  b$b $ms"
    done
    echo "$line"
  done
done
