# batched GRID 82x82 at several batch sizes under environment variants: bash profiles/r5_ab_batch.sh "ENV_A" "ENV_B" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "$@"; do
    line="$cfg rep $rep:"
    for b in 16 32 64; do
      ms=$(env $cfg timeout 300 python bench.py --workload grid82 --batch $b --no-cpu-baseline --no-extras --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f (%.0e)' % (d['ms_per_step'], d.get('residual_probe') or -1))")
      line="$line  b$b $ms"
    done
    echo "$line"
  done
done
