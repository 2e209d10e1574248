# A/B of environment switches on one box, alternating runs: bash profiles/r5_ab_env.sh TAG "bench args" "ENV_A" "ENV_B" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ARGS=$2; shift; shift
for rep in 1 2 3; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout 300 python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab_${TAG}_${i}_$rep.json
    python - <<PY
import json
d=json.loads(open('gpurun_out/ab_${TAG}_${i}_$rep.json').read())
print('%-40s rep $rep  %.4f ms  probe %.1e ' % ('$cfg', d['ms_per_step'], d.get('residual_probe') or -1), {k: round(v[0],3) for k, v in d.get('kernel_ms', {}).items()})
PY
  done
done
