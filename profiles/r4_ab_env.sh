# A/B of one environment switch on the headline workload, alternating runs (same box)
# usage: bash profiles/r4_ab_env.sh TAG NAME VALUE_A VALUE_B
mkdir -p gpurun_out
TAG=$1; NAME=$2; VA=$3; VB=$4
for rep in 1 2 3; do
  for v in a b; do
    if [ $v = a ]; then export $NAME=$VA; else export $NAME=$VB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/ab_${TAG}_$v$rep.json 2> gpurun_out/ab_${TAG}_$v$rep.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/ab_${TAG}_$v$rep.json').read().strip().splitlines()[-1])
print('$NAME=' + ('$VA' if '$v' == 'a' else '$VB'), d['ms_per_step'], d['residual_probe'], {k: v[0] for k, v in d['kernel_ms'].items()})
PY
  done
done
