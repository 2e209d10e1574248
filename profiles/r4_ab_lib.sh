# A/B of two builds of the library on the headline workload, alternating runs (same box)
# usage: bash profiles/r4_ab_lib.sh TAG LIB_B [extra bench args]
mkdir -p gpurun_out
TAG=$1; LIBB=$2; shift; shift
for rep in 1 2 3; do
  for v in a b; do
    if [ $v = b ]; then export BSP_LIB_PATH=$PWD/$LIBB; else unset BSP_LIB_PATH; fi
    timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/ab_${TAG}_$v$rep.json 2> gpurun_out/ab_${TAG}_$v$rep.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/ab_${TAG}_$v$rep.json').read().strip().splitlines()[-1])
print('$v$rep', d['ms_per_step'], d['residual_probe'], {k: v[0] for k, v in d['kernel_ms'].items()}, {k: v[0] for k, v in d['kernel_ms_isolated'].items()})
PY
  done
done
