# generator sensitivity (verdict item 3b): bench + gather traffic on three BAL-871 stand-ins
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sens
for wl in bal871 bal871-clustered bal871-banded; do
  timeout 600 python bench.py --workload $wl --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/sens/$wl.json
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/sens_pmc
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/sens_pmc -o p -- python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /dev/null 2>&1
    python profiles/summarize_pmc.py gpurun_out/sens_pmc/p_results.db | grep -E "elimGather|elimFactor" | sed "s/^/$wl /"
  done > gpurun_out/sens/$wl.pmc.txt
  python - <<PY
import json
d=json.loads(open('gpurun_out/sens/$wl.json').read())
print('$wl', d['ms_per_step'], 'ms', d['value'], 'GF/s', d['config']['description'][-40:], {k: round(v[0],3) for k,v in d['kernel_ms'].items()}, 'probe', d['residual_probe'], 'solve1', d.get('solve1_ms'))
PY
  cat gpurun_out/sens/$wl.pmc.txt
done
rm -rf gpurun_out/sens_pmc
