mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_factor_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r4_t4.txt
tail -3 gpurun_out/r4_t4.txt
for g in 1 2 3 4 6; do
  BSP_DL_GROUP=$g timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r4_b4_g$g.json 2> gpurun_out/r4_b4_g$g.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_b4_g$g.json').read().strip().splitlines()[-1])
print('group $g', d['ms_per_step'], d['residual_probe'], d['kernel_ms'], d['kernel_ms_isolated'])
PY
done
bash profiles/r4_trace.sh g2 BSP_DL_GROUP=2 > /dev/null 2>&1
bash profiles/r4_trace.sh g4 BSP_DL_GROUP=4 > /dev/null 2>&1
