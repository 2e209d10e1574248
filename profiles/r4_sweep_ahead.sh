mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_factor_gpu.py tests/test_stress_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r4_t3.txt
for a in 0 1 2 4 100; do
  BSP_BULK_AHEAD=$a timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r4_b3_a$a.json 2> gpurun_out/r4_b3_a$a.err
done
tail -3 gpurun_out/r4_t3.txt
for a in 0 1 2 4 100; do python - <<PY
import json
d=json.loads(open('gpurun_out/r4_b3_a$a.json').read().strip().splitlines()[-1])
print('$a', d['ms_per_step'], d['residual_probe'], d['kernel_ms'], d['kernel_ms_isolated'])
PY
done
