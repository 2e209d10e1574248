cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3_base_tests.txt
python bench.py --no-extras --steps 20 --warmup 5 > gpurun_out/r3_base_bench.json 2> gpurun_out/r3_base_bench.err
cat gpurun_out/r3_base_tests.txt; python -c "
import json; d=json.load(open('gpurun_out/r3_base_bench.json')); print(d['ms_per_step'], d['kernel_ms'], d['kernel_ms_isolated'], d['cpu_baseline'])"
