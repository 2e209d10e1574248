cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_driver_style.json; python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_driver_style.json').read()); print(d['ms_per_step'], d['value'], d['steps'], d['warmup'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
