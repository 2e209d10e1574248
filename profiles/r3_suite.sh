cd $GRAFT_REPO_ROOT
python bench.py --suite ref --suite-out gpurun_out/r03_ref_suite.json > gpurun_out/r03_ref_suite.log 2> gpurun_out/r03_ref_suite.err
tail -1 gpurun_out/r03_ref_suite.log; tail -3 gpurun_out/r03_ref_suite.err
(time python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_default.json')); print(d['ms_per_step'], d['cpu_baseline'].get('kernel_sets_tried'), d['roofline'].get('traffic'), d['roofline'].get('traffic_stale'), d['roofline'].get('frac_whole_factor'))"
python bench.py --gpus 2; echo "rc=$?"
