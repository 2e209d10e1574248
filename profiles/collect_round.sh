# End-of-round evidence, all from the default bench command (run through gpurun from the repo root):
#   the plain bench line, kernel-trace + stats of `python bench.py`, a per-stream timeline, the PMC
#   passes (one counter group per run, kernel-trace only), bench lines of the other workloads.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
python bench.py > gpurun_out/${TAG}_bench_bal871.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o s -- python bench.py --no-cpu-baseline > gpurun_out/${TAG}_stats_bench.json 2> gpurun_out/${TAG}_stats.err
python profiles/roofline_from_rocprof.py gpurun_out/${TAG}_stats gpurun_out/${TAG}_stats_bench.json ${TAG}_bench_bal871 > gpurun_out/${TAG}_roofline.txt 2>&1
cp profiles/${TAG}_bench_bal871_kernel_stats.txt profiles/rocprof_roofline.json gpurun_out/
python profiles/stream_timeline.py gpurun_out/${TAG}_stats > gpurun_out/${TAG}_bench_bal871_timeline.txt 2>&1
bash profiles/collect_pmc.sh
python profiles/make_pmc_traffic.py bal871 gpurun_out/pmc_d/d_results.db gpurun_out/pmc_e/e_results.db 2 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras (profiles/collect_pmc.sh), build ${TAG}" > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/
for d in a b c d e f; do python profiles/summarize_pmc.py gpurun_out/pmc_$d/${d}_results.db; done > gpurun_out/${TAG}_bench_bal871_pmc.txt 2>&1
for w in tridiag grid82 flat50k bal-small; do python bench.py --workload $w --no-extras > gpurun_out/${TAG}_bench_$w.json 2>/dev/null; done
python bench.py > gpurun_out/${TAG}_bench_bal871_final.json 2>/dev/null
