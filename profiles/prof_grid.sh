# kernel stats of the batched C4 workload (64 x GRID 82x82) and of the single matrix
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g64.json
python bench.py --workload grid82 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g1.json
rm -rf gpurun_out/g64_stats gpurun_out/g1_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/g64_stats -o s -- python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/g1_stats -o s -- python bench.py --workload grid82 --no-extras --no-cpu-baseline --no-profile --steps 3 --warmup 1 > /dev/null 2>&1
python profiles/kstats.py gpurun_out/g64_stats/s_results.db 4 > gpurun_out/g64_kstats.txt
python profiles/kstats.py gpurun_out/g1_stats/s_results.db 4 > gpurun_out/g1_kstats.txt
