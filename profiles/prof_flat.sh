# kernel stats of FLAT-50k and the C5 factor (BAL-1723, fp32)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/flat_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/flat_stats -o s -- python bench.py --workload flat50k --no-extras --no-cpu-baseline --no-profile --steps 3 --warmup 1 > /dev/null 2>&1
python profiles/kstats.py gpurun_out/flat_stats/s_results.db 4 > gpurun_out/flat_kstats.txt
