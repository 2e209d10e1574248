# kernel trace of the headline workload only (no extras), for per-launch inspection
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/bal_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/bal_stats -o s -- python bench.py --no-extras --no-cpu-baseline --no-profile --steps 3 --warmup 1 > /dev/null 2>&1
python profiles/kstats.py gpurun_out/bal_stats/s_results.db 4 > gpurun_out/bal_kstats.txt
python profiles/stream_timeline.py gpurun_out/bal_stats 2 > gpurun_out/bal_timeline.txt 2>&1
