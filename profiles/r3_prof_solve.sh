cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/solve_stats
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/solve_stats -o s -- python tools/solve_time.py bal871 > gpurun_out/solve_time.txt 2>&1
tail -3 gpurun_out/solve_time.txt
python profiles/kstats.py gpurun_out/solve_stats/s_results.db 1 2>/dev/null | grep -E "solve|kernel" | head -14
