cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/g8_stats
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/g8_stats -o s -- python bench.py --workload grid82 --batch 8 --no-extras --no-cpu-baseline --no-profile --steps 2 --warmup 1 > gpurun_out/g8.log 2>&1
python profiles/kseq.py gpurun_out/g8_stats/s_results.db elimFactor > gpurun_out/g8_seq.txt 2>&1
rm -rf gpurun_out/g8_stats
tail -2 gpurun_out/g8.log | cut -c1-300
