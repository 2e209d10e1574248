# round-6 evidence on the final sources (run through gpurun from the repo root): bash profiles/r6_final.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_gpu_tests.txt; cat gpurun_out/r06_gpu_tests.txt
timeout 1800 bash profiles/collect_round.sh r06
python profiles/kseq.py gpurun_out/r06_stats/s_results.db elimFactor > gpurun_out/r06_launch_sequence_bal871.txt 2>&1
timeout 900 bash profiles/r5_c4_profile.sh r06
timeout 600 bash profiles/r5_prof_solve.sh r06 pmc
timeout 300 bash profiles/r5_seq.sh r06_grid82 "--workload grid82"; cp gpurun_out/r06_grid82_seq.txt gpurun_out/r06_launch_sequence_grid82.txt
timeout 300 bash profiles/r5_seq.sh r06_grid8 "--workload grid82 --batch 8"; cp gpurun_out/r06_grid8_seq.txt gpurun_out/r06_launch_sequence_grid_batch8.txt
python bench.py --workload grid82 --batch 8 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_bench_grid82_batch8.json
python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_bench_grid82_batch64.json
BSP_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r06_bench_two_ranks_one_gpu_control_flow.json
timeout 1500 python bench.py --suite ref --suite-out gpurun_out/r06_ref_suite.json > gpurun_out/r06_ref_suite.log 2>/dev/null; tail -1 gpurun_out/r06_ref_suite.log | cut -c1-300
rm -rf gpurun_out/*_stats gpurun_out/pmc_?
ls gpurun_out | grep r06
