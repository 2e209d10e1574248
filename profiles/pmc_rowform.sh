cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d gpurun_out/pr_a -o a -- $CMD > gpurun_out/pr_a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pr_d -o d -- $CMD > gpurun_out/pr_d.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/pr_b -o b -- $CMD > gpurun_out/pr_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d gpurun_out/pr_c -o c -- $CMD > gpurun_out/pr_c.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pr_a/a_results.db gpurun_out/pr_d/d_results.db gpurun_out/pr_b/b_results.db gpurun_out/pr_c/c_results.db | grep -E "==|elimRow"
