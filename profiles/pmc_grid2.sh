cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --workload grid82 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
rm -rf gpurun_out/g64_pmc_a gpurun_out/g64_pmc_b gpurun_out/g64_pmc_c
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d gpurun_out/g64_pmc_a -o a -- $CMD > gpurun_out/g64_pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/g64_pmc_b -o b -- $CMD > gpurun_out/g64_pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d gpurun_out/g64_pmc_c -o c -- $CMD > gpurun_out/g64_pmc_c.log 2>&1
python profiles/summarize_pmc.py gpurun_out/g64_pmc_a/a_results.db gpurun_out/g64_pmc_b/b_results.db gpurun_out/g64_pmc_c/c_results.db 2>&1 | grep -E "==|elimGatherTiny|updateTile |trsmPanel " > gpurun_out/g64_pmc2.txt
tail -3 gpurun_out/g64_pmc_a.log
