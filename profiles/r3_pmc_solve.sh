cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
rm -rf gpurun_out/pmc_s$v
BSP_SOLVE_ELIM_STAGED=$v timeout 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum -d gpurun_out/pmc_s$v -o f -- python tools/solve_time.py bal871 > gpurun_out/pmc_s$v.log 2>&1
echo "STAGED=$v"; python profiles/summarize_pmc.py gpurun_out/pmc_s$v/f_results.db 2>&1 | grep -E "solveElim" | cut -c1-330
rm -rf gpurun_out/pmc_t$v
BSP_SOLVE_ELIM_STAGED=$v timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD -d gpurun_out/pmc_t$v -o f -- python tools/solve_time.py bal871 > gpurun_out/pmc_t$v.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_t$v/f_results.db 2>&1 | grep -E "solveElimLumps" | cut -c1-330
done
