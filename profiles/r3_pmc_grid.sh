cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --workload grid82 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
rm -rf gpurun_out/pmc_gf gpurun_out/pmc_ga
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum -d gpurun_out/pmc_gf -o f -- $CMD > gpurun_out/pmc_gf.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_gf/f_results.db 2>&1 | grep -E "==|updateTile|trsmPanel|elimGatherTiny|potrfPanel|chainStep"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_ga -o a -- $CMD > gpurun_out/pmc_ga.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_ga/a_results.db 2>&1 | grep -E "==|updateTile|trsmPanel|elimGatherTiny|potrfPanel|chainStep"
