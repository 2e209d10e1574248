cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
rm -rf gpurun_out/pmc_f
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum -d gpurun_out/pmc_f -o f -- $CMD > gpurun_out/pmc_f.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_f/f_results.db 2>&1 | grep -E "==|elimGather|elimFactor|updateTileBulk|chainStep"
rm -rf gpurun_out/pmc_h
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_h -o h -- $CMD > gpurun_out/pmc_h.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_h/h_results.db 2>&1 | grep -E "==|elimGather"
