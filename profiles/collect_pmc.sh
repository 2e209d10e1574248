# PMC passes of bench.py (one counter group per run, kernel-trace only), written under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_a -o a -- $CMD > gpurun_out/pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_b -o b -- $CMD > gpurun_out/pmc_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum -d gpurun_out/pmc_c -o c -- $CMD > gpurun_out/pmc_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_d -o d -- $CMD > gpurun_out/pmc_d.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_e -o e -- $CMD > gpurun_out/pmc_e.log 2>&1
# round 3: texture-addresser / L1 / L2-to-fabric view of the scattered kernels (sparse-elimination update)
# (every pass under `timeout`: a pass with TCC_EA0_RDREQ_DRAM* / TCC_BUSY_avr aborted inside rocprofv3 and
#  then hung for 15 minutes -- those counters are not collected)
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum -d gpurun_out/pmc_f -o f -- $CMD > gpurun_out/pmc_f.log 2>&1
ls gpurun_out/pmc_*/ 
# round 4: MFMA instruction counts per kernel (the dense phase's matrix-pipe work against the algorithmic count)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA -d gpurun_out/pmc_g -o g -- $CMD > gpurun_out/pmc_g.log 2>&1
