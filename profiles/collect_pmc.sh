# PMC passes of bench.py (one counter group per run, kernel-trace only), written under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_a -o a -- $CMD > gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_b -o b -- $CMD > gpurun_out/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum -d gpurun_out/pmc_c -o c -- $CMD > gpurun_out/pmc_c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_d -o d -- $CMD > gpurun_out/pmc_d.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_e -o e -- $CMD > gpurun_out/pmc_e.log 2>&1
ls gpurun_out/pmc_*/ 
