cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --workload grid82 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
rm -rf gpurun_out/pmc_gf
timeout 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum -d gpurun_out/pmc_gf -o f -- $CMD > gpurun_out/pmc_gf.log 2>&1
python profiles/summarize_pmc.py gpurun_out/pmc_gf/f_results.db 2>&1 | grep -E "==|updateTile|trsmPanel|elimGatherTiny|potrfPanel|chainStep" | cut -c1-330
rm -rf gpurun_out/pmc_gf
