# kernel stats + PMC of the batched metric config C4 (64 x GRID 82x82): bash profiles/r5_c4_profile.sh TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
CMD="python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline --no-profile"
python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_g64.json
rm -rf gpurun_out/c4_stats
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/c4_stats -o s -- $CMD --steps 3 --warmup 1 > /dev/null 2>&1
python profiles/kstats.py gpurun_out/c4_stats/s_results.db 4 24 > gpurun_out/${TAG}_batched_grid82_kernel_stats.txt
python profiles/kseq.py gpurun_out/c4_stats/s_results.db elimFactor > gpurun_out/${TAG}_launch_sequence_grid_batch64.txt 2>&1
rm -rf gpurun_out/c4_stats
P="--steps 1 --warmup 1"
rm -rf gpurun_out/c4_pmc_*
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/c4_pmc_a -o a -- $CMD $P > gpurun_out/c4_pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d gpurun_out/c4_pmc_b -o b -- $CMD $P > gpurun_out/c4_pmc_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/c4_pmc_d -o d -- $CMD $P > gpurun_out/c4_pmc_d.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/c4_pmc_e -o e -- $CMD $P > gpurun_out/c4_pmc_e.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d gpurun_out/c4_pmc_f -o f -- $CMD $P > gpurun_out/c4_pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/c4_pmc_g -o g -- $CMD $P > gpurun_out/c4_pmc_g.log 2>&1
for d in a b d e f g; do python profiles/summarize_pmc.py gpurun_out/c4_pmc_$d/${d}_results.db; done > gpurun_out/${TAG}_batched_grid82_pmc.txt 2>&1
rm -rf gpurun_out/c4_pmc_*/
cat gpurun_out/${TAG}_g64.json | cut -c1-200
