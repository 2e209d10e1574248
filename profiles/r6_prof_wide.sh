# backward elimination pass with the right-hand sides across the lanes: time + counters of one solve-10 run
# bash profiles/r6_prof_wide.sh [NRHS]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-10}
OUT=gpurun_out/r06_wide_nrhs$N.txt
rm -rf gpurun_out/wide_stats
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/wide_stats -o s -- python tools/solve_time.py bal871 --nrhs $N > gpurun_out/wide_time.txt 2>&1
grep nRHS gpurun_out/wide_time.txt > $OUT
python profiles/kstats.py gpurun_out/wide_stats/s_results.db 14 40 2>/dev/null | grep -E "solve|kernel" >> $OUT
for c in "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum"; do
  rm -rf gpurun_out/wide_pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/wide_pmc -o p -- python tools/solve_time.py bal871 --nrhs $N > /dev/null 2>&1
  python profiles/summarize_pmc.py gpurun_out/wide_pmc/p_results.db | grep -E "ElimLumps|ElimGather|ElimDiag|RowsToWide" >> $OUT
done
rm -rf gpurun_out/wide_stats gpurun_out/wide_pmc
cat $OUT
