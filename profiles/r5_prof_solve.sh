# solve(): kernel stats for nRHS = 1 and 10 in separate runs: bash profiles/r5_prof_solve.sh TAG [pmc]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
for n in 1 10; do
  rm -rf gpurun_out/solve_stats
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/solve_stats -o s -- python tools/solve_time.py bal871 --nrhs $n > gpurun_out/${TAG}_solve_time_nrhs$n.txt 2>&1
  grep nRHS gpurun_out/${TAG}_solve_time_nrhs$n.txt
  # 21 solve calls per run (7 each of solve / solveL / solveLt = 14 forward + 14 backward passes)
  python profiles/kstats.py gpurun_out/solve_stats/s_results.db 14 40 2>/dev/null | grep -E "solve|kernel" > gpurun_out/${TAG}_solve_kernel_stats_nrhs$n.txt
  cat gpurun_out/${TAG}_solve_kernel_stats_nrhs$n.txt
  if [ -n "$2" ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf gpurun_out/solve_pmc
      timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/solve_pmc -o p -- python tools/solve_time.py bal871 --nrhs $n > /dev/null 2>&1
      python profiles/summarize_pmc.py gpurun_out/solve_pmc/p_results.db | grep -E "==|solve" | sed "s/^/nrhs$n /"
    done > gpurun_out/${TAG}_solve_pmc_nrhs$n.txt
    cat gpurun_out/${TAG}_solve_pmc_nrhs$n.txt
  fi
done
rm -rf gpurun_out/solve_stats gpurun_out/solve_pmc
