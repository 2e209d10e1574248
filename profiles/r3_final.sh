cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r03_gpu_tests.txt; cat gpurun_out/r03_gpu_tests.txt
timeout 1500 bash profiles/collect_round.sh r03
python bench.py --suite ref --suite-out gpurun_out/r03_ref_suite.json > gpurun_out/r03_ref_suite.log 2>/dev/null; tail -1 gpurun_out/r03_ref_suite.log
timeout 300 bash profiles/prof_grid.sh
python tools/vendor_compare.py > gpurun_out/r03_vendor_compare.txt 2>&1; cat gpurun_out/r03_vendor_compare.txt
BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/libbaspacho_amd_trace.so timeout 300 python tools/trace_potrf.py > gpurun_out/r03_trace_potrf.txt 2>&1; tail -4 gpurun_out/r03_trace_potrf.txt
rm -rf gpurun_out/*_stats gpurun_out/pmc_? gpurun_out/g64_stats gpurun_out/g1_stats
ls gpurun_out | head -80
