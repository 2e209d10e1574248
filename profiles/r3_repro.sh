cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r03_gpu_tests.txt; cat gpurun_out/r03_gpu_tests.txt
