cd $GRAFT_REPO_ROOT
python -m pytest tests/test_factor_gpu.py -x -q -m gpu -k "chain_window" 2>&1 | tail -4
BSP_CHAIN_WINDOW=1 python -m pytest tests/test_factor_gpu.py tests/test_full_size_gpu.py tests/test_stress_gpu.py tests/test_solve_gpu.py -x -q -m gpu 2>&1 | tail -4
bash profiles/ab_run.sh "BSP_CHAIN_WINDOW=0" "BSP_CHAIN_WINDOW=1" "BSP_CHAIN_WINDOW=0" "BSP_CHAIN_WINDOW=1"
