cd $GRAFT_REPO_ROOT
python -m pytest tests/test_factor_gpu.py tests/test_full_size_gpu.py tests/test_stress_gpu.py tests/test_bal_gpu.py -x -q -m gpu -k "sparse_elim or bal or elim or schedule or c3 or c5 or random_case or packed or overlapped" 2>&1 | tail -4
bash profiles/ab_run.sh "BSP_GATHER_FUSED_LOAD=1" "BSP_GATHER_FUSED_LOAD=2" "BSP_GATHER_FUSED_LOAD=1" "BSP_GATHER_FUSED_LOAD=2"
