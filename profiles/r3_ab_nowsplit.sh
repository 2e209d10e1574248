cd $GRAFT_REPO_ROOT
python -m pytest tests/test_factor_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "lookahead or schedule or c3_bal871_full or c3_bal871_weak or c2_flat_50k_full or wide or many or captured" 2>&1 | tail -4
bash profiles/ab_run.sh "BSP_NOW_SPLIT=0" "BSP_NOW_SPLIT=1" "BSP_NOW_SPLIT=0" "BSP_NOW_SPLIT=1"
