cd $GRAFT_REPO_ROOT
python -m pytest tests/test_factor_gpu.py -x -q -m gpu -k "chain_window" 2>&1 | tail -3
bash profiles/ab_run.sh "BSP_CHAIN_WINDOW=0" "BSP_CHAIN_WINDOW=1" "BSP_CHAIN_WINDOW=1 BSP_MERGED_CHAIN=0" "BSP_CHAIN_WINDOW=0 BSP_MERGED_CHAIN=0" "BSP_CHAIN_WINDOW=1 BSP_MERGED_CHAIN=0 BSP_BULK_AHEAD=0.4"
