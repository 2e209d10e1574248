cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "offsets" 2>&1 | tail -4
