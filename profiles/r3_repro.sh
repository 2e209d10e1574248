cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_factor_gpu.py -x -q -m gpu -k "indefinite" 2>&1 | tail -12 | cut -c1-250
