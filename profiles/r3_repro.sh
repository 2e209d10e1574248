cd $GRAFT_REPO_ROOT
python -m pytest tests/test_solve_gpu.py tests/test_solve_perop_gpu.py -x -q -m gpu 2>&1 | tail -5
