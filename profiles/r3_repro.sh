cd $GRAFT_REPO_ROOT
python -m pytest tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -3
