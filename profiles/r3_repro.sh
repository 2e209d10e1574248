cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_concurrency_gpu.py -x -q -m gpu 2>&1 | tail -4; done
