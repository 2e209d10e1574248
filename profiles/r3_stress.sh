cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress.py 9000 500 2>&1 | tail -3
timeout 900 python tools/stress.py 12000 150 big 2>&1 | tail -3
