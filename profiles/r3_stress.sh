cd $GRAFT_REPO_ROOT
timeout 1500 python tools/stress.py 20000 3000 2>&1 | tail -6
timeout 1500 python tools/stress.py 40000 700 big 2>&1 | tail -6
