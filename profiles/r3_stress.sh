cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress.py 5000 400 2>&1 | tail -5
timeout 900 python tools/stress.py 7000 120 big 2>&1 | tail -5
