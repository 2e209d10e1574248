cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress.py 90000 400 2>&1 | grep -v amdgpu | tail -6
timeout 900 python tools/stress.py 95000 120 big 2>&1 | grep -v amdgpu | tail -6
