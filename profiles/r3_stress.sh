cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress.py 70000 400 2>&1 | grep -v amdgpu | tail -6
timeout 900 python tools/stress.py 80000 150 big 2>&1 | grep -v amdgpu | tail -6
