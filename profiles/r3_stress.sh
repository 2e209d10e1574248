cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress.py 50000 600 2>&1 | grep -v amdgpu | tail -6
timeout 900 python tools/stress.py 60000 200 big 2>&1 | grep -v amdgpu | tail -6
