cd $GRAFT_REPO_ROOT
timeout 1500 python tools/stress.py 0 400 families 2>&1 | grep -v amdgpu | tail -8
