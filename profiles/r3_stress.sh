cd $GRAFT_REPO_ROOT
timeout 420 python tools/stress.py 200000 1500 2>&1 | grep -v amdgpu | tail -3
timeout 200 python tools/stress.py 5000 1000 families 2>&1 | grep -v amdgpu | tail -3
timeout 300 python tools/sweep_widths.py 3301 5200 97 2>&1 | grep -v amdgpu | tail -3
