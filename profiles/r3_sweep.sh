cd $GRAFT_REPO_ROOT
timeout 1700 python tools/sweep_widths.py 1 1100 1 2>&1 | grep -v amdgpu.ids | tail -15
