cd $GRAFT_REPO_ROOT
timeout 1200 python tools/stress_wide_spans.py 0 150 2>&1 | grep -v amdgpu | tail -8
