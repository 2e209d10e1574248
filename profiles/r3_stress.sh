cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 700 python tools/stress.py 100000 1500 2>&1 | grep -v amdgpu | tail -4
timeout 500 python tools/stress.py 110000 300 big 2>&1 | grep -v amdgpu | tail -4
timeout 300 python tools/stress.py 1000 800 families 2>&1 | grep -v amdgpu | tail -4
timeout 300 python tools/stress_wide_spans.py 1000 300 2>&1 | grep -v amdgpu | tail -4
