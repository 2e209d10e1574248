# round 6: one launch per tree level in solve() (solveLevelFused) against the two launches it replaces
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_solve_gpu.py tests/test_solve_perop_gpu.py tests/test_sweep_gpu.py tests/test_solve_wide_gpu.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2; do
python tools/ab_solve.py "-" 2>&1 | grep -v "Warning\|amdgpu.ids"
done
