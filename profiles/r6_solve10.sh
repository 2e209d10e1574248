# round 6: solve() with ten right-hand sides: from which width of a run of columns does the matrix-core sweep pay?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/ab_solve.py --probs=10,21,11,12,32,flat50k,bal1723,bal871 "BSP_SWEEP_MFMA_MIN_WIDTH=0" "BSP_SWEEP_MFMA_MIN_WIDTH=100000" 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/ diff [0-9e.+-]*\/[0-9e.+-]*//g'
