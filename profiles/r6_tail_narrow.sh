# round 6, item 1b: persistent tail on NARROW root lumps (2-5 outer blocks), one matrix, reference families
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python tools/ab_suite.py --reps=15 "--filter=30_GRID|33_GRID|40_MERI|41_MERI|grid82" "BSP_TAIL_NARROW_MIN=0" - 2>&1 | grep -v "Warning\|amdgpu.ids"
done
python tools/ab_suite.py --reps=9 "--filter=^1|^2|31_|32_" "BSP_TAIL_NARROW_MIN=0" - 2>&1 | grep -v "Warning\|amdgpu.ids"
