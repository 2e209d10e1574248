# round 6: narrow root lumps that follow other levels -- the WHOLE lump as the persistent tail (not all but the first block)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_tail_gpu.py -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
python tools/ab_suite.py --reps=15 "--filter=30_GRID|33_GRID|40_MERI|41_MERI|grid82" "BSP_TAIL_WHOLE=0" - 2>&1 | grep -v "Warning\|amdgpu.ids"
done
