# randomised sweeps on the final sources of round 6 (fresh seed ranges): the narrow-root tails, the folded
# potrf and the level-occupancy rule are on by default; the last block forces every root lump of >= 2
# outer blocks through the tail (BSP_TAIL_NARROW_MIN=2 is the default; BSP_TAIL_MIN_BLOCKS=2 makes the
# wide rule take 2-5 block lumps for batches too)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "# tools/stress.py 2000000 1500 / 2100000 500 families / 2200000 400 big; then BSP_TAIL_MIN_BLOCKS=2 BSP_TAIL_BLOCKS=3: 2300000 300 families / 2400000 300 big"
timeout 700 python tools/stress.py 2000000 1500 2>&1 | tail -3
timeout 600 python tools/stress.py 2100000 500 families 2>&1 | tail -3
timeout 700 python tools/stress.py 2200000 400 big 2>&1 | tail -3
export BSP_TAIL_MIN_BLOCKS=2 BSP_TAIL_BLOCKS=3
timeout 500 python tools/stress.py 2300000 300 families 2>&1 | tail -3
timeout 600 python tools/stress.py 2400000 300 big 2>&1 | tail -3
} > gpurun_out/r06_stress.txt 2>&1
cat gpurun_out/r06_stress.txt
