# randomised sweeps on the final sources of round 6 (fresh seed ranges): the narrow-root tails, the folded
# potrf, the level-occupancy rule and the no-tail plan of batches are on by default; the second part forces
# every batch through the two-halves path; the last forces every root lump of >= 2 outer blocks through a tail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "# tools/stress.py 3000000 1500 / 3100000 500 families / 3200000 400 big; BSP_SUB_BATCH_MIN=2: 3300000 600 / 3400000 200 families; BSP_TAIL_MIN_BLOCKS=2 BSP_TAIL_BLOCKS=3: 3500000 300 families / 3600000 300 big"
timeout 700 python tools/stress.py 3000000 1500 2>&1 | tail -3
timeout 600 python tools/stress.py 3100000 500 families 2>&1 | tail -3
timeout 700 python tools/stress.py 3200000 400 big 2>&1 | tail -3
export BSP_SUB_BATCH_MIN=2
timeout 500 python tools/stress.py 3300000 600 2>&1 | tail -3
timeout 400 python tools/stress.py 3400000 200 families 2>&1 | tail -3
unset BSP_SUB_BATCH_MIN
export BSP_TAIL_MIN_BLOCKS=2 BSP_TAIL_BLOCKS=3
timeout 500 python tools/stress.py 3500000 300 families 2>&1 | tail -3
timeout 600 python tools/stress.py 3600000 300 big 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_stress_final.txt
cat gpurun_out/r06_stress_final.txt
