# randomised sweeps on the final sources of round 5 (fresh seed ranges); the second half forces the new
# schedule paths: every batch as two concurrent halves (BSP_SUB_BATCH_MIN=2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "# tools/stress.py 1000000 1500 / 1100000 400 families / 1200000 300 big; then the same generators with BSP_SUB_BATCH_MIN=2: 1300000 1200 / 1400000 300 families / 1500000 200 big"
timeout 600 python tools/stress.py 1000000 1500 2>&1 | tail -3
timeout 500 python tools/stress.py 1100000 400 families 2>&1 | tail -3
timeout 600 python tools/stress.py 1200000 300 big 2>&1 | tail -3
export BSP_SUB_BATCH_MIN=2
timeout 600 python tools/stress.py 1300000 1200 2>&1 | tail -3
timeout 500 python tools/stress.py 1400000 300 families 2>&1 | tail -3
timeout 600 python tools/stress.py 1500000 200 big 2>&1 | tail -3
} > gpurun_out/r05_stress.txt 2>&1
cat gpurun_out/r05_stress.txt
