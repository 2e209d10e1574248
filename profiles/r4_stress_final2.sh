# randomised sweep on the FINAL sources of round 4 (after the staged-cap / plan-default changes; fresh seeds)
{
echo "# tools/stress.py 1000000 2000 / 1100000 500 families / 1200000 300 big / stress_wide_spans.py 1300000 120"
timeout 600 python tools/stress.py 1000000 2000 2>&1 | tail -2
timeout 400 python tools/stress.py 1100000 500 families 2>&1 | tail -2
timeout 500 python tools/stress.py 1200000 300 big 2>&1 | tail -2
timeout 300 python tools/stress_wide_spans.py 1300000 120 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_stress_final2.txt
