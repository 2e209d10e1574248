# a second, longer randomised sweep on the final sources (fresh seed ranges)
cd $GRAFT_REPO_ROOT
(
timeout 900 python tools/stress.py 200000 4000 2>&1 | tail -3
timeout 900 python tools/stress.py 300000 1200 big 2>&1 | tail -3
timeout 600 python tools/stress.py 400000 800 families 2>&1 | tail -3
timeout 600 python tools/stress_wide_spans.py 500000 150 2>&1 | tail -3
) > gpurun_out/r04_stress_more.txt 2>&1
cat gpurun_out/r04_stress_more.txt
