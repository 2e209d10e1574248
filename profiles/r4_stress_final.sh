# randomised sweep on the final sources of round 4 (fresh seed ranges), written to gpurun_out/r04_stress_final.txt
{
echo "# tools/stress.py 600000 2500 / 700000 600 families / 800000 400 big / stress_wide_spans.py 900000 150, final sources of round 4"
timeout 700 python tools/stress.py 600000 2500 2>&1 | tail -3
timeout 500 python tools/stress.py 700000 600 families 2>&1 | tail -3
timeout 700 python tools/stress.py 800000 400 big 2>&1 | tail -3
timeout 400 python tools/stress_wide_spans.py 900000 150 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_bench_driver_style.json
} > gpurun_out/r04_stress_final.txt 2>&1
