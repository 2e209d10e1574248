# quick check of a build: GRID benches (residual probe = parity smoke) + the factor tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r5try}
for cfg in "g1:" "g8:--batch 8" "g64:--batch 64"; do
  w=${cfg%%:*}; a=${cfg#*:}
  timeout 300 python bench.py --workload grid82 $a --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_$w.err | tail -1 > gpurun_out/${TAG}_$w.json
done
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/${TAG}_bal871.json
python - <<PY
import json
for w in ["bal871", "g1", "g8", "g64"]:
    try:
        d = json.loads(open("gpurun_out/${TAG}_%s.json" % w).read())
        print(w, d["ms_per_step"], d.get("residual_probe"), {k: v for k, v in d.get("kernel_ms", {}).items()})
    except Exception as e:
        print(w, "FAILED", e)
PY
if [ -n "$2" ]; then timeout 1200 python -m pytest $2 -x -q -m gpu 2>&1 | tail -15; fi
