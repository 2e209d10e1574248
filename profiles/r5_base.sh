# round 5 baseline on one box: headline + the latency-bound / batched GRID workloads, kernel stats of C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r5base}
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/${TAG}_bal871.json
python bench.py --workload grid82 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_g1.json
python bench.py --workload grid82 --batch 8 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_g8.json
python bench.py --workload grid82 --batch 64 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_g64.json
python - <<PY
import json
for w in ["bal871", "g1", "g8", "g64"]:
    try:
        d = json.loads(open("gpurun_out/${TAG}_%s.json" % w).read())
        print(w, d["ms_per_step"], d.get("residual_probe"), {k: v for k, v in d.get("kernel_ms", {}).items()})
    except Exception as e:
        print(w, "FAILED", e)
PY
