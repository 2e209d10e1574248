# A/B of schedule switches on the default bench workload: ms_per_step + in-situ kernel times
# usage (on the GPU box): bash profiles/ab_run.sh "ENV1=a ENV2=b" "ENV1=c" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/ab_$tag.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_$tag.json"))
print("%-48s %.4f ms  %s" % ("$cfg", d["ms_per_step"], {k: v[0] for k, v in d["kernel_ms"].items()}))
PY
done
