# A/B of schedule switches on the two extra workloads of the bench line (C4 batched, C5 fp32)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg python - <<PY
import sys, json, time, torch
sys.path.insert(0, ".")
import bench
ctx = {"rank": 0, "world": 1, "device": torch.device("cuda", 0), "dist": None}
r = bench.Runner(ctx, "grid82", 64, True)
e = r.run(3, 1)
c5 = bench.c5_block(ctx["device"])
print("%-44s C4 %.3f ms   C5 factor_f32 %.3f ms refine %.3f ms" % ("$cfg", e["ms_per_step"], c5["factor_f32_ms"], c5["refine_ms"]))
PY
done
