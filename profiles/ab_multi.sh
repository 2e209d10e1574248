# A/B of library variants over several workloads, alternating runs:
#   bash profiles/ab_multi.sh [reps] lib1 [lib2 ...]     ("" = the product library is always included)
# prints one line per run: workload, library, ms_per_step, residual probe
reps=${1:-3}; shift
for wl in "bal871:--steps 10" "grid82:--workload grid82 --steps 50" "grid82x8:--workload grid82 --batch 8 --steps 30" "grid82x64:--workload grid82 --batch 64 --steps 10" "flat50k:--workload flat50k --steps 5"; do
  name=${wl%%:*}; args=${wl#*:}
  for rep in $(seq $reps); do
    for lib in "" $*; do
      if [ -n "$lib" ]; then export BSP_LIB_PATH=$GRAFT_REPO_ROOT/baspacho_amd/lib$lib.so; else unset BSP_LIB_PATH; fi
      python bench.py $args --no-extras --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', '${lib:-new}', d['ms_per_step'], d['residual_probe'])"
    done
  done
done
