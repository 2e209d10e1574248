#!/bin/bash
# Build the MI355X shared library in-tree (gfx950 only).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${BSP_OUT:-../libbaspacho_amd.so}
BUILD=${BSP_BUILD_DIR:-../_build}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics ${BSP_KTRACE:+-DBSP_KTRACE=1} ${BSP_EXTRA_DEFS:-}"
SRCS="bsp_utils.cpp sparse_structure.cpp min_degree.cpp computation_model.cpp elimination_tree.cpp skeleton.cpp solver.cpp hip_plan.cpp c_api.cpp"
mkdir -p $BUILD
OBJS=""
pids=()
for f in $SRCS; do
  o=$BUILD/${f%.cpp}.o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -name '*.h' -newer "$o" -print -quit)" ] || [ ../../include/baspacho_amd.h -nt "$o" ]; then
    $HIPCC $FLAGS -x hip -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for f in hip_backend.hip bal_pipeline.hip; do
  o=$BUILD/${f%.hip}.o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -name '*.h' -newer "$o" -print -quit)" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC -shared -fPIC --offload-arch=gfx950 $OBJS -o $OUT
echo "built $OUT"
