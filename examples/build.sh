#!/bin/bash
# Compile the C++ callers against the library's own headers (gfx950; hipcc cross-compiles without a GPU)
set -euo pipefail
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for ex in bal_bench accessor_views; do
  if [ ! -f examples/$ex ] || [ examples/$ex.cpp -nt examples/$ex ] || [ baspacho_amd/libbaspacho_amd.so -nt examples/$ex ] \
     || [ -n "$(find baspacho_amd/csrc -name '*.h' -newer examples/$ex -print -quit)" ]; then
    $HIPCC -O2 -std=c++17 --offload-arch=gfx950 -I. examples/$ex.cpp -Lbaspacho_amd -lbaspacho_amd \
      -Wl,-rpath,'$ORIGIN/../baspacho_amd' -o examples/$ex
  fi
done
echo "built examples/bal_bench examples/accessor_views"
