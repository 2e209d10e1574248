#!/bin/bash
# Mutation test of the dense-lump schedule verifier (csrc/hip_plan.cpp, verifyDenseLump): each
# mutation removes one ordering rule of buildDenseLump in a scratch copy of the library; the verifier
# must reject the resulting plan.  Host only.  usage: bash tools/mutate_verifier.sh
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/baspacho_amd/csrc/hip_plan.cpp
cp "$SRC" /tmp/hip_plan.cpp.orig
trap 'cp /tmp/hip_plan.cpp.orig "$SRC"; touch "$SRC"' EXIT
muts=(
  's/          op(kDlWait, 1, optDone\[h\]);/          ;/'
  's/      if (evT >= 0) op(kDlWait, 2, evT);/      ;/'
  's/groupFork \&\& u.dl >= b + group + 3/groupFork \&\& u.dl >= b + group + 2/'
  's/          if (ride \&\& evDuePrev >= 0) op(kDlWait, 0, evDuePrev);/          ;/'
  's/    if (hasT || anyDue) op(kDlWait, 1, evCH);/    ;/'
  's/std::min(i - 2, c)});/std::min(i - 1, c)});/'
)
fail=0
for m in "${muts[@]}"; do
  cp /tmp/hip_plan.cpp.orig "$SRC"
  sed -i "$m" "$SRC"
  if cmp -s "$SRC" /tmp/hip_plan.cpp.orig; then echo "MUTATION DID NOT APPLY: $m"; fail=1; continue; fi
  (cd "$ROOT/baspacho_amd/csrc" && BSP_OUT=/tmp/libmut.so BSP_BUILD_DIR=/tmp/_build_mut bash build.sh > /tmp/mut_build.log 2>&1) || { echo "build failed: $m"; fail=1; continue; }
  out=$(cd "$ROOT" && BSP_BULK_AHEAD=3 BSP_LIB_PATH=/tmp/libmut.so python - <<'PY'
import numpy as np
import baspacho_amd as B
from baspacho_amd import testing as T
caught = 0
for n in (1700, 3000):
    ss = T.columns_to_structure([set(range(i, n)) for i in range(n)])
    sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
    try:
        sol._testVerifyDenseLumps()
    except RuntimeError as e:
        caught += 1
        msg = str(e)
print("caught" if caught else "MISSED", caught, msg[-110:] if caught else "")
PY
)
  echo "$m -> $out"
  case "$out" in caught*) ;; *) fail=1;; esac
done
exit $fail
