cd $GRAFT_REPO_ROOT
timeout 1700 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from stress_cases import run_width_case
bad = 0; n = 0
for k in range(4, 15):
    for r in (0, 1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255):
        for tail in (0, 70):
            W = 256 * k + r
            err = run_width_case(W, tail)
            n += 1
            if not err < 1e-12:
                bad += 1
                print("W", W, "tail", tail, "err", err)
print("residue sweep: %d cases, %d failures" % (n, bad))
PY
