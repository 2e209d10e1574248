"""randomised parity stress (long sweeps; tests/test_stress_gpu.py runs a seeded slice of the same
cases under pytest).  usage: python tools/stress.py [first_seed] [count] [big|families]
(big: 300-900 parameters, denser: wide lumps, chain steps and lookahead units;
 families: the reference's grid / meridians / flat + Schur generators at random small sizes)"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from stress_cases import run_case, run_family_case

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
big = len(sys.argv) > 3 and sys.argv[3] == "big"
families = len(sys.argv) > 3 and sys.argv[3] == "families"
bad = 0
for seed in range(first, first + count):
    try:
        run_family_case(seed) if families else run_case(seed, big)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED:", repr(e)[:400])
print("stress: %d cases, %d failures" % (count, bad))
sys.exit(1 if bad else 0)
