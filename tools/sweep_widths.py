"""every width of a dense last lump (and of a dense lump followed by a second one): device factor
against numpy (tests/stress_cases.py run_width_case).  usage: python tools/sweep_widths.py [first] [last] [step]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from stress_cases import run_width_case

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bad = 0
for W in range(first, last + 1, step):
    for tail in (0, 70):
        err = run_width_case(W, tail)
        if not err < 1e-12:
            bad += 1
            print("W %d tail %d: rel err %.3e" % (W, tail, err))
print("sweep %d..%d step %d: %d failures" % (first, last, step, bad))
sys.exit(1 if bad else 0)
