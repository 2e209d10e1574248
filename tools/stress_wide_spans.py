"""random structures whose parameter blocks are wider than a panel (64) or an outer block (256):
python tools/stress_wide_spans.py [first_seed] [count]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import dense_lower_chol, lower_of, spd_data, to_dev

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    size = int(rng.integers(3, 40))
    sizes = np.where(rng.random(size) < 0.3, rng.integers(65, 400, size=size), rng.integers(1, 70, size=size)).astype(np.int64)
    cols = T.random_cols(size, float(rng.choice([0.1, 0.3, 0.7])), 100 + seed)
    ss = T.columns_to_structure(cols)
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    tol = 1e-9 if dtype == np.float64 else 3e-4
    sol = B.create_solver(B.Settings(findSparseEliminationRanges=bool(rng.random() < 0.5)), sizes, ss, [])
    n = sol.order()
    try:
        data = spd_data(sol, seed, dtype=dtype)
        dev = to_dev(data)
        sol.factor(dev)
        L, A = dense_lower_chol(sol, data)
        got = lower_of(sol, dev.cpu().numpy())
        err = np.linalg.norm(got - L) / np.linalg.norm(L)
        assert err < tol, ("factor", err)
        rhs = rng.standard_normal(n * 2).astype(dtype)
        v = to_dev(rhs)
        sol.solve(dev, v, n, 2)
        X = np.linalg.solve(A, rhs.astype(np.float64).reshape(2, n).T)
        serr = np.linalg.norm(v.cpu().numpy().astype(np.float64).reshape(2, n).T - X) / np.linalg.norm(X)
        assert serr < tol * 50, ("solve", serr)
        # pseudoFactorFrom on wide spans (PartialFactorSolveTest.cpp:296-395) against the per-span dense form
        if sol.numLumps() >= 2:
            lump = int(rng.integers(1, sol.numLumps()))
            span = int(sol.skel()["lumpToSpan"][lump])
            d2 = to_dev(data)
            sol.factorUpTo(d2, span)
            sol.factorFrom(d2, span)
            g2 = lower_of(sol, d2.cpu().numpy())
            perr = np.linalg.norm(g2 - L) / np.linalg.norm(L)
            assert perr < tol, ("partial", span, perr)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("seed", seed, "order", n, "sizes", list(sizes[:12]), "FAILED:", repr(e)[:300])
print("wide spans: %d cases, %d failures" % (count, bad))
sys.exit(1 if bad else 0)
