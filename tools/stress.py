"""randomised parity stress: random structures / parameter sizes / elimination sets / dtypes / batch
sizes, factor + solve + addMvFrom on the device against dense numpy.
usage: python tools/stress.py [first_seed] [count] [big]   (big: 300-900 parameters, denser: wide lumps,
chain steps and lookahead units)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import dense_lower_chol, lower_of, spd_data, to_dev

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
big = len(sys.argv) > 3 and sys.argv[3] == "big"
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    size = int(rng.integers(300, 900)) if big else int(rng.integers(20, 260))
    fill = float(rng.choice([0.05, 0.15, 0.5])) if big else float(rng.choice([0.01, 0.03, 0.08, 0.3]))
    pmax = int(rng.choice([3, 6, 9])) if big else int(rng.choice([1, 3, 5, 9, 23]))
    sizes = rng.integers(1, pmax + 1, size=size).astype(np.int64)
    cols = T.random_cols(size, fill, 100 + seed)
    ranges = []
    if rng.random() < 0.5:
        k = int(rng.integers(5, max(6, size // 2)))
        cols = T.make_independent_elim_set(cols, 0, k)
        if rng.random() < 0.5:
            ranges = [0, k]
    ss = T.columns_to_structure(cols)
    st = B.Settings(findSparseEliminationRanges=bool(rng.random() < 0.7))
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    tol = 1e-8 if dtype == np.float64 else 3e-4
    try:
        sol = B.create_solver(st, sizes, ss, ranges)
        n = sol.order()
        nb = int(rng.choice([1, 1, 2, 5]))
        datas = [spd_data(sol, 7 * seed + q, dtype=dtype) for q in range(nb)]
        devs = [to_dev(d) for d in datas]
        sol.factor(devs if nb > 1 else devs[0])
        for q in range(nb):
            L, A = dense_lower_chol(sol, datas[q])
            got = lower_of(sol, devs[q].cpu().numpy())
            err = np.linalg.norm(got - L) / np.linalg.norm(L)
            assert err < tol, ("factor", q, err)
        nrhs = int(rng.choice([1, 3]))
        rhs = rng.standard_normal(n * nrhs).astype(dtype)
        v = to_dev(rhs)
        sol.solve(devs[0], v, n, nrhs)
        L, A = dense_lower_chol(sol, datas[0])
        X = np.linalg.solve(A, rhs.astype(np.float64).reshape(nrhs, n).T)
        got = v.cpu().numpy().astype(np.float64).reshape(nrhs, n).T
        err = np.linalg.norm(got - X) / np.linalg.norm(X)
        assert err < tol * 50, ("solve", err)
        # addMvFrom on the un-factored matrix
        a_dev = to_dev(datas[0])
        xin = rng.standard_normal(n).astype(dtype)
        yout = to_dev(np.zeros(n, dtype=dtype))
        sol.addMvFrom(a_dev, 0, to_dev(xin), n, yout, n, 1, 1.0)
        ref = A @ xin.astype(np.float64)
        err = np.linalg.norm(yout.cpu().numpy().astype(np.float64) - ref) / np.linalg.norm(ref)
        assert err < tol * 10, ("addMv", err)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("seed", seed, "size", size, "fill", fill, "pmax", pmax, "ranges", ranges, dtype.__name__, "FAILED:", repr(e)[:300])
print("stress: %d cases, %d failures" % (count, bad))
sys.exit(1 if bad else 0)
