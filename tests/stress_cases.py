"""Randomised parity cases shared by tests/test_stress_gpu.py (a seeded slice, run by the driver)
and tools/stress.py (long sweeps): random structures / parameter sizes / elimination sets / dtypes /
batch sizes, factor + solve + addMvFrom on the device against dense numpy, the protocol of the
reference's random families (tests/FactorTest.cpp:75-107,185-219; tests/SolveTest.cpp)."""
import numpy as np

import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import dense_lower_chol, lower_of, spd_data, to_dev


def run_case(seed, big=False):
    """raises AssertionError (with the failing stage and error) when the device result is off"""
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
    desc = dict(seed=seed, size=size, fill=fill, pmax=pmax, ranges=ranges, dtype=dtype.__name__)
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
        assert err < tol, ("factor", q, err, desc)
    nrhs = int(rng.choice([1, 3, 2, 7]))  # (round 5: 2 and 7 take the other widths of the multi-RHS elimination solves)
    rhs = rng.standard_normal(n * nrhs).astype(dtype)
    v = to_dev(rhs)
    sol.solve(devs[0], v, n, nrhs)
    L, A = dense_lower_chol(sol, datas[0])
    X = np.linalg.solve(A, rhs.astype(np.float64).reshape(nrhs, n).T)
    got = v.cpu().numpy().astype(np.float64).reshape(nrhs, n).T
    err = np.linalg.norm(got - X) / np.linalg.norm(X)
    assert err < tol * 50, ("solve", err, desc)
    # addMvFrom on the un-factored matrix
    a_dev = to_dev(datas[0])
    xin = rng.standard_normal(n).astype(dtype)
    yout = to_dev(np.zeros(n, dtype=dtype))
    sol.addMvFrom(a_dev, 0, to_dev(xin), n, yout, n, 1, 1.0)
    ref = A @ xin.astype(np.float64)
    err = np.linalg.norm(yout.cpu().numpy().astype(np.float64) - ref) / np.linalg.norm(ref)
    assert err < tol * 10, ("addMv", err, desc)
    # partial solves at a random lump boundary (PartialFactorSolveTest.cpp:157-262): forward
    # substitution over the leading then the trailing columns is solveL, the backward one in the
    # opposite order solveLt, and solveL then solveLt is solve (checked against numpy above)
    er0 = sol.sparseEliminationRanges()
    first_dense = int(er0[-1]) if len(er0) else 0
    if sol.numLumps() - first_dense >= 2:
        # (a boundary inside an elimination range is a precondition failure, Solver.cpp:281-290)
        span = int(sol.skel()["lumpToSpan"][int(rng.integers(first_dense + 1, sol.numLumps()))])
        ref_v = v.cpu().numpy().astype(np.float64)
        w1 = to_dev(rhs)
        sol.solveLUpTo(devs[0], span, w1, n, nrhs)
        sol.solveLFrom(devs[0], span, w1, n, nrhs)
        w2 = to_dev(rhs)
        sol.solveL(devs[0], w2, n, nrhs)
        a1, a2 = w1.cpu().numpy().astype(np.float64), w2.cpu().numpy().astype(np.float64)
        err = np.linalg.norm(a1 - a2) / np.linalg.norm(a2)
        assert err < tol * 10, ("solveL split", span, err, desc)
        sol.solveLtFrom(devs[0], span, w1, n, nrhs)
        sol.solveLtUpTo(devs[0], span, w1, n, nrhs)
        sol.solveLt(devs[0], w2, n, nrhs)
        a1, a2 = w1.cpu().numpy().astype(np.float64), w2.cpu().numpy().astype(np.float64)
        err = np.linalg.norm(a1 - ref_v) / np.linalg.norm(ref_v)
        assert err < tol * 50, ("solveLt split", span, err, desc)
        err = np.linalg.norm(a2 - ref_v) / np.linalg.norm(ref_v)
        assert err < tol * 50, ("solveL + solveLt", err, desc)
    # one case in five: the same factor and solve through the reference's op-by-op loops over the
    # per-op NumericCtx / SolveCtx virtuals (MatOps.h:113-184; bsp_force_per_op)
    if rng.random() < 0.2:
        d3 = to_dev(datas[0])
        w3 = to_dev(rhs)
        with sol.forcePerOp():
            sol.factor(d3)
            sol.solve(d3, w3, n, nrhs)
        got = lower_of(sol, d3.cpu().numpy())
        err = np.linalg.norm(got - L) / np.linalg.norm(L)
        assert err < tol, ("per-op factor", err, desc)
        a3 = w3.cpu().numpy().astype(np.float64).reshape(nrhs, n).T
        err = np.linalg.norm(a3 - X) / np.linalg.norm(X)
        assert err < tol * 50, ("per-op solve", err, desc)
    # factorUpTo(k) then factorFrom(k) == factor, for a random lump boundary beyond the elimination
    # ranges (PartialFactorSolveTest.cpp:37-155)
    er = sol.sparseEliminationRanges()
    dense_from = int(er[-1]) if len(er) else 0
    if sol.numLumps() - dense_from >= 2:
        lump = int(rng.integers(dense_from + 1, sol.numLumps()))
        span = int(sol.skel()["lumpToSpan"][lump])
        d2 = to_dev(datas[0])
        sol.factorUpTo(d2, span)
        sol.factorFrom(d2, span)
        got = lower_of(sol, d2.cpu().numpy())
        err = np.linalg.norm(got - L) / np.linalg.norm(L)
        assert err < tol, ("partial", span, err, desc)
    return desc


def run_width_case(W, tail, dtype=np.float64):
    """a dense lump of width W (spans of 8, a ragged last one), optionally followed by a second lump of
    `tail` columns that most of its spans see; returns the relative error of the device factor
    against numpy.  tools/sweep_widths.py walks every W; tests/test_stress_gpu.py a slice around the
    panel (64) and outer-block (256) boundaries."""
    sizes = [8] * (W // 8) + ([W % 8] if W % 8 else [])
    n0 = len(sizes)
    sizes = sizes + [7] * (tail // 7)
    nparam = len(sizes)
    cols = [list(range(c, n0)) + [q for q in range(n0, nparam) if (q + c) % 4 != 0] for c in range(n0)]
    cols += [list(range(c, nparam)) for c in range(n0, nparam)]
    ss = T.columns_to_structure(cols)
    st = B.Settings(findSparseEliminationRanges=False)
    sol = B.create_solver(st, np.asarray(sizes, dtype=np.int64), ss, [])
    d = spd_data(sol, 5 + W, dtype=dtype)
    dev = to_dev(d)
    sol.factor(dev)
    L, A = dense_lower_chol(sol, d)
    got = lower_of(sol, dev.cpu().numpy())
    err = np.linalg.norm(got - L) / np.linalg.norm(L)
    # the device solve on the same factor (block triangles through inverted diagonal blocks, wide lumps)
    n = sol.order()
    rhs = np.random.default_rng(W).standard_normal(n).astype(dtype)
    v = to_dev(rhs)
    sol.solve(dev, v, n, 1)
    X = np.linalg.solve(A, rhs.astype(np.float64))
    serr = np.linalg.norm(v.cpu().numpy().astype(np.float64) - X) / np.linalg.norm(X)
    return max(err, serr * (1e-2 if dtype == np.float64 else 1e-1))


def run_family_case(seed):
    """the reference's structured families at random small sizes (TestingMatGen.cpp: grid, meridians,
    flat + Schur set; Bench.cpp:290-367 uses them at benchmark size): deep elimination trees, which
    the random column structures of run_case do not produce.  Device factor + solve against numpy."""
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        w, h = int(rng.integers(4, 40)), int(rng.integers(4, 40))
        ss = T.gen_grid(w, h, float(rng.choice([1.0, 0.6, 0.25])), int(rng.integers(1, 4)), seed)
        ranges = []
    elif kind == 1:
        band = int(rng.integers(3, 25))
        ss = T.gen_meridians(int(rng.integers(2, 7)), band + int(rng.integers(0, 130)), 0.5, band,
                             band + int(rng.integers(0, 40)), int(rng.integers(0, 3)), int(rng.integers(0, 3)), seed)
        ranges = []
    else:
        base = T.gen_flat(int(rng.integers(20, 200)), float(rng.choice([0.02, 0.1, 0.3])), seed)
        k = int(rng.integers(10, 400))
        ss = T.add_schur_set(base, k, float(rng.choice([0.02, 0.1])), seed + 1000)
        ranges = [0, k] if rng.random() < 0.6 else []
    n_par = ss.order()
    pmax = int(rng.choice([1, 2, 3, 6]))
    sizes = rng.integers(1, pmax + 1, size=n_par).astype(np.int64) if rng.random() < 0.5 else np.full(n_par, pmax, dtype=np.int64)
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    tol = 1e-9 if dtype == np.float64 else 3e-4
    desc = dict(seed=seed, kind=["grid", "meridians", "flat+schur"][kind], params=n_par, pmax=pmax, dtype=dtype.__name__)
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    n = sol.order()
    if n > 6000:
        return desc  # (dense check too slow)
    data = spd_data(sol, 3 + seed, dtype=dtype)
    dev = to_dev(data)
    sol.factor(dev)
    L, A = dense_lower_chol(sol, data)
    got = lower_of(sol, dev.cpu().numpy())
    err = np.linalg.norm(got - L) / np.linalg.norm(L)
    assert err < tol, ("factor", err, desc)
    rhs = rng.standard_normal(n).astype(dtype)
    v = to_dev(rhs)
    sol.solve(dev, v, n, 1)
    X = np.linalg.solve(A, rhs.astype(np.float64))
    err = np.linalg.norm(v.cpu().numpy().astype(np.float64) - X) / np.linalg.norm(X)
    assert err < tol * 50, ("solve", err, desc)
    return desc
