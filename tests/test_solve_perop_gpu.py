"""GPU parity tests of the per-op SolveCtx boundary (MatOps.h:139-184): with bsp_force_per_op the
solver drives the reference's op-by-op loops (Solver.cpp:270-397 for solveL / solveLt, :400-449 for
addMvFrom) over sparseElimSolveL/Lt, symm, solveL, gemv, assembleVec, solveLt, gemvT, assembleVecT
and -- when every span of the range is a lump of its own and nRHS == 1 -- over fragmentedMV /
fragmentedSolveL / fragmentedSolveLt (MatOpsFast.cpp:613-1018).  Same role as
test_per_op_boundary_drives_reference_loop has for the factor side.  Tolerances as
tests/SolveTest.cpp:32-41; the fused path is the second reference."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import solver_random, spd_data, dense_lower_chol, lower_of, to_dev

pytestmark = pytest.mark.gpu

EPS = {np.float64: 1e-8, np.float32: 4e-5}


def _cols(v, n, nrhs):
    return v.cpu().numpy().astype(np.float64).reshape(nrhs, n).T


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nrhs", [1, 5])
def test_solve_per_op_many(dtype, nrhs):
    """SolveTest.Solve*_Many through the op-by-op driver: merged lumps (model 'hip' merges, so
    spans != lumps -> the nine per-op virtuals) and unmerged ones, with and without sparse
    elimination ranges"""
    for i in range(8):
        ranges = [0, 60] if i % 2 else ()
        model = "hip" if i % 4 < 2 else "openblas"
        sol, _, _ = solver_random(57 + i, fill=0.03, elim=(0, 60), ranges=ranges, model=model,
                                  psize_seed=47 + i)
        data = spd_data(sol, 9 + i, dtype=dtype)
        _, A = dense_lower_chol(sol, data)
        n = sol.order()
        rhs = T.random_data(n * nrhs, -1, 1, 37 + i)
        d = to_dev(data)
        sol.factor(d)
        fused = to_dev(rhs.astype(dtype))
        sol.solve(d, fused, n, nrhs)
        with sol.forcePerOp():
            v = to_dev(rhs.astype(dtype))
            sol.solve(d, v, n, nrhs)
            vl = to_dev(rhs.astype(dtype))
            sol.solveL(d, vl, n, nrhs)
            sol.solveLt(d, vl, n, nrhs)
        X = _cols(v, n, nrhs)
        want = np.linalg.solve(A, rhs.reshape(nrhs, n).T)
        assert np.linalg.norm(X - want) < EPS[dtype], (i, np.linalg.norm(X - want))
        assert np.linalg.norm(X - _cols(fused, n, nrhs)) < EPS[dtype], i
        assert np.linalg.norm(X - _cols(vl, n, nrhs)) < EPS[dtype] * 1e-2, i


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_partial_solves_and_add_mv_per_op(dtype):
    """PartialFactorSolveTest.cpp:157-262,340-395 through the op-by-op driver: solveLUpTo / LtUpTo /
    LFrom / LtFrom at a lump boundary with `data` taken as the factor itself, and addMvFrom as the
    symm / gemv / assembleVec / assembleVecT / gemvT sequence"""
    for i in range(4):
        sol, _, _ = solver_random(57 + i, size=215, fill=0.03, elim=(0, 150), psize_seed=47,
                                  pmin=2, pmax=3, model="hip" if i % 2 else "openblas")
        sk = sol.skel()
        ranges = sol.sparseEliminationRanges()
        dense_from = int(ranges[-1]) if len(ranges) else 0
        lump = dense_from + (7 * i) % max(1, sol.numLumps() - dense_from)
        span = int(sk["lumpToSpan"][lump])
        n, nrhs = sol.order(), 3
        bar = int(sk["spanStart"][span])
        data = T.random_data(sol.dataSize(), -1.0, 1.0, 9 + i).astype(dtype)
        sol.damp(data, dtype(0), dtype(3.0))
        Lm = lower_of(sol, data)
        d = to_dev(data)
        tol = 1e-9 if dtype == np.float64 else 2e-5
        rhs = T.random_data(n * nrhs, -1.0, 1.0, 49 + i)
        V = rhs.reshape(nrhs, n).T

        def run(name, *args):
            v = to_dev(rhs.astype(dtype))
            with sol.forcePerOp():
                getattr(sol, name)(d, *args, v, n, nrhs)
            return _cols(v, n, nrhs)

        ref = V.copy()
        if bar:
            ref[:bar] = np.linalg.solve(Lm[:bar, :bar], V[:bar])
            ref[bar:] -= Lm[bar:, :bar] @ ref[:bar]
        assert np.linalg.norm(run("solveLUpTo", span) - ref) / np.linalg.norm(ref) < tol, ("LUpTo", i)
        ref = V.copy()
        if bar:
            ref[:bar] -= Lm[bar:, :bar].T @ ref[bar:]
            ref[:bar] = np.linalg.solve(Lm[:bar, :bar].T, ref[:bar])
        assert np.linalg.norm(run("solveLtUpTo", span) - ref) / np.linalg.norm(ref) < tol, ("LtUpTo", i)
        ref = V.copy()
        if bar < n:
            ref[bar:] = np.linalg.solve(Lm[bar:, bar:], V[bar:])
        assert np.linalg.norm(run("solveLFrom", span) - ref) / np.linalg.norm(ref) < tol, ("LFrom", i)
        ref = V.copy()
        if bar < n:
            ref[bar:] = np.linalg.solve(Lm[bar:, bar:].T, V[bar:])
        assert np.linalg.norm(run("solveLtFrom", span) - ref) / np.linalg.norm(ref) < tol, ("LtFrom", i)

        # addMvFrom on the trailing block, nRHS = 3 (per-op sequence) and nRHS = 1
        sdata = spd_data(sol, 9 + i, beta_factor=2.0, dtype=dtype)
        A = sol.densify(sdata.astype(np.float64), fill_upper_half=True)
        dm = to_dev(sdata)
        for k in (3, 1):
            vin = T.random_data(n * k, -1.0, 1.0, 49 + i)
            vout = T.random_data(n * k, -1.0, 1.0, 149 + i)
            ref = vout.reshape(k, n).T.copy()
            ref[bar:] += 0.75 * (A[bar:, bar:] @ vin.reshape(k, n).T[bar:])
            di, do = to_dev(vin.astype(dtype)), to_dev(vout.astype(dtype))
            with sol.forcePerOp():
                sol.addMvFrom(dm, span, di, n, do, n, k, 0.75)
            got = _cols(do, n, k)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, ("addMv", i, k)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fragmented_ops(dtype):
    """raw skeleton with every span a lump of its own (the layout block-Jacobi / PCG callers use,
    Solver.cpp:299-301,352-355,413-416): nRHS == 1 goes through fragmentedSolveL / SolveLt / MV"""
    cols = T.random_cols(90, 0.05, 71)
    ss = T.columns_to_structure(cols)
    ps = T.random_vec(90, 2, 5, 47)
    # no merges: a cost model whose fixed costs are zero never merges ... simpler: raw skeleton of
    # the filled structure with lumpToSpan = identity
    st = B.Settings(findSparseEliminationRanges=False)
    base = B.create_solver(st, ps, ss)
    sk = base.skel()
    n_sp = base.numSpans()
    # rebuild as unmerged skeleton: one lump per span, columns = the filled block columns
    span_start = np.asarray(sk["spanStart"])
    dense = base.densify(np.ones(base.dataSize()), fill_upper_half=False) != 0
    ptrs, inds = [0], []
    for c in range(n_sp):
        c0 = span_start[c]
        rows = [r for r in range(c, n_sp) if dense[span_start[r], c0]]
        inds.extend(rows)
        ptrs.append(len(inds))
    sol = B.Solver.from_skeleton(span_start, np.arange(n_sp + 1), ptrs, inds)
    assert sol.numLumps() == sol.numSpans()
    data = spd_data(sol, 3, dtype=dtype)
    L, A = dense_lower_chol(sol, data)
    n = sol.order()
    tol = EPS[dtype]
    d = to_dev(data)
    sol.factor(d)
    rhs = T.random_data(n, -1, 1, 5)
    with sol.forcePerOp():
        v = to_dev(rhs.astype(dtype))
        sol.solveL(d, v, n, 1)
        assert np.linalg.norm(_cols(v, n, 1)[:, 0] - np.linalg.solve(L, rhs)) < tol
        v = to_dev(rhs.astype(dtype))
        sol.solveLt(d, v, n, 1)
        assert np.linalg.norm(_cols(v, n, 1)[:, 0] - np.linalg.solve(L.T, rhs)) < tol
        v = to_dev(rhs.astype(dtype))
        sol.solve(d, v, n, 1)
        assert np.linalg.norm(_cols(v, n, 1)[:, 0] - np.linalg.solve(A, rhs)) < tol
        span = n_sp // 3
        bar = int(span_start[span])
        dm = to_dev(data)
        vout = T.random_data(n, -1, 1, 6)
        do = to_dev(vout.astype(dtype))
        sol.addMvFrom(dm, span, to_dev(rhs.astype(dtype)), n, do, n, 1, -0.5)
        ref = vout.copy()
        ref[bar:] += -0.5 * (A[bar:, bar:] @ rhs[bar:])
        assert np.linalg.norm(_cols(do, n, 1)[:, 0] - ref) / np.linalg.norm(ref) < \
            (1e-9 if dtype == np.float64 else 2e-5)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_solve_per_op(dtype):
    """the per-op ops with the batch types (std::vector<T*>, MatOps.h:38-42): equal to the fused
    batched solve"""
    sol, _, _ = solver_random(71, fill=0.03, elim=(0, 60), ranges=[0, 60], model="hip")
    n, nrhs, batch = sol.order(), 2, 3
    mats, rhs = [], []
    for q in range(batch):
        data = spd_data(sol, 20 + q, dtype=dtype)
        d = to_dev(data)
        sol.factor(d)
        mats.append(d)
        rhs.append(T.random_data(n * nrhs, -1, 1, 90 + q).astype(dtype))
    fused = [to_dev(r) for r in rhs]
    sol.solve(mats, fused, n, nrhs)
    vecs = [to_dev(r) for r in rhs]
    with sol.forcePerOp():
        sol.solve(mats, vecs, n, nrhs)
    for q in range(batch):
        a, b = vecs[q].cpu().numpy().astype(np.float64), fused[q].cpu().numpy().astype(np.float64)
        assert np.linalg.norm(a - b) <= (1e-10 if dtype == np.float64 else 1e-4) * \
            max(1.0, np.linalg.norm(b)), q


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pseudo_factor_wide_spans(dtype):
    """pseudoFactorFrom with parameter blocks wider than 16 columns (no width limit in the
    reference, MatOpsCuda.cu:188-233): spans of 3..90 columns, merged into lumps or not"""
    for i, model in enumerate(["openblas", "hip"]):
        cols = T.random_cols(40, 0.12, 11 + i)
        ss = T.columns_to_structure(cols)
        ps = T.random_vec(40, 3, 90, 5 + i)
        assert ps.max() > 64
        st = B.Settings(findSparseEliminationRanges=False,
                        computationModel=None if model == "hip" else
                        __import__("baspacho_amd.csrc_models", fromlist=["x"]).MODEL_OPENBLAS_I7)
        sol = B.create_solver(st, ps, ss)
        sk = sol.skel()
        data = spd_data(sol, 9 + i, beta_factor=2.0, dtype=dtype)
        want = sol.densify(data.astype(np.float64), fill_upper_half=False)
        ss_ = sk["spanStart"]
        for j in range(sol.numSpans()):
            a, b = int(ss_[j]), int(ss_[j + 1])
            Ld = np.linalg.cholesky(want[a:b, a:b])
            want[a:b, a:b] = Ld
            want[b:, a:b] = np.linalg.solve(Ld, want[b:, a:b].T).T
        d = to_dev(data)
        sol.pseudoFactorFrom(d, 0)
        got = lower_of(sol, d.cpu().numpy())
        want = np.tril(want)
        # blocks outside the skeleton do not exist in `got`
        mask = lower_of(sol, np.ones(sol.dataSize())) != 0
        err = np.linalg.norm((got - want)[mask]) / np.linalg.norm(want[mask])
        assert err < (1e-9 if dtype == np.float64 else 2e-5), (model, err)
