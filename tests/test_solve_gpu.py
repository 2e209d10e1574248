"""GPU parity tests of Solver::solve / solveL / solveLt (device path) against dense triangular
solves and the CPU oracle.  Mirrors tests/SolveTest.cpp / CudaSolveTest.cpp of the reference:
tolerances 1e-10 (tiny fixed case) and 1e-8 (random families) in fp64, 1e-5 / 4e-5 in fp32
(tests/SolveTest.cpp:32-41); nRHS = 5 as there."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from oracle import cref
from helpers import solver_random, spd_data, dense_lower_chol, lower_of, to_dev

pytestmark = pytest.mark.gpu

EPS = {np.float64: (1e-10, 1e-8), np.float32: (1e-5, 4e-5)}


def _factor_on_gpu(sol, data):
    d = to_dev(data)
    sol.factor(d)
    return d


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tiny_solve_l_lt(golden, dtype):
    """SolveTest.SolveL / SolveLt (tests/SolveTest.cpp:43-111): the (un-factored) data itself is
    used as a lower-triangular operator"""
    g = golden["tiny_factor"]
    a = g["answer"]
    sol = B.Solver.from_skeleton(g["spanStart"], g["lumpToSpan"], a["groupedPtrs"],
                                 a["groupedInds"])
    data = np.arange(13, 13 + sol.dataSize(), dtype=np.float64)
    sol.damp(data, 5.0, 50.0)
    n, nrhs = sol.order(), 5
    Lop = np.tril(sol.densify(data))
    rhs = T.random_data(n * nrhs, -1, 1, 37)
    Bm = rhs.reshape(nrhs, n).T
    d = to_dev(data.astype(dtype))
    for name, op in (("solveL", Lop), ("solveLt", Lop.T)):
        v = to_dev(rhs.astype(dtype))
        getattr(sol, name)(d, v, n, nrhs)
        got = v.cpu().numpy().astype(np.float64).reshape(nrhs, n).T
        want = np.linalg.solve(op, Bm)
        assert np.linalg.norm(got - want) < EPS[dtype][0] * (1 if dtype == np.float64 else 50), name


@pytest.mark.parametrize("model", ["openblas", "hip"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_solve_many(dtype, model):
    """SolveTest.Solve*_SparseElimAndFactor_Many: factor on the GPU, then solve A x = b"""
    for i in range(12):
        ranges = [0, 60] if i % 2 else ()
        sol, _, _ = solver_random(57 + i, fill=0.03, elim=(0, 60), ranges=ranges, model=model,
                                  psize_seed=47 + i)
        data = spd_data(sol, 9 + i, dtype=dtype)
        _, A = dense_lower_chol(sol, data)
        n, nrhs = sol.order(), 5
        rhs = T.random_data(n * nrhs, -1, 1, 37 + i)
        Bm = rhs.reshape(nrhs, n).T
        d = _factor_on_gpu(sol, data)
        v = to_dev(rhs.astype(dtype))
        sol.solve(d, v, n, nrhs)
        X = v.cpu().numpy().astype(np.float64).reshape(nrhs, n).T
        want = np.linalg.solve(A, Bm)
        assert np.linalg.norm(X - want) < EPS[dtype][1], (i, np.linalg.norm(X - want))
        if dtype == np.float64:
            # and the oracle's solve on the GPU factor
            ref = rhs.copy()
            cref.solve(sol.skel(), d.cpu().numpy(), ref, n, nrhs)
            assert np.linalg.norm(X - ref.reshape(nrhs, n).T) < 1e-10


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_solve(dtype):
    """Solver::solve<std::vector<T*>>: several factored matrices of one structure, one block of
    right-hand sides each; must equal the per-matrix solves (also solveL / solveLt)"""
    sol, _, _ = solver_random(71, fill=0.03, elim=(0, 60), ranges=[0, 60])
    n, nrhs, batch = sol.order(), 3, 4
    mats, rhs = [], []
    for q in range(batch):
        data = spd_data(sol, 20 + q, dtype=dtype)
        mats.append(_factor_on_gpu(sol, data))
        rhs.append(T.random_data(n * nrhs, -1, 1, 90 + q).astype(dtype))
    for name in ("solve", "solveL", "solveLt"):
        single = []
        for q in range(batch):
            v = to_dev(rhs[q])
            getattr(sol, name)(mats[q], v, n, nrhs)
            single.append(v.cpu().numpy())
        vecs = [to_dev(r) for r in rhs]
        getattr(sol, name)(mats, vecs, n, nrhs)
        for q in range(batch):
            got = vecs[q].cpu().numpy()
            # same kernels, same order of operations except for the atomics: equal to rounding
            assert np.linalg.norm(got - single[q]) <= 1e-12 * (1 if dtype == np.float64 else 1e7) * \
                max(1.0, np.linalg.norm(single[q])), (name, q)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_partial_solves(dtype):
    """PartialFactorSolveTest.cpp:157-262: solveLUpTo / solveLtUpTo at a lump boundary, with `data`
    taken as the factor itself (diagonal damped to 3) exactly as the reference's test does; plus
    solveLFrom / solveLtFrom on the trailing block"""
    for i in range(6):
        sol, _, _ = solver_random(57 + i, size=215, fill=0.03, elim=(0, 150), psize_seed=47,
                                  pmin=2, pmax=3)
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
        for j in range(3):
            rhs = T.random_data(n * nrhs, -1.0, 1.0, 49 + j + i)
            V = rhs.reshape(nrhs, n).T

            def run(name, *args):
                v = to_dev(rhs.astype(dtype))
                getattr(sol, name)(d, *args, v, n, nrhs)
                return v.cpu().numpy().astype(np.float64).reshape(nrhs, n).T

            ref = V.copy()
            if bar:
                ref[:bar] = np.linalg.solve(Lm[:bar, :bar], V[:bar])
                ref[bar:] -= Lm[bar:, :bar] @ ref[:bar]
            got = run("solveLUpTo", span)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, ("LUpTo", i, j)

            ref = V.copy()
            if bar:
                ref[:bar] -= Lm[bar:, :bar].T @ ref[bar:]
                ref[:bar] = np.linalg.solve(Lm[:bar, :bar].T, ref[:bar])
            got = run("solveLtUpTo", span)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, ("LtUpTo", i, j)

            ref = V.copy()
            if bar < n:
                ref[bar:] = np.linalg.solve(Lm[bar:, bar:], V[bar:])
            got = run("solveLFrom", span)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, ("LFrom", i, j)

            ref = V.copy()
            if bar < n:
                ref[bar:] = np.linalg.solve(Lm[bar:, bar:].T, V[bar:])
            got = run("solveLtFrom", span)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, ("LtFrom", i, j)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pseudo_factor_and_add_mv(dtype):
    """PartialFactorSolveTest.cpp:296-395: pseudoFactorFrom(0) against the dense per-span
    computation, addMvFrom(span) against the dense symmetric product on the trailing block"""
    for i in range(6):
        sol, _, _ = solver_random(57 + i, size=215, fill=0.03, elim=(0, 150), psize_seed=47,
                                  pmin=2, pmax=3)
        sk = sol.skel()
        n = sol.order()
        data = spd_data(sol, 9 + i, beta_factor=2.0, dtype=dtype)
        tol = 1e-9 if dtype == np.float64 else 2e-5
        # ---- pseudo-factor of every span
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
        # (only blocks present in the skeleton exist in `got`; the dense computation fills none in)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < tol, ("pseudo", i)
        # ---- addMvFrom on the trailing block
        ranges = sol.sparseEliminationRanges()
        dense_from = int(ranges[-1]) if len(ranges) else 0
        lump = dense_from + (7 * i) % max(1, sol.numLumps() - dense_from)
        span = int(sk["lumpToSpan"][lump])
        bar = int(ss_[span])
        A = sol.densify(data.astype(np.float64), fill_upper_half=True)
        nrhs = 3
        vin = T.random_data(n * nrhs, -1.0, 1.0, 49 + i)
        vout = T.random_data(n * nrhs, -1.0, 1.0, 149 + i)
        ref = vout.reshape(nrhs, n).T.copy()
        ref[bar:] += 0.75 * (A[bar:, bar:] @ vin.reshape(nrhs, n).T[bar:])
        dm = to_dev(data)
        di, do = to_dev(vin.astype(dtype)), to_dev(vout.astype(dtype))
        sol.addMvFrom(dm, span, di, n, do, n, nrhs, 0.75)
        got = do.cpu().numpy().astype(np.float64).reshape(nrhs, n).T
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, ("addMv", i)


@pytest.mark.parametrize("precond", ["lowerprec", "jacobi", "identity"])
def test_pcg_on_schur_complement(precond):
    """mixed direct / iterative solve as examples/Optimizer.h:710-747: factorUpTo(span), reduce the
    right-hand side, PCG on the Schur complement with the preconditioners of
    examples/Preconditioner.h, back-substitute; against the direct solve"""
    import torch
    from baspacho_amd import pcg
    sol, _, _ = solver_random(63, size=215, fill=0.03, elim=(0, 150), psize_seed=47, pmin=2, pmax=3)
    sk = sol.skel()
    ranges = sol.sparseEliminationRanges()
    dense_from = int(ranges[-1]) if len(ranges) else 0
    lump = dense_from + (sol.numLumps() - dense_from) // 3
    span = int(sk["lumpToSpan"][lump])
    n = sol.order()
    bar = int(sk["spanStart"][span])
    data = spd_data(sol, 9, beta_factor=2.0)
    A = sol.densify(data, fill_upper_half=True)
    b = T.random_data(n, -1.0, 1.0, 49)
    want = np.linalg.solve(A, b)
    d = to_dev(data)
    sol.factorUpTo(d, span)                      # Schur complement in the bottom-right blocks
    v = to_dev(b.copy())
    sol.solveLUpTo(d, span, v, n, 1)             # reduced right-hand side in v[bar:]
    op = pcg.TrailingOperator(sol, d, span)
    M = {"lowerprec": lambda: pcg.LowerPrecSolvePrecond(sol, d, span),
         "jacobi": lambda: pcg.BlockJacobiPrecond(sol, d, span),
         "identity": lambda: pcg.IdentityPrecond()}[precond]()
    x_tail = torch.zeros(n - bar, dtype=torch.float64, device=v.device)
    iters, res = pcg.PCG(M, op, 1e-12, 400).solve(x_tail, v[bar:].clone())
    assert res <= 1e-12, (precond, iters, res)
    if precond == "lowerprec":
        assert iters <= 4, iters                 # a single-precision factor is nearly exact
    v[bar:] = x_tail
    sol.solveLtUpTo(d, span, v, n, 1)            # back-substitution through the factored part
    got = v.cpu().numpy()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-9, precond


def test_solve_l_then_lt_equals_solve_with_stride():
    """solveL followed by solveLt == solve; leading dimension larger than the order"""
    sol, _, _ = solver_random(63, fill=0.03, elim=(0, 60), ranges=[0, 60])
    data = spd_data(sol, 5)
    n, nrhs, ld = sol.order(), 3, sol.order() + 7
    d = _factor_on_gpu(sol, data)
    rhs = np.zeros(ld * nrhs)
    for q in range(nrhs):
        rhs[q * ld:q * ld + n] = T.random_data(n, -1, 1, 40 + q)
    v1, v2 = to_dev(rhs), to_dev(rhs)
    sol.solve(d, v1, ld, nrhs)
    sol.solveL(d, v2, ld, nrhs)
    sol.solveLt(d, v2, ld, nrhs)
    a, b = v1.cpu().numpy(), v2.cpu().numpy()
    assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(a)
    pad = np.concatenate([a[q * ld + n:(q + 1) * ld] for q in range(nrhs)])
    assert np.all(pad == 0)  # the padding rows are untouched


def test_solve_bal_like_residual():
    """bundle-adjustment shaped problem: ||A x - b|| / ||b|| after factor + solve on the GPU"""
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=60, num_pts=6000, band=8, seed=5)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 6000])
    data = spd_data(sol, 11, beta_factor=1.2)
    n = sol.order()
    b = T.random_data(n, -1, 1, 3)
    d = _factor_on_gpu(sol, data)
    v = to_dev(b)
    sol.solve(d, v, n, 1)
    x = v.cpu().numpy()
    # A x through the oracle's block-sparse probe: ||L(L^T x) - A x|| is the factor residual;
    # here we check A x = b with A applied by the dense matrix of this mid-size problem
    A = sol.densify(data, fill_upper_half=True)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12


def test_mixed_precision_refinement():
    """BASELINE config 5 in small: fp32 factor + fp64 iterative refinement reaches ||r||/||b|| <
    1e-10 on a bundle-adjustment shaped problem (stand-in for BAL-1723)"""
    import torch
    from baspacho_amd.refine import solve_refined
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=80, num_pts=8000, band=10, seed=7)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 8000])
    data = spd_data(sol, 13, beta_factor=1.2)
    b = T.random_data(sol.order(), -1, 1, 4)
    x, iters, hist = solve_refined(sol, to_dev(data), to_dev(b), tol=1e-10)
    assert hist[-1] < 1e-10, hist
    assert iters <= 6, hist
    assert hist[0] < 1e-4  # the fp32 solve alone is already a decent solution
    A = sol.densify(data, fill_upper_half=True)
    assert np.linalg.norm(A @ x.cpu().numpy() - b) / np.linalg.norm(b) < 1e-10


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_partial_factor_and_solves(dtype):
    """the batch forms of the partial operations (Solver.cpp:491-519: factorUpTo / factorFrom /
    solveLUpTo / solveLtUpTo for std::vector<T*>; solveLFrom / solveLtFrom with them): each entry of
    a batch must come out as the single-matrix call on the same data does"""
    sol, _, _ = solver_random(63, size=215, fill=0.03, elim=(0, 150), psize_seed=47, pmin=2, pmax=3)
    sk = sol.skel()
    ranges = sol.sparseEliminationRanges()
    dense_from = int(ranges[-1]) if len(ranges) else 0
    lump = (dense_from + sol.numLumps()) // 2
    span = int(sk["lumpToSpan"][lump])
    n, nrhs, nb = sol.order(), 2, 4
    tol = 1e-12 if dtype == np.float64 else 1e-5
    datas = [spd_data(sol, 30 + q, dtype=dtype) for q in range(nb)]
    # factorUpTo + factorFrom, batched, against the single calls
    single = []
    for h in datas:
        d = to_dev(h)
        sol.factorUpTo(d, span)
        mid = d.cpu().numpy().copy()
        sol.factorFrom(d, span)
        single.append((mid, d.cpu().numpy().copy()))
    devs = [to_dev(h) for h in datas]
    sol.factorUpTo(devs, span)
    for q in range(nb):
        ref = single[q][0]
        assert np.linalg.norm(devs[q].cpu().numpy() - ref) / np.linalg.norm(ref) < tol, ("UpTo", q)
    sol.factorFrom(devs, span)
    for q in range(nb):
        ref = single[q][1]
        assert np.linalg.norm(devs[q].cpu().numpy() - ref) / np.linalg.norm(ref) < tol, ("From", q)
    # partial solves on the factors
    rhs = [T.random_data(n * nrhs, -1.0, 1.0, 70 + q).astype(dtype) for q in range(nb)]
    for name in ("solveLUpTo", "solveLtUpTo", "solveLFrom", "solveLtFrom"):
        refs = []
        for q in range(nb):
            v = to_dev(rhs[q])
            getattr(sol, name)(devs[q], span, v, n, nrhs)
            refs.append(v.cpu().numpy().astype(np.float64))
        vs = [to_dev(r) for r in rhs]
        getattr(sol, name)(devs, span, vs, n, nrhs)
        for q in range(nb):
            got = vs[q].cpu().numpy().astype(np.float64)
            assert np.linalg.norm(got - refs[q]) / max(np.linalg.norm(refs[q]), 1e-30) < tol * 10, (name, q)


def test_partial_solve_boundary_inside_an_elimination_range_is_an_error():
    """Solver.cpp:281-290: solveLUpTo / solveLFrom with a boundary strictly inside a sparse-elimination
    range is a precondition failure ("Check failed"), on the fused device path as on the op-by-op one;
    a boundary at the end of the ranges or beyond is fine (tools/stress.py checks those numerically)"""
    sol, _, _ = solver_random(61, fill=0.03, elim=(0, 40))
    ranges = sol.sparseEliminationRanges()
    assert len(ranges) >= 2 and ranges[1] - ranges[0] >= 2
    sk = sol.skel()
    inside = int(sk["lumpToSpan"][int(ranges[0]) + 1])
    data = spd_data(sol, 3)
    d = to_dev(data)
    sol.factor(d)
    n = sol.order()
    v = to_dev(T.random_data(n, -1, 1, 5))
    with pytest.raises(RuntimeError, match="Check failed"):
        sol.solveLUpTo(d, inside, v, n, 1)
    with pytest.raises(RuntimeError, match="Check failed"):
        sol.solveLFrom(d, inside, v, n, 1)
    with pytest.raises(RuntimeError, match="Check failed"):
        sol.solveLtFrom(d, inside, v, n, 1)
    edge = int(sk["lumpToSpan"][int(ranges[-1])])
    sol.solveLUpTo(d, edge, v, n, 1)
    sol.solveLFrom(d, edge, v, n, 1)


@pytest.mark.parametrize("cond", [1e4, 1e8])
def test_block_solve_by_inverses_on_ill_conditioned_lump(monkeypatch, cond):
    """the wide-lump solves multiply by explicitly inverted 64 x 64 diagonal blocks
    (BSP_SOLVE_INV, default on) where the reference substitutes (trsm / trsv, MatOpsCuda.cu:550-566):
    the error of that form grows with the condition of a diagonal block of L.  One dense lump of 700
    columns with a GRADED spectrum (A = Q diag(1 .. 1/cond) Q^T): both forms against numpy and the
    oracle's substitution, with the bound each is held to -- backward error 1e-14 for substitution,
    1e-16 x sqrt(cond) x 100 for the inverses (a diagonal block of L is at worst as ill conditioned as
    sqrt(cond(A)))."""
    import torch
    n = 700
    rng = np.random.default_rng(5)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q * np.logspace(0, -np.log10(cond), n)) @ Q.T
    A = 0.5 * (A + A.T)
    b = rng.standard_normal(n)
    want = np.linalg.solve(A, b)
    res = {}
    for inv in ("1", "0"):
        monkeypatch.setenv("BSP_SOLVE_INV", inv)
        ss = T.columns_to_structure([set(range(i, n)) for i in range(n)])
        sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
        assert sol.numLumps() == 1
        # one lump: the data vector IS the (permuted) n x n matrix, row-major; undo the ordering
        perm = np.asarray(sol.paramToSpan())
        Ap = np.empty_like(A)
        Ap[np.ix_(perm, perm)] = A
        d = to_dev(Ap.reshape(-1).copy())
        sol.factor(d)
        v = to_dev(b.copy())
        sol.solve(d, v, n, 1)
        torch.cuda.synchronize()
        x = v.cpu().numpy()
        eta = np.linalg.norm(A @ x - b, np.inf) / (np.linalg.norm(A, np.inf) * np.linalg.norm(x, np.inf)
                                                  + np.linalg.norm(b, np.inf))
        res[inv] = (eta, np.linalg.norm(x - want) / np.linalg.norm(want))
    assert res["0"][0] < 1e-14, res
    assert res["1"][0] < 1e-14 * np.sqrt(cond), res
    # forward error: both within cond x eps of numpy's solution
    assert res["0"][1] < 1e-13 * cond and res["1"][1] < 1e-13 * cond, res
