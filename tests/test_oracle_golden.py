"""CPU: the oracle and the product's host logic against the reference's known answers."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from oracle import cref
from oracle import skel as OS
from oracle import structure as ST
from helpers import columns_to_csc, solver_random, spd_data, dense_lower_chol, lower_of


def _skel_inputs(g):
    ptrs, inds = columns_to_csc([set(c) for c in g["columnParams"]])
    return g["spanStart"], g["lumpToSpan"], ptrs, inds


def test_oracle_skeleton_literals(golden):
    g = golden["skeleton_9span"]
    sk = OS.build_skeleton(*_skel_inputs(g))
    for name, want in g["expected"].items():
        assert sk[name].tolist() == want, name


def test_product_skeleton_literals(golden):
    """CoalescedBlockMatrixTest.BasicAssertions on the C++ host library"""
    g = golden["skeleton_9span"]
    sol = B.Solver.from_skeleton(*_skel_inputs(g))
    sk = sol.skel()
    for name, want in g["expected"].items():
        assert sk[name].tolist() == want, name
    assert sol.order() == 16 and sol.dataSize() == 97


def test_densify_literals(golden):
    g, d = golden["skeleton_9span"], golden["densify_9span"]
    sk = OS.build_skeleton(*_skel_inputs(g))
    data = np.arange(13, 13 + OS.data_size(sk), dtype=np.float64)
    want = np.array(d["expected_lower_16x16"], dtype=np.float64)
    assert np.linalg.norm(OS.densify(sk, data) - want) < 1e-10
    want2 = np.array(d["expected_full_from_span1_15x15"], dtype=np.float64)
    assert np.linalg.norm(OS.densify(sk, data, True, 1) - want2) < 1e-10
    sol = B.Solver.from_skeleton(*_skel_inputs(g))
    assert np.linalg.norm(sol.densify(data) - want) < 1e-10
    assert np.linalg.norm(sol.densify(data, True, 1) - want2) < 1e-10


def test_damp(golden):
    g = golden["skeleton_9span"]
    sk = OS.build_skeleton(*_skel_inputs(g))
    sol = B.Solver.from_skeleton(*_skel_inputs(g))
    data = np.arange(13, 13 + OS.data_size(sk), dtype=np.float64)
    mat = OS.densify(sk, data)
    d1 = OS.damp(sk, data.copy(), 2.0, 100.0)
    d2 = sol.damp(data.copy(), 2.0, 100.0)
    want = mat.copy()
    want[np.diag_indices_from(want)] = np.diag(mat) * 3.0 + 100.0
    assert np.linalg.norm(OS.densify(sk, d1) - want) < 1e-5
    assert np.array_equal(d1, d2)


def test_transpose_and_sym_permutation_literals(golden):
    g = golden["transpose"]
    assert ST.transpose(g["ptrs"], g["inds"]) == (g["expected_ptrs"], g["expected_inds"])
    t = B.SparseStructure(g["ptrs"], g["inds"]).transpose()
    assert t.ptrs.tolist() == g["expected_ptrs"] and t.inds.tolist() == g["expected_inds"]

    g = golden["sym_permutation"]
    ss = B.SparseStructure(g["ptrs"], g["inds"])
    up = ss.symmetricPermutation(g["mapPerm"], lowerHalf=False)
    assert up.ptrs.tolist() == g["expected_upper_ptrs"]
    assert up.inds.tolist() == g["expected_upper_inds"]
    lo = ss.symmetricPermutation(g["mapPerm"], lowerHalf=True)
    assert lo.ptrs.tolist() == g["expected_lower_ptrs"]
    assert lo.inds.tolist() == g["expected_lower_inds"]
    assert ST.symmetric_permutation(g["ptrs"], g["inds"], g["mapPerm"], False) == \
        (g["expected_upper_ptrs"], g["expected_upper_inds"])


def test_fill_reducing_permutation_quality(golden):
    """SparseStructureTest.FillReducingPermutation: nnz(L) <= 130 ("should be 120")"""
    g = golden["amd_24"]
    ss = B.SparseStructure(g["ptrs"], g["inds"]).clear()
    perm = ss.fillReducingPermutation()
    assert sorted(perm.tolist()) == list(range(24))
    inv = np.empty(24, dtype=np.int64)
    inv[perm] = np.arange(24)
    filled = ss.symmetricPermutation(inv, lowerHalf=False).addFullEliminationFill()
    assert len(filled.inds) <= g["max_fill_nnz"]


@pytest.mark.parametrize("size,fill", [(10, 0.15), (20, 0.23), (30, 0.3), (40, 0.15)])
def test_elimination_fill_vs_naive(size, fill):
    """SparseStructureTest.{IndependentEliminationFill,FullEliminationFill}: the product's fill
    algorithms against the naive set algorithm the reference uses as ground truth"""
    cols_orig = T.random_cols(size, fill, 37 + size)
    full = [set(c) for c in cols_orig]
    ST.naive_add_elimination_entries(full, 0, size)
    got = T.columns_to_structure(cols_orig).addFullEliminationFill()
    want = T.columns_to_structure(full)
    assert got.ptrs.tolist() == want.ptrs.tolist() and got.inds.tolist() == want.inds.tolist()
    for start in range(0, size * 2 // 3, 3):
        for end in range(start + 3, size, 3):
            cols = T.make_independent_elim_set(cols_orig, start, end)
            ss = T.columns_to_structure(cols)
            naive = [set(c) for c in cols]
            ST.naive_add_elimination_entries(naive, start, end)
            want = T.columns_to_structure(naive)
            got = ss.addIndependentEliminationFill(start, end)
            assert got.ptrs.tolist() == want.ptrs.tolist()
            assert got.inds.tolist() == want.inds.tolist()


def test_oracle_tiny_factor(golden):
    """FactorTest.CoalescedFactor: oracle C restatement vs the committed dense answer"""
    g = golden["tiny_factor"]
    a = g["answer"]
    sk = OS.build_skeleton(g["spanStart"], g["lumpToSpan"], a["groupedPtrs"], a["groupedInds"])
    assert OS.data_size(sk) == a["dataSize"]
    L = np.array(a["L_lower"])
    for dtype, tol in ((np.float64, g["tol_f64"]), (np.float32, g["tol_f32"])):
        data = np.arange(13, 13 + a["dataSize"], dtype=np.float64)
        OS.damp(sk, data, g["damp_alpha"], g["damp_beta"])
        data = data.astype(dtype)
        cref.factor(sk, data)
        got = np.tril(OS.densify(sk, data.astype(np.float64)))
        assert np.linalg.norm(got - L) < tol * (1 if dtype == np.float64 else 20)


@pytest.mark.parametrize("seed", range(6))
def test_oracle_random_factor_and_solve(seed):
    """FactorTest.CoalescedFactor_Many + SolveTest protocol on the oracle, on skeletons built by
    the PRODUCT's symbolic analysis (so this also validates createSolver): 1e-8"""
    sol, _, _ = solver_random(57 + seed)
    sk = sol.skel()
    data = spd_data(sol, 9 + seed)
    L, A = dense_lower_chol(sol, data)
    fact = data.copy()
    cref.factor(sk, fact, sol.sparseEliminationRanges())
    assert np.linalg.norm(lower_of(sol, fact) - L) < 1e-8
    n, nrhs = sol.order(), 3
    b = T.random_data(n * nrhs, -1, 1, 37 + seed)
    x = b.copy()
    cref.solve(sk, fact, x, n, nrhs)
    X, Bm = x.reshape(nrhs, n).T, b.reshape(nrhs, n).T
    assert np.linalg.norm(A @ X - Bm) / np.linalg.norm(Bm) < 1e-10


@pytest.mark.parametrize("seed", range(4))
def test_oracle_sparse_elim(seed):
    """FactorTest.SparseElim_Many / SparseElimAndFactor_Many on the oracle"""
    sol, _, _ = solver_random(57 + seed, fill=0.03, elim=(0, 60), ranges=[0, 60])
    ranges = sol.sparseEliminationRanges()
    assert len(ranges) >= 2
    sk = sol.skel()
    data = spd_data(sol, 9 + seed)
    L, _ = dense_lower_chol(sol, data)
    only = data.copy()
    cref.do_elimination(sk, only, int(ranges[0]), int(ranges[1]))
    ncol = int(sk["lumpStart"][ranges[1]])
    assert np.linalg.norm((lower_of(sol, only) - L)[:, :ncol]) < 1e-10
    full = data.copy()
    cref.factor(sk, full, ranges)
    assert np.linalg.norm(lower_of(sol, full) - L) < 1e-8


@pytest.mark.parametrize("seed", range(3))
def test_blas_baseline_matches_plain_oracle(seed):
    """the BackendFast restatement (cpu_baseline of bench.py) against the plain-loop oracle"""
    sol, _, _ = solver_random(57 + seed, fill=0.03, elim=(0, 60), ranges=[0, 60] if seed else ())
    sk = sol.skel()
    data = spd_data(sol, 9 + seed)
    a, b = data.copy(), data.copy()
    cref.factor(sk, a, sol.sparseEliminationRanges())
    cref.blas_factor(sk, b, sol.sparseEliminationRanges())
    assert np.linalg.norm(lower_of(sol, a) - lower_of(sol, b)) < 1e-9


def test_oracle_math_utils():
    """MathUtilsTest (tests/MathUtilsTest.cpp:21-75): the scalar cholesky / solveUpperT / solveUpper
    of the oracle against numpy, same sizes, damping and tolerance (1e-7) as the reference"""
    import ctypes
    lib = cref.lib()
    n = 10
    dp = ctypes.POINTER(ctypes.c_double)
    A = T.random_data(n * n, -1.0, 1.0, 37).reshape(n, n).copy()
    A[np.arange(n), np.arange(n)] += n * 1.3
    sym = np.tril(A) + np.tril(A, -1).T       # the kernels read the lower triangle
    L = np.linalg.cholesky(sym)
    got = A.copy()
    lib.orc_cholesky_f64(got.ctypes.data_as(dp), ctypes.c_int64(n), ctypes.c_int64(n))
    assert np.linalg.norm(np.tril(got) - L) < 1e-7
    # solveUpperT: v <- v L^-T ; solveUpper: v <- v L^-1, L = lower triangle of a damped matrix
    M = T.random_data(n * n, -1.0, 1.0, 37).reshape(n, n).copy()
    M[np.arange(n), np.arange(n)] += n * 0.3
    Lm = np.tril(M)
    v = T.random_data(n, -1.0, 1.0, 39)
    a = v.copy()
    lib.orc_solve_upper_t_f64(M.ctypes.data_as(dp), ctypes.c_int64(n), ctypes.c_int64(n), a.ctypes.data_as(dp))
    assert np.linalg.norm(a - np.linalg.solve(Lm, v)) < 1e-7          # v L^-T = (L^-1 v^T)^T
    b = v.copy()
    lib.orc_solve_upper_f64(M.ctypes.data_as(dp), ctypes.c_int64(n), ctypes.c_int64(n), b.ctypes.data_as(dp))
    assert np.linalg.norm(b - np.linalg.solve(Lm.T, v)) < 1e-7        # v L^-1 = (L^-T v^T)^T
