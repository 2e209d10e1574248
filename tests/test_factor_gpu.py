"""GPU parity tests of the HIP factor path, through the C ABI, against the CPU oracle and the
dense Cholesky (numpy).  Mirrors the reference's FactorTest / CudaFactorTest /
BatchedCudaFactorTest / CreateSolverTest protocols and tolerances:
   fp64: 1e-10 (fixed tiny case), 1e-8 (random families);  fp32: 1e-5 / 5e-5
   (tests/FactorTest.cpp:32-41).  Errors are Frobenius norms of the lower triangle."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from oracle import cref
from oracle import skel as OS
from helpers import solver_random, spd_data, dense_lower_chol, lower_of, to_dev

pytestmark = pytest.mark.gpu

EPS = {np.float64: (1e-10, 1e-8), np.float32: (1e-5, 5e-5)}


def _gpu_factor(sol, data):
    d = to_dev(data)
    sol.factor(d)
    return d.cpu().numpy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tiny_coalesced_factor(golden, dtype):
    """FactorTest.CoalescedFactor / CudaFactor.CoalescedFactor (tests/FactorTest.cpp:43-65)"""
    g = golden["tiny_factor"]
    a = g["answer"]
    sol = B.Solver.from_skeleton(g["spanStart"], g["lumpToSpan"], a["groupedPtrs"],
                                 a["groupedInds"])
    data = np.arange(13, 13 + sol.dataSize(), dtype=np.float64)
    sol.damp(data, float(g["damp_alpha"]), float(g["damp_beta"]))
    got = _gpu_factor(sol, data.astype(dtype))
    err = np.linalg.norm(lower_of(sol, got) - np.array(a["L_lower"]))
    assert err < EPS[dtype][0] * (1 if dtype == np.float64 else 20), err


@pytest.mark.parametrize("model", ["openblas", "hip"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_coalesced_factor_many(dtype, model):
    """FactorTest.CoalescedFactor_Many (tests/FactorTest.cpp:75-107): 20 random patterns"""
    for i in range(20):
        sol, _, _ = solver_random(57 + i, model=model, find_ranges=False)
        data = spd_data(sol, 9 + i, dtype=dtype)
        L, _ = dense_lower_chol(sol, data)
        got = _gpu_factor(sol, data)
        err = np.linalg.norm(lower_of(sol, got) - L)
        assert err < EPS[dtype][1], (i, err)
        # and against the oracle (run in fp64 on the same input), on the same skeleton
        ref = data.astype(np.float64)
        cref.factor(sol.skel(), ref, sol.sparseEliminationRanges())
        assert np.linalg.norm(lower_of(sol, got) - lower_of(sol, ref)) < EPS[dtype][1]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sparse_elim_many(dtype):
    """FactorTest.SparseElim_Many (tests/FactorTest.cpp:129-167): doElimination alone, compare
    the eliminated columns only"""
    for i in range(20):
        # the reference runs EliminationTree on the un-reordered pattern; through createSolver
        # the same situation is a user-given elimination range over the independent set
        sol, _, _ = solver_random(57 + i, fill=0.03, elim=(0, 60), ranges=[0, 60])
        ranges = sol.sparseEliminationRanges()
        assert len(ranges) >= 2
        data = spd_data(sol, 9 + i, dtype=dtype)
        L, _ = dense_lower_chol(sol, data)
        d = to_dev(data)
        sol.doElimination(d, 0)
        got = d.cpu().numpy()
        ncol = int(sol.skel()["lumpStart"][ranges[1]])
        err = np.linalg.norm((lower_of(sol, got) - L)[:, :ncol])
        assert err < EPS[dtype][0] * (1 if dtype == np.float64 else 5), (i, err)
        # the Schur-complement part must match the oracle's doElimination too
        # (whole matrix incl. the un-factored Schur complement, entries ~ order: relative norm)
        ref = data.astype(np.float64)
        cref.do_elimination(sol.skel(), ref, int(ranges[0]), int(ranges[1]))
        rel = np.linalg.norm(lower_of(sol, got) - lower_of(sol, ref)) / np.linalg.norm(lower_of(sol, ref))
        assert rel < (1e-13 if dtype == np.float64 else 1e-6), (i, rel)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sparse_elim_and_factor_many(dtype):
    """FactorTest.SparseElimAndFactor_Many (tests/FactorTest.cpp:185-219)"""
    for i in range(20):
        sol, _, _ = solver_random(57 + i, fill=0.03, elim=(0, 60),
                                  ranges=[0, 60] if i % 2 else ())
        if i % 2:
            assert len(sol.sparseEliminationRanges()) >= 2
        data = spd_data(sol, 9 + i, dtype=dtype)
        L, _ = dense_lower_chol(sol, data)
        got = _gpu_factor(sol, data)
        err = np.linalg.norm(lower_of(sol, got) - L)
        assert err < EPS[dtype][1], (i, err)


def test_given_elim_ranges_and_wide_elim_lumps():
    """user-given elimination range whose lumps are wider than the small-kernel limit (16):
    they must go through the panel kernels inside doElimination"""
    size = 60
    cols = T.make_independent_elim_set(T.random_cols(size, 0.08, 91), 0, 25)
    ss = T.columns_to_structure(cols)
    ps = np.where(np.arange(size) % 5 == 0, 23, 4).astype(np.int64)  # some 23-wide params
    sol = B.create_solver(B.Settings(), ps, ss, [0, 25])
    assert sol.sparseEliminationRanges().tolist()[:2] == [0, 25]
    data = spd_data(sol, 5)
    L, _ = dense_lower_chol(sol, data)
    got = _gpu_factor(sol, data)
    assert np.linalg.norm(lower_of(sol, got) - L) < 1e-8


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_factor(dtype):
    """BatchedCudaFactorTest (tests/BatchedCudaFactorTest.cpp:44-158): several matrices of one
    structure in one call; every matrix must match its own dense Cholesky"""
    batch_sizes = T.random_vec(6, 3, 31, 37)
    for i, bs in enumerate(batch_sizes):
        sol, _, _ = solver_random(57 + i, fill=0.03, elim=(0, 60) if i % 2 else None)
        datas = [spd_data(sol, 100 * i + q, dtype=dtype) for q in range(int(bs))]
        devs = [to_dev(d) for d in datas]
        sol.factor(devs)
        for q in range(int(bs)):
            L, _ = dense_lower_chol(sol, datas[q])
            err = np.linalg.norm(lower_of(sol, devs[q].cpu().numpy()) - L)
            assert err < EPS[dtype][1], (i, q, err)


@pytest.mark.product_defaults
@pytest.mark.parametrize("bs", [2, 5, 17])
def test_batch_as_concurrent_halves(bs, monkeypatch):
    """round 5: a batch is factored as two concurrent halves (second half on the auxiliary stream,
    its own slice of the per-matrix scratch) when the plan keeps its lookahead units in line --
    here forced for small batches and odd splits; structures with an elimination range, a multi-
    panel tree and a one-panel chain (staging buffer and inverted diagonal blocks per matrix)"""
    monkeypatch.setenv("BSP_SUB_BATCH_MIN", "2")
    n = 300
    cols = [set(range(i, min(n, i + 3))) | (set(range(200, n)) if i >= 150 else set()) for i in range(n)]
    for sol in (solver_random(61, fill=0.03, elim=(0, 60))[0],
                B.create_solver(B.Settings(), np.full(n, 2, dtype=np.int64), T.columns_to_structure(cols))):
        datas = [spd_data(sol, 300 + q) for q in range(bs)]
        devs = [to_dev(d) for d in datas]
        before = sol.runCounters()["sub_batches_enqueued"]
        sol.factor(devs)
        # (the run counters say which path ran: two halves, each enqueued on a stream of its own)
        assert sol.runCounters()["sub_batches_enqueued"] == before + 2, "the batch was not split"
        for q in range(bs):
            L, _ = dense_lower_chol(sol, datas[q])
            err = np.linalg.norm(lower_of(sol, devs[q].cpu().numpy()) - L) / np.linalg.norm(L)
            assert err < 1e-12, (q, err)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_wide_dense_lump(dtype):
    """a batch through the one-panel-level chain kernels (chain step, staging buffer and inverted
    diagonal blocks have one slot per matrix) and the lookahead units: three matrices of one wide
    supernode, each against its own dense factor"""
    n = 1100
    ss = T.columns_to_structure([set(range(i, n)) for i in range(n)])
    sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
    datas = [spd_data(sol, 40 + q, beta_factor=1.2) for q in range(3)]
    devs = [to_dev(d.astype(dtype)) for d in datas]
    sol.factor(devs)
    tol = 1e-10 if dtype == np.float64 else 5e-5
    for q in range(3):
        _, A = dense_lower_chol(sol, datas[q])
        Lg = lower_of(sol, devs[q].cpu().numpy()).astype(np.float64)
        assert np.linalg.norm(Lg @ Lg.T - A) / np.linalg.norm(A) < tol, q


@pytest.mark.parametrize("elim_set,last_ids", [(False, False), (True, False), (False, True),
                                               (True, True)])
def test_create_solver_policies(elim_set, last_ids):
    """CreateSolverTest (tests/CreateSolverTest.cpp:43-140): fill policies x given elimination
    ranges x elimLastIds; a partial factor leaves the Schur complement in the bottom-right"""
    for i in range(5):
        n_params = 215
        cols = T.random_cols(n_params, 0.03, 57 + i)
        ranges = []
        if elim_set:
            cols = T.make_independent_elim_set(cols, 0, 150)
            ranges = [0, 90]
        else:
            cols = T.make_independent_elim_set(cols, 0, 60)
        last = set()
        if last_ids:
            last = {105, 123, 165, 194, 209, 214}
            if not elim_set:
                last |= {0, 30, 49, 87}
        ss = T.columns_to_structure(cols)
        ps = T.random_vec(n_params, 2, 3, 47)
        policies = [B.AddFillComplete]
        if not last_ids:
            policies += [B.AddFillForAutoElims, B.AddFillForGivenElims, B.AddFillNone]
        for pol in policies:
            sol = B.create_solver(B.Settings(addFillPolicy=pol), ps, ss, ranges, last)
            up_to = sol.canFactorUpToSpan()
            if pol == B.AddFillComplete:
                assert up_to == n_params
            if pol == B.AddFillNone:
                assert up_to == 0
            if pol == B.AddFillForGivenElims and elim_set:
                assert up_to == 90
            data = spd_data(sol, 9 + 4 * i, beta_factor=2.0)
            L, A = dense_lower_chol(sol, data)
            bar = int(sol.skel()["spanStart"][up_to])
            want = np.tril(L.copy())
            # bottom-right = Schur complement of the un-factored part
            want[bar:, bar:] = np.tril(A[bar:, bar:] - L[bar:, :bar] @ L[bar:, :bar].T)
            d = to_dev(data)
            sol.factorUpTo(d, up_to)
            got = lower_of(sol, d.cpu().numpy())
            assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-9, (i, pol)
            if last:
                s = len(last)
                assert sol.skel()["spanOffsetInLump"][sol.numSpans() - s] == 0
                p2s = sol.paramToSpan()
                assert all(p2s[e] >= sol.numSpans() - s for e in last)


def test_factor_up_to_then_from():
    """PartialFactorSolveTest: factorUpTo(k) followed by factorFrom(k) == factor"""
    sol, _, _ = solver_random(61, fill=0.03, elim=(0, 60))
    data = spd_data(sol, 3)
    L, _ = dense_lower_chol(sol, data)
    ranges = sol.sparseEliminationRanges()
    sk = sol.skel()
    dense_from = int(ranges[-1]) if len(ranges) else 0
    mid_lump = (dense_from + sol.numLumps()) // 2
    mid_span = int(sk["lumpToSpan"][mid_lump])
    d = to_dev(data)
    sol.factorUpTo(d, mid_span)
    sol.factorFrom(d, mid_span)
    assert np.linalg.norm(lower_of(sol, d.cpu().numpy()) - L) < 1e-8


def test_bal_like_and_grid_against_oracle():
    """larger structured cases (sizes the oracle finishes in seconds): a bundle-adjustment shaped
    problem with a given point-elimination range, and a grid; compared with the oracle on the
    same skeleton via the relative residual of the factors"""
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=60, num_pts=6000, band=8, seed=5)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 6000])
    data = spd_data(sol, 11, beta_factor=1.2)
    ref = data.copy()
    cref.factor(sol.skel(), ref, sol.sparseEliminationRanges())
    got = _gpu_factor(sol, data)
    mask = sol.lowerMask()  # strictly-upper entries of diagonal blocks are unspecified
    assert np.linalg.norm((got - ref)[mask]) / np.linalg.norm(ref[mask]) < 1e-12

    ss = T.gen_grid(24, 24, 1.0, 2, 37)
    sol = B.create_solver(B.Settings(), np.full(24 * 24, 3), ss)
    data = spd_data(sol, 12, beta_factor=1.2)
    L, A = dense_lower_chol(sol, data)
    got = _gpu_factor(sol, data)
    Lg = lower_of(sol, got)
    assert np.linalg.norm(Lg @ Lg.T - A) / np.linalg.norm(A) < 1e-10
    assert np.linalg.norm(Lg - L) < 1e-8


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lump_widths_around_panel_and_block_boundaries(dtype):
    """explicit skeletons (Solver(skel, ...), Solver.h:37-38): one lump of width W (spans of
    mixed sizes) followed by rows that belong to two further lumps, for W around the panel (64)
    and outer-block (256) boundaries -- exercises the direct / fused-potrf / split-update launches
    and their hand-over to the generic task-list launches (boards of the lump column)"""
    for W in (1, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 321, 515):
        # spans: sizes cycle 1,2,3 until the lump holds W columns; then 2 lumps of 40 and 30
        sizes, tot = [], 0
        while tot < W:
            s = min(1 + len(sizes) % 3, W - tot)
            sizes.append(s)
            tot += s
        n0 = len(sizes)
        tail1 = [3] * 13 + [1]      # 40
        tail2 = [2] * 15            # 30
        sizes = sizes + tail1 + tail2
        span_start = np.concatenate([[0], np.cumsum(sizes)])
        l2s = [0, n0, n0 + len(tail1), n0 + len(tail1) + len(tail2)]
        nspans = len(sizes)
        # lump 0 sees every span of the tail except a few (sparse boards), lumps 1 and 2 are dense
        rows0 = list(range(n0)) + [q for q in range(n0, nspans) if (q * 7 + W) % 5 != 0]
        rows1 = list(range(n0, nspans))
        rows2 = list(range(n0 + len(tail1), nspans))
        col_ptr = np.cumsum([0, len(rows0), len(rows1), len(rows2)])
        sol = B.Solver.from_skeleton(span_start, l2s, col_ptr, rows0 + rows1 + rows2)
        data = spd_data(sol, 100 + W, beta_factor=1.5, dtype=dtype)
        L, A = dense_lower_chol(sol, data)
        got = lower_of(sol, _gpu_factor(sol, data))
        err = np.linalg.norm(got @ got.T - A) / np.linalg.norm(A)
        assert err < (1e-10 if dtype == np.float64 else 5e-5), (W, err)
        assert np.linalg.norm(got - L) < EPS[dtype][1] * (1 if dtype == np.float64 else 10), W
        # and the device solve on the same skeleton (panels narrower than 64, boards)
        d = to_dev(data)
        sol.factor(d)
        n = sol.order()
        rhs = T.random_data(n * 2, -1, 1, 7 + W)
        v = to_dev(rhs.astype(dtype))
        sol.solve(d, v, n, 2)
        X = v.cpu().numpy().astype(np.float64).reshape(2, n).T
        want = np.linalg.solve(A, rhs.reshape(2, n).T)
        assert np.linalg.norm(X - want) / np.linalg.norm(want) < (1e-10 if dtype == np.float64 else 1e-4), W


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_last_lump_ending_in_short_outer_blocks(dtype):
    """ONE dense lump (no rows below it) whose last outer block is short: widths 256 k + r for
    r = 1 .. 64 .. 255.  With r = 64 and k >= 2 the block-last step of the last full block updates a
    single tile, is therefore not fused with the next potrf, and used to apply the block's first
    three panels to that tile twice (they had applied their update early): relative error ~1e-5 in
    the last 64 x 64 block, found by tools/stress.py seed 9426 in round 3."""
    for W in (320, 576, 832, 1600, 513, 575, 577, 639, 640, 704, 768, 1088 + 8, 1344):
        sizes, tot = [], 0
        while tot < W:
            s = min(8 if W % 8 == 0 else 1 + len(sizes) % 5, W - tot)
            sizes.append(s)
            tot += s
        nparam = len(sizes)
        cols = [list(range(c, nparam)) for c in range(nparam)]
        ss = T.columns_to_structure(cols)
        sol = B.create_solver(B.Settings(), np.asarray(sizes, dtype=np.int64), ss, [])
        data = spd_data(sol, 11 + W, dtype=dtype)
        L, A = dense_lower_chol(sol, data)
        got = lower_of(sol, _gpu_factor(sol, data))
        err = np.linalg.norm(got - L) / np.linalg.norm(L)
        assert err < (1e-12 if dtype == np.float64 else 2e-5), (W, err)
        tail = np.linalg.norm(got[-64:, -64:] - L[-64:, -64:]) / np.linalg.norm(L[-64:, -64:])
        assert tail < (1e-11 if dtype == np.float64 else 1e-4), (W, tail)


@pytest.mark.parametrize("knob", ["BSP_NO_LOOKAHEAD=1", "BSP_DUE_STREAM=0", "BSP_BULK_AHEAD=0",
                                  "BSP_LOOKAHEAD_MIN_GF=1000"])
def test_schedule_variants(monkeypatch, knob):
    """every optimisation of the launch schedule can be switched off (the environment is read
    when the solver is created); each fallback must still factor correctly"""
    k, v = knob.split("=")
    monkeypatch.setenv(k, v)
    n = 700
    ss = T.columns_to_structure([set(range(i, n)) for i in range(n)])
    sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
    data = spd_data(sol, 21, beta_factor=1.2)
    _, A = dense_lower_chol(sol, data)
    Lg = lower_of(sol, _gpu_factor(sol, data))
    assert np.linalg.norm(Lg @ Lg.T - A) / np.linalg.norm(A) < 1e-10
    # (the switch was honoured: no launch went to the auxiliary streams with the lookahead off or
    #  priced out; the 700-wide lump is too small for the lookahead to pay in any variant)
    if k in ("BSP_NO_LOOKAHEAD", "BSP_LOOKAHEAD_MIN_GF"):
        assert sol.runCounters()["lookahead_forks"] == 0
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=60, num_pts=6000, band=8, seed=5)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 6000])
    data = spd_data(sol, 11, beta_factor=1.2)
    ref = data.copy()
    cref.factor(sol.skel(), ref, sol.sparseEliminationRanges())
    got = _gpu_factor(sol, data)
    mask = sol.lowerMask()
    assert np.linalg.norm((got - ref)[mask]) / np.linalg.norm(ref[mask]) < 1e-12


@pytest.mark.parametrize("split", ["1", "0"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_split_k_tiles_of_small_levels(monkeypatch, split, dtype):
    """round 5: the block-wide update tiles (K = 128 .. 256) of a small multi-panel level are listed again
    as slices of at most 96 source columns that accumulate with atomics (BSP_SPLIT_K=0: one tile each).
    Four independent dense blocks of 130 / 130 / 200 / 200 parameters over a common separator: the
    levels hold four or two panels, the block-closing ones update the separator with K = 130 / 200."""
    monkeypatch.setenv("BSP_SPLIT_K", split)
    widths, sep = [130, 130, 200, 200], 150
    n = sum(widths) + sep
    cols, base = [], 0
    for w in widths:
        for i in range(w):
            cols.append(set(range(base + i, base + w)) | set(range(n - sep, n)))
        base += w
    for i in range(sep):
        cols.append(set(range(n - sep + i, n)))
    sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), T.columns_to_structure(cols))
    for bs in (1, 3):
        datas = [spd_data(sol, 70 + q, beta_factor=1.2) for q in range(bs)]
        devs = [to_dev(d.astype(dtype)) for d in datas]
        before = sol.runCounters()["split_lists_used"]
        sol.factor(devs if bs > 1 else devs[0])
        used = sol.runCounters()["split_lists_used"] - before
        assert (used > 0) == (split == "1"), ("split-K lists used by %d launches" % used, split)
        for q in range(bs):
            _, A = dense_lower_chol(sol, datas[q])
            Lg = lower_of(sol, devs[q].cpu().numpy()).astype(np.float64)
            assert np.linalg.norm(Lg @ Lg.T - A) / np.linalg.norm(A) < (1e-10 if dtype == np.float64 else 5e-5), q


@pytest.mark.parametrize("fold", ["1", "0"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_potrf_folded_into_the_trsm_launch_of_small_levels(monkeypatch, fold, dtype):
    """round 6 (trsmPanelPotrf): a multi-panel level whose trsm launch is one round of workgroups drops its
    potrf launch -- every row tile factors its own copy of the diagonal block in LDS, ONE potrfPanel launch
    stores the factors after the last level.  Same arithmetic per panel: the factor must be the one of the
    three-launch form to rounding.  Four independent blocks over a separator (levels of four and two panels, ragged
    widths), one matrix and a batch; the counter says which form ran; the solve reads the stored factors."""
    widths, sep = [130, 97, 200, 70], 150
    n = sum(widths) + sep
    cols, base = [], 0
    for w in widths:
        for i in range(w):
            cols.append(set(range(base + i, base + w)) | set(range(n - sep, n)))
        base += w
    for i in range(sep):
        cols.append(set(range(n - sep + i, n)))
    ss = T.columns_to_structure(cols)
    results = {}
    for f in (fold, "0"):
        monkeypatch.setenv("BSP_POTRF_IN_TRSM", f)
        sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
        for bs in (1, 3):
            datas = [spd_data(sol, 90 + q, beta_factor=1.2) for q in range(bs)]
            devs = [to_dev(d.astype(dtype)) for d in datas]
            before = sol.runCounters()["potrf_folded_levels"]
            sol.factor(devs if bs > 1 else devs[0])
            folded = sol.runCounters()["potrf_folded_levels"] - before
            assert (folded > 0) == (f == "1"), (folded, f)
            for q in range(bs):
                got = devs[q].cpu().numpy()
                results.setdefault((bs, q), []).append(got)
                _, A = dense_lower_chol(sol, datas[q])
                Lg = lower_of(sol, got).astype(np.float64)
                assert np.linalg.norm(Lg @ Lg.T - A) / np.linalg.norm(A) < (1e-10 if dtype == np.float64 else 5e-5), q
            if bs == 1 and dtype == np.float64:
                rhs = np.random.default_rng(5).standard_normal(n)
                v = to_dev(rhs.copy())
                sol.solve(devs[0], v, n, 1)
                X = np.linalg.solve(dense_lower_chol(sol, datas[0])[1], rhs)
                assert np.linalg.norm(v.cpu().numpy() - X) / np.linalg.norm(X) < 1e-10
    # (the same arithmetic per panel; the separator's columns take atomics from the four blocks in whatever
    #  order the launch runs them, so two runs of EITHER form agree to rounding, not bit for bit)
    for key, (a, b) in results.items():
        d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b).max()
        assert d < (1e-13 if dtype == np.float64 else 1e-5), ("folded and three-launch factors differ", key, d)


@pytest.mark.parametrize("ahead", ["0", "0.6", "100"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lookahead_units_over_many_outer_blocks(monkeypatch, ahead, dtype):
    """a dense lump of seven outer blocks: the side stream applies (source block -> column block)
    units in deadline order -- lazily (BSP_BULK_AHEAD=0: every column block receives all its
    pending source blocks at once, accumulated with atomics), by the default budget, or eagerly
    (100: every unit as soon as its source block is done); widths that leave a ragged last block"""
    monkeypatch.setenv("BSP_BULK_AHEAD", ahead)
    n = 1700
    ss = T.columns_to_structure([set(range(i, n)) for i in range(n)])
    sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
    data = spd_data(sol, 23, beta_factor=1.2)
    _, A = dense_lower_chol(sol, data)
    Lg = lower_of(sol, _gpu_factor(sol, data.astype(dtype))).astype(np.float64)
    tol = 1e-10 if dtype == np.float64 else 5e-5
    assert np.linalg.norm(Lg @ Lg.T - A) / np.linalg.norm(A) < tol


def test_wide_dense_lump_residual():
    """one wide supernode (multi-panel, intra-lump trailing updates): north-star residual
    ||L L^T - A|| / ||A|| < 1e-10 in fp64"""
    n = 700
    cols = [set(range(i, n)) for i in range(n)]
    ss = T.columns_to_structure(cols)
    sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
    data = spd_data(sol, 21, beta_factor=1.2)
    _, A = dense_lower_chol(sol, data)
    Lg = lower_of(sol, _gpu_factor(sol, data))
    assert np.linalg.norm(Lg @ Lg.T - A) / np.linalg.norm(A) < 1e-10


def test_errors_are_loud():
    """wrong-size data and CPU backends must fail with an exception, not fall back"""
    sol, ps, ss = solver_random(57)
    import torch
    with pytest.raises(ValueError):
        sol.factor(torch.zeros(3, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        sol.factor(torch.zeros(sol.dataSize(), dtype=torch.float64))  # host memory
    with pytest.raises(RuntimeError):
        B.create_solver(B.Settings(backend=B.BackendFast), ps, ss)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_per_op_boundary_drives_reference_loop(dtype):
    """the reference's driver loop (Solver.cpp:164-219: doElimination, then per target lump
    prepareAssemble / saveSyrkGemm+assemble per board / potrf / trsm) executed op by op through
    the NumericCtx boundary of the HIP backend; must equal the dense Cholesky and the fused path"""
    for i in range(6):
        ranges = [0, 60] if i % 2 else ()
        sol, _, _ = solver_random(57 + i, fill=0.03, elim=(0, 60), ranges=ranges,
                                  model="openblas" if i < 4 else "hip")
        data = spd_data(sol, 9 + i, dtype=dtype)
        L, _ = dense_lower_chol(sol, data)
        d = to_dev(data)
        sol.factorPerOp(d)
        got = d.cpu().numpy()
        err = np.linalg.norm(lower_of(sol, got) - L)
        assert err < EPS[dtype][1], (i, err)
        fused = _gpu_factor(sol, data)
        mask = sol.lowerMask()
        rel = np.linalg.norm((got - fused)[mask].astype(np.float64)) / np.linalg.norm(fused[mask])
        assert rel < (1e-13 if dtype == np.float64 else 2e-6), (i, rel)


def test_per_op_wide_lump():
    """per-op potrf/trsm on a multi-panel (two-level blocked) lump with rows below"""
    n = 300
    cols = [set(range(i, n)) for i in range(n)]
    cols += [set([n + j]) | set() for j in range(40)]
    for j in range(40):
        cols[(7 * j) % n].add(n + j)   # a few rows below the wide dense part
    ss = T.columns_to_structure(cols)
    sol = B.create_solver(B.Settings(), np.ones(n + 40, dtype=np.int64), ss)
    data = spd_data(sol, 33, beta_factor=1.2)
    L, _ = dense_lower_chol(sol, data)
    d = to_dev(data)
    sol.factorPerOp(d)
    assert np.linalg.norm(lower_of(sol, d.cpu().numpy()) - L) < 1e-8


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_chain_like_structures(dtype):
    """pose-chain shaped problems (block-tridiagonal + loop closures): createSolver's chain
    contraction turns them into a few sparse-elimination ranges + a small dense tail; the device
    factor must match the dense Cholesky of the permuted matrix and the oracle"""
    rng = np.random.default_rng(4)
    for n, closures, bs in [(400, 0, 3), (700, 5, 6), (300, 12, 2)]:
        rows = list(range(1, n))
        cols = list(range(0, n - 1))
        for _ in range(closures):
            a, b = sorted(rng.choice(n, 2, replace=False))
            rows.append(int(b))
            cols.append(int(a))
        ss = T.structure_from_pairs(n, np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64))
        sol = B.create_solver(B.Settings(), np.full(n, bs, dtype=np.int64), ss)
        assert len(sol.sparseEliminationRanges()) >= 3
        data = spd_data(sol, 21 + n, dtype=dtype)
        L, _ = dense_lower_chol(sol, data)
        got = _gpu_factor(sol, data)
        # (relative: these matrices are 5-10x the order of the reference's random families, for
        #  which its absolute tolerances were chosen)
        nL = np.linalg.norm(L)
        assert np.linalg.norm(lower_of(sol, got) - L) / nL < EPS[dtype][1] * 0.1, (n, closures)
        ref = data.astype(np.float64)
        cref.factor(sol.skel(), ref, sol.sparseEliminationRanges())
        assert np.linalg.norm(lower_of(sol, got) - lower_of(sol, ref)) / nL < EPS[dtype][1] * 0.1


@pytest.mark.product_defaults
def test_lookahead_choice_follows_the_plan():
    """product defaults: a plan whose lookahead units are too small for their forks runs them in
    line (GRID: 1.46 against 1.73 ms with side streams), a wide dense lump keeps the side streams;
    both orders give the same factor"""
    ss = T.gen_grid(40, 40, 1.0, 2, 37)
    sol = B.create_solver(B.Settings(), np.full(1600, 3), ss)
    st = sol.planStats()
    assert st["num_fork_levels"] == 0 or st["deferred_flops"] / st["num_fork_levels"] < 3e9
    data = spd_data(sol, 12, beta_factor=1.2)
    ref = data.copy()
    cref.factor(sol.skel(), ref, sol.sparseEliminationRanges())
    got = _gpu_factor(sol, data)
    mask = sol.lowerMask()
    assert np.linalg.norm((got - ref)[mask]) / np.linalg.norm(ref[mask]) < 1e-12
    # one dense lump of 7800 columns (the width of BAL-871's camera block): ~4.7 GF per fork
    n = 2600
    full = T.structure_from_pairs(n, *np.tril_indices(n, -1))
    sol = B.create_solver(B.Settings(findSparseEliminationRanges=False), np.full(n, 3), full)
    st = sol.planStats()
    assert st["num_fork_levels"] > 0 and st["deferred_flops"] / st["num_fork_levels"] >= 3e9


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_in_register_cholesky_and_row_solves(dtype):
    """MathUtilsTest on the device (tests/MathUtilsTest.cpp:21-75): the scalar
    cholesky / solveUpperT of the reference live inside the sparse-elimination factor kernels here
    (elimFactorTiny for lumps up to 4 wide, elimFactorSmall up to 16).  Independent lumps of every
    width 1..16 with one block of rows below each, eliminated alone (doElimination), against numpy
    per lump: diagonal block = chol(A_ll), rows below = A_rl L^-T; tolerance 1e-7 as the reference
    (fp32: 1e-4)"""
    tol = 1e-7 if dtype == np.float64 else 1e-4
    for n in range(1, 17):
        K, m = 70, 5          # K independent params of width n, one tail param of m rows
        sizes = np.array([n] * K + [m], dtype=np.int64)
        ss = T.structure_from_pairs(K + 1, np.full(K, K, dtype=np.int64), np.arange(K, dtype=np.int64))
        sol = B.create_solver(B.Settings(findSparseEliminationRanges=False), sizes, ss, [0, K])
        assert sol.sparseEliminationRanges().tolist() == [0, K]
        data = spd_data(sol, 100 + n, dtype=dtype)
        A = sol.densify(data.astype(np.float64), fill_upper_half=True)
        d = to_dev(data)
        sol.doElimination(d, 0)
        got = sol.densify(d.cpu().numpy().astype(np.float64))
        p2s = sol.paramToSpan()
        ss_ = sol.skel()["spanStart"]
        t0 = int(ss_[p2s[K]])
        for k in range(K):
            c0 = int(ss_[p2s[k]])
            L = np.linalg.cholesky(A[c0:c0 + n, c0:c0 + n])
            assert np.linalg.norm(np.tril(got[c0:c0 + n, c0:c0 + n]) - L) < tol, (n, k)
            X = np.linalg.solve(L, A[t0:t0 + m, c0:c0 + n].T).T      # rows * L^-T
            assert np.linalg.norm(got[t0:t0 + m, c0:c0 + n] - X) < tol, (n, k)


def test_empty_and_trivial_structures():
    """edge cases: no parameters at all (factor and solve are no-ops on empty device buffers), a
    single 1 x 1 block, a block-diagonal matrix (every lump eliminated, no update at all), fp64 + fp32"""
    import torch
    ss = T.columns_to_structure([])
    sol = B.create_solver(B.Settings(), np.zeros(0, dtype=np.int64), ss, [])
    assert sol.order() == 0 and sol.dataSize() == 0
    d = torch.zeros(0, dtype=torch.float64, device="cuda")
    sol.factor(d)
    sol.solve(d, torch.zeros(0, dtype=torch.float64, device="cuda"), 0, 1)
    for dtype in (np.float64, np.float32):
        sol = B.create_solver(B.Settings(), np.ones(1, dtype=np.int64), T.columns_to_structure([[0]]), [])
        d = to_dev(np.array([9.0], dtype=dtype))
        sol.factor(d)
        assert abs(float(d.cpu()[0]) - 3.0) < 1e-6
        v = to_dev(np.array([6.0], dtype=dtype))
        sol.solve(d, v, 1, 1)
        assert abs(float(v.cpu()[0]) - 6.0 / 9.0) < 1e-6
        for find in (True, False):
            sol = B.create_solver(B.Settings(findSparseEliminationRanges=find), np.full(50, 3, dtype=np.int64),
                                  T.columns_to_structure([[c] for c in range(50)]), [])
            data = spd_data(sol, 4, dtype=dtype)
            L, A = dense_lower_chol(sol, data)
            got = lower_of(sol, _gpu_factor(sol, data))
            assert np.linalg.norm(got - L) / np.linalg.norm(L) < (1e-13 if dtype == np.float64 else 1e-5)


def test_indefinite_input_fails_silently_with_nans():
    """numerical breakdown is silent in the reference (NaN / Inf in `data`, no exception, no hang:
    SURVEY 8b "Errors"); the same here, for a wide dense lump (chain + lookahead units, the bounded
    yield spins) and for a structure with an elimination range"""
    import torch
    nparam = 1100 // 4
    cols = [list(range(c, nparam)) for c in range(nparam)]
    sol = B.create_solver(B.Settings(), np.full(nparam, 4, dtype=np.int64), T.columns_to_structure(cols), [])
    data = T.random_data(sol.dataSize(), -1.0, 1.0, 5)      # no damping: indefinite
    d = to_dev(data)
    sol.factor(d)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(d).all())
    v = to_dev(T.random_data(sol.order(), -1, 1, 6))
    sol.solve(d, v, sol.order(), 1)
    torch.cuda.synchronize()
    sol2, _, _ = solver_random(61, fill=0.05, elim=(0, 40))
    d2 = to_dev(T.random_data(sol2.dataSize(), -1.0, 1.0, 7))
    sol2.factor(d2)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(d2).all())
