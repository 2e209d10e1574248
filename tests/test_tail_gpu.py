"""Persistent, flag-synchronised Cholesky of the tail of a wide root lump (csrc/hip_tail_kernel.h,
round 6): ONE launch factors the last outer blocks of a lump that has nothing below it, in place of
the per-panel chain launches (cusolverDnDpotrf + cublasDtrsm + cublasDgemm, MatOpsCuda.cu:508-590).
Parity against numpy's Cholesky at the reference's tolerances (tests/FactorTest.cpp:32-41) for tails
of 2 .. all-but-one outer blocks, ragged last panels, fp64 / fp32, a batch; the run counters assert
that the tail launch was the path taken."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import spd_data, dense_lower_chol, lower_of, to_dev

pytestmark = pytest.mark.gpu


def _dense_solver(W, span=8):
    sizes = [span] * (W // span) + ([W % span] if W % span else [])
    nparam = len(sizes)
    cols = [list(range(c, nparam)) for c in range(nparam)]
    ss = T.columns_to_structure(cols)
    return B.create_solver(B.Settings(findSparseEliminationRanges=False), np.asarray(sizes, dtype=np.int64), ss, [])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("blocks", [2, 4, 100])
def test_tail_of_one_wide_lump(monkeypatch, dtype, blocks):
    monkeypatch.setenv("BSP_TAIL_BLOCKS", str(blocks))
    for W in (1536, 1599, 1601, 1664, 2049, 2500):
        sol = _dense_solver(W, span=8 if W % 2 == 0 else 7)
        data = spd_data(sol, 3 + W, dtype=dtype)
        L, A = dense_lower_chol(sol, data)
        dev = to_dev(data)
        before = sol.runCounters()["tail_launches"]
        sol.factor(dev)
        got = lower_of(sol, dev.cpu().numpy())
        assert sol.runCounters()["tail_launches"] == before + 1, "the persistent tail was not the path taken"
        err = np.linalg.norm(got - L) / np.linalg.norm(L)
        assert err < (1e-12 if dtype == np.float64 else 2e-5), (W, blocks, err)
        tail = np.linalg.norm(got[-64:, -64:] - L[-64:, -64:]) / np.linalg.norm(L[-64:, -64:])
        assert tail < (1e-11 if dtype == np.float64 else 1e-4), (W, blocks, tail)
        # and a solve on that factor (the tail's panels are ordinary panels to the solve paths)
        n = sol.order()
        rhs = np.random.default_rng(W).standard_normal(n)
        v = to_dev(rhs.astype(dtype))
        sol.solve(dev, v, n, 1)
        X = np.linalg.solve(A, rhs)
        serr = np.linalg.norm(v.cpu().numpy().astype(np.float64) - X) / np.linalg.norm(X)
        assert serr < (1e-10 if dtype == np.float64 else 1e-3), (W, blocks, serr)


def test_tail_panels_in_the_multi_launch_solves(monkeypatch):
    """with the persistent sweeps off, the solves walk the tail's panels through the block and level
    kernels (a last outer block of ONE panel is a plain level: found by the 65 600-wide lump)"""
    monkeypatch.setenv("BSP_TAIL_BLOCKS", "6")
    monkeypatch.setenv("BSP_SOLVE_SWEEP", "0")
    for W in (1536 + 64, 1536 + 40, 2048 + 1):
        sol = _dense_solver(W)
        data = spd_data(sol, 3 + W)
        _, A = dense_lower_chol(sol, data)
        dev = to_dev(data)
        sol.factor(dev)
        assert sol.runCounters()["tail_launches"] == 1
        n = sol.order()
        rhs = np.random.default_rng(W).standard_normal((2, n))
        v = to_dev(rhs.reshape(-1).copy())
        sol.solve(dev, v, n, 2)
        X = np.linalg.solve(A, rhs.T)
        err = np.linalg.norm(v.cpu().numpy().reshape(2, n).T - X) / np.linalg.norm(X)
        assert err < 1e-10, (W, err)


def test_a_batch_keeps_the_level_schedule(monkeypatch):
    """the tail is a latency device for ONE matrix (profiles/r06_tail_batches.txt): factor() of a batch runs
    the second plan, without it; one matrix on the same Solver takes the tail"""
    monkeypatch.setenv("BSP_TAIL_BLOCKS", "3")
    sol = _dense_solver(1700)
    mats, dense = [], []
    for q in range(3):
        data = spd_data(sol, 40 + q)
        mats.append(to_dev(data))
        dense.append(dense_lower_chol(sol, data)[0])
    before = sol.runCounters()["tail_launches"]
    sol.factor(mats)
    assert sol.runCounters()["tail_launches"] == before
    one = to_dev(spd_data(sol, 40))
    sol.factor(one)
    assert sol.runCounters()["tail_launches"] == before + 1
    assert np.linalg.norm(lower_of(sol, one.cpu().numpy()) - dense[0]) / np.linalg.norm(dense[0]) < 1e-12
    for q in range(3):
        got = lower_of(sol, mats[q].cpu().numpy())
        assert np.linalg.norm(got - dense[q]) / np.linalg.norm(dense[q]) < 1e-12, q


def test_no_tail_for_a_narrow_lump_when_switched_off(monkeypatch):
    monkeypatch.setenv("BSP_TAIL_BLOCKS", "4")
    monkeypatch.setenv("BSP_TAIL_NARROW_MIN", "0")
    sol = _dense_solver(1100)  # five outer blocks: below the minimum of the wide-lump rule
    data = spd_data(sol, 5)
    dev = to_dev(data)
    sol.factor(dev)
    assert sol.runCounters()["tail_launches"] == 0
    L, _ = dense_lower_chol(sol, data)
    got = lower_of(sol, dev.cpu().numpy())
    assert np.linalg.norm(got - L) / np.linalg.norm(L) < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tail_of_a_narrow_root_lump(dtype):
    """round 6, verdict item 1b: a root lump of 2 .. 5 outer blocks (GRID 82x82: 990 columns) hands
    everything after its first block to the persistent tail when ONE matrix is factored; a batch takes
    the second plan without it (the tail loses there: profiles/r06_tail_narrow.txt)"""
    for W in (577, 600, 640, 822, 990, 1100, 1279, 1281, 1535):
        sol = _dense_solver(W, span=8 if W % 2 == 0 else 7)
        want = 1 if W - 256 > 5 * 64 else 0
        assert (sol.planStats()["num_tail_panels"] > 0) == bool(want), W
        data = spd_data(sol, 11 + W, dtype=dtype)
        L, A = dense_lower_chol(sol, data)
        dev = to_dev(data)
        before = sol.runCounters()["tail_launches"]
        sol.factor(dev)
        assert sol.runCounters()["tail_launches"] == before + want, W
        got = lower_of(sol, dev.cpu().numpy())
        err = np.linalg.norm(got - L) / np.linalg.norm(L)
        assert err < (1e-12 if dtype == np.float64 else 2e-5), (W, err)
        worst = np.abs(got - L).max() / np.abs(L).max()
        assert worst < (1e-11 if dtype == np.float64 else 1e-4), (W, worst)
        # a batch of the same solver: the plan without a tail
        mats = [to_dev(spd_data(sol, 70 + q, dtype=dtype)) for q in range(3)]
        dense = [dense_lower_chol(sol, m.cpu().numpy())[0] for m in mats]
        before = sol.runCounters()["tail_launches"]
        sol.factor(mats)
        assert sol.runCounters()["tail_launches"] == before, W
        for q in range(3):
            got = lower_of(sol, mats[q].cpu().numpy())
            assert np.linalg.norm(got - dense[q]) / np.linalg.norm(dense[q]) < (1e-12 if dtype == np.float64 else 2e-5), (W, q)


def _cliques(widths, span=8):
    cols, base = [], 0
    for w in widths:
        k = w // span
        cols += [list(range(base + c, base + k)) for c in range(k)]
        base += k
    return np.full(base, span, dtype=np.int64), T.columns_to_structure(cols)


@pytest.mark.parametrize("widths", [(1600, 1600), (1600, 3200), (900, 900), (832, 1216, 640)])
def test_tails_in_a_forest_of_wide_roots(widths):
    """A tail level is ONE launch that factors nothing but the tail, so a lump may hand columns to it only
    if those panels are alone in their levels.  Block-diagonal problems (two cliques = two root lumps on
    the same levels) used to take the tail path for panels that shared a level and came out WRONG
    (found in round 6 on the reference's MERI family); the plan now checks the level occupancy first."""
    sizes, ss = _cliques(widths)
    sol = B.create_solver(B.Settings(findSparseEliminationRanges=False), sizes, ss, [])
    data = spd_data(sol, 21 + len(widths))
    L, _ = dense_lower_chol(sol, data)
    dev = to_dev(data)
    sol.factor(dev)
    got = lower_of(sol, dev.cpu().numpy())
    err = np.abs(got - L).max() / np.abs(L).max()
    assert err < 1e-11, (widths, err, sol.planStats()["num_tail_panels"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_whole_narrow_root_lump_behind_its_children(dtype):
    """a narrow root lump that FOLLOWS other levels (two dense blocks of 500, each coupled to 150 columns of a 900-column separator:
    the structure of GRID 82x82's top) is the persistent tail as a whole -- first outer block included:
    15 panels (all but the first block: 11); the level before it is the children's last one"""
    widths, sep, link = [500, 500], 900, 150
    n = sum(widths) + sep
    cols, base = [], 0
    for k, w in enumerate(widths):
        lo = n - sep + (0 if k == 0 else sep - link)
        for i in range(w):
            cols.append(set(range(base + i, base + w)) | set(range(lo, lo + link)))
        base += w
    for i in range(sep):
        cols.append(set(range(n - sep + i, n)))
    sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), T.columns_to_structure(cols))
    assert sol.planStats()["num_tail_panels"] == 15
    data = spd_data(sol, 77, beta_factor=1.2, dtype=dtype)
    L, A = dense_lower_chol(sol, data)
    dev = to_dev(data)
    before = sol.runCounters()["tail_launches"]
    sol.factor(dev)
    assert sol.runCounters()["tail_launches"] == before + 1
    got = lower_of(sol, dev.cpu().numpy())
    assert np.linalg.norm(got - L) / np.linalg.norm(L) < (1e-12 if dtype == np.float64 else 2e-5)
    assert np.abs(got - L).max() / np.abs(L).max() < (1e-11 if dtype == np.float64 else 1e-4)
    rhs = np.random.default_rng(3).standard_normal(n)
    v = to_dev(rhs.astype(dtype))
    sol.solve(dev, v, n, 1)
    X = np.linalg.solve(A, rhs)
    assert np.linalg.norm(v.cpu().numpy().astype(np.float64) - X) / np.linalg.norm(X) < (1e-10 if dtype == np.float64 else 1e-3)
    mats = [to_dev(spd_data(sol, 80 + q, beta_factor=1.2, dtype=dtype)) for q in range(2)]
    dense = [dense_lower_chol(sol, m.cpu().numpy())[0] for m in mats]
    sol.factor(mats)  # (a batch: the plan without tails)
    assert sol.runCounters()["tail_launches"] == before + 1
    for q in range(2):
        got = lower_of(sol, mats[q].cpu().numpy())
        assert np.linalg.norm(got - dense[q]) / np.linalg.norm(dense[q]) < (1e-12 if dtype == np.float64 else 2e-5), q


def test_meri_family_with_a_narrow_root():
    """the reference's 40_MERI problem (Bench.cpp:355-360): its 822-column root lump takes the narrow
    tail; vector residual probe against the CPU oracle"""
    import bench
    name = [k for k in bench.ref_suite_problems() if k.startswith("40_MERI")][0]
    sizes, ss = bench.ref_suite_problems()[name](37)
    sol = B.create_solver(B.Settings(findSparseEliminationRanges=True), sizes, ss)
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
    sol.damp(h, 0.0, sol.order() * 1.2)
    dev = to_dev(h)
    sol.factor(dev)
    assert bench.residual_probe(sol, h, dev, nprobe=2) < 1e-13


def test_backend_options_reach_the_schedule_without_the_environment():
    """Settings.hipOptions -> bsp_hip_options -> HipBackendOptions (csrc/backend_options.h): the switches
    of the tail launch and of the persistent sweeps, set by the CALLER (no environment variable)"""
    sizes = [8] * 200  # one dense lump of 1600 columns
    ss = T.columns_to_structure([list(range(c, len(sizes))) for c in range(len(sizes))])
    for opts, want_tail, want_sweep in (({}, True, True), ({"tail_blocks": 0}, False, True),
                                        ({"solve_sweep": 0, "tail_blocks": 3}, True, False)):
        st = B.Settings(findSparseEliminationRanges=False, hipOptions=opts or None)
        sol = B.create_solver(st, np.asarray(sizes, dtype=np.int64), ss, [])
        data = spd_data(sol, 8)
        _, A = dense_lower_chol(sol, data)
        dev = to_dev(data)
        sol.factor(dev)
        n = sol.order()
        rhs = np.random.default_rng(2).standard_normal(n)
        v = to_dev(rhs.copy())
        sol.solve(dev, v, n, 1)
        X = np.linalg.solve(A, rhs)
        assert np.linalg.norm(v.cpu().numpy() - X) / np.linalg.norm(X) < 1e-10, opts
        c = sol.runCounters()
        assert (c["tail_launches"] > 0) == want_tail and (c["sweep_launches"] > 0) == want_sweep, (opts, c)
        assert (sol.planStats()["num_tail_panels"] > 0) == want_tail
