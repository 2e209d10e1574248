"""Persistent, flag-synchronised solve sweeps over wide lumps (csrc/hip_sweep_kernels.h, round 6):
ONE launch per direction walks a whole run of one-panel levels instead of two launches per 256
columns (the cublas trsm / gemv chains of MatOpsCuda.cu:1093-1181).  Parity against dense numpy
solves at the reference's tolerances (tests/SolveTest.cpp:32-41), for run widths around the panel
(64) and sweep-block (192) boundaries, with and without rows below the run, several right-hand
sides, batches, both directions on their own -- and every test ASSERTS through the run counters
that the sweep was the path taken.  The watchdog test injects a spine that never publishes."""
import time

import numpy as np
import pytest
import torch

import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import spd_data, dense_lower_chol, lower_of, to_dev

pytestmark = pytest.mark.gpu

TOL = {np.float64: 1e-9, np.float32: 2e-4}


def _wide_solver(W, tail, span=8):
    """one dense lump of width W (spans of `span`, a ragged last one), optionally followed by a second
    lump of `tail` columns that most of the spans of the first one see (rows below the run)"""
    sizes = [span] * (W // span) + ([W % span] if W % span else [])
    n0 = len(sizes)
    sizes = sizes + [7] * (tail // 7)
    nparam = len(sizes)
    cols = [list(range(c, n0)) + [q for q in range(n0, nparam) if (q + c) % 4 != 0] for c in range(n0)]
    cols += [list(range(c, nparam)) for c in range(n0, nparam)]
    ss = T.columns_to_structure(cols)
    st = B.Settings(findSparseEliminationRanges=False)
    return B.create_solver(st, np.asarray(sizes, dtype=np.int64), ss, [])


def _check_all_solves(sol, data, dtype, nrhs, want_sweeps=True):
    dev = to_dev(data)
    sol.factor(dev)
    L, A = dense_lower_chol(sol, data)
    Ld = lower_of(sol, dev.cpu().numpy())
    n = sol.order()
    rng = np.random.default_rng(n + nrhs)
    rhs = rng.standard_normal((nrhs, n))
    before = sol.runCounters()["sweep_launches"]
    for name, op in (("solve", A), ("solveL", Ld), ("solveLt", Ld.T)):
        v = to_dev(rhs.astype(dtype).reshape(-1))
        getattr(sol, name)(dev, v, n, nrhs)
        got = v.cpu().numpy().astype(np.float64).reshape(nrhs, n).T
        want = np.linalg.solve(op, rhs.T)
        err = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert err < TOL[dtype], (name, n, nrhs, err)
    took = sol.runCounters()["sweep_launches"] - before
    if want_sweeps:
        assert took >= 4, ("the persistent sweep was not the path taken", took)
    else:
        assert took == 0, took


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("tail", [0, 77, 420])
def test_sweep_run_widths(monkeypatch, dtype, tail):
    """run widths around the panel and block boundaries of the sweep, ragged last panels and blocks"""
    monkeypatch.setenv("BSP_SWEEP_MIN_WIDTH", "128")
    for W in (130, 191, 192, 193, 256, 383, 384, 385, 449, 576, 577, 641, 700, 960, 1101):
        sol = _wide_solver(W, tail, span=8 if W % 2 == 0 else 5)
        data = spd_data(sol, 5 + W, dtype=dtype)
        _check_all_solves(sol, data, dtype, 1 if W % 3 else 2)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sweep_default_threshold_and_rhs_counts(dtype):
    """the default configuration: a 1500-wide root lump + a 2100-wide lump with 500 rows below it,
    1 .. 5 right-hand sides, each its own set of workgroups of the launch (40 for the first structure,
    66 for the second -- three of those still fit the 256 CUs)"""
    for W, tail, counts in ((1500, 0, (1, 3, 5)), (2100, 500, (1, 2, 3))):
        sol = _wide_solver(W, tail)
        data = spd_data(sol, 77 + W, dtype=dtype)
        for nrhs in counts:
            _check_all_solves(sol, data, dtype, nrhs)


def test_sweep_falls_back_when_the_launch_would_not_be_resident():
    """more instances (right-hand sides x batch) than the GPU holds at once: the multi-launch block
    path runs instead, same results"""
    sol = _wide_solver(1100, 140)
    data = spd_data(sol, 3)
    _check_all_solves(sol, data, np.float64, 150, want_sweeps=False)  # ten groups of 16 x 41 workgroups


def test_sweep_switched_off(monkeypatch):
    monkeypatch.setenv("BSP_SOLVE_SWEEP", "0")
    sol = _wide_solver(1100, 140)
    _check_all_solves(sol, spd_data(sol, 3), np.float64, 2, want_sweeps=False)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sweep_batched(monkeypatch, dtype):
    """solve<std::vector<T*>>: one sweep instance per batch entry in the same launch"""
    monkeypatch.setenv("BSP_SWEEP_MIN_WIDTH", "128")
    sol = _wide_solver(705, 91)
    n, nrhs, batch = sol.order(), 2, 3
    mats, rhs, dense = [], [], []
    for q in range(batch):
        data = spd_data(sol, 20 + q, dtype=dtype)
        d = to_dev(data)
        sol.factor(d)
        mats.append(d)
        dense.append(dense_lower_chol(sol, data)[1])
        rhs.append(np.random.default_rng(q).standard_normal((nrhs, n)))
    before = sol.runCounters()["sweep_launches"]
    vecs = [to_dev(r.astype(dtype).reshape(-1)) for r in rhs]
    sol.solve(mats, vecs, n, nrhs)
    for q in range(batch):
        got = vecs[q].cpu().numpy().astype(np.float64).reshape(nrhs, n).T
        want = np.linalg.solve(dense[q], rhs[q].T)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < TOL[dtype], q
    assert sol.runCounters()["sweep_launches"] - before >= 2


def test_sweep_partial_solves(monkeypatch):
    """solveLFrom / solveLtFrom on the trailing block: the plan of a lump range has its own runs"""
    monkeypatch.setenv("BSP_SWEEP_MIN_WIDTH", "128")
    sol = _wide_solver(400, 350)
    data = spd_data(sol, 9)
    dev = to_dev(data)
    sol.factor(dev)
    Ld = lower_of(sol, dev.cpu().numpy())
    sk = sol.skel()
    n = sol.order()
    lump = sol.numLumps() - 1
    span = int(sk["lumpToSpan"][lump])
    bar = int(sk["spanStart"][span])
    rhs = np.random.default_rng(4).standard_normal(n)
    before = sol.runCounters()["sweep_launches"]
    for name, op in (("solveLFrom", Ld[bar:, bar:]), ("solveLtFrom", Ld[bar:, bar:].T)):
        v = to_dev(rhs.copy())
        getattr(sol, name)(dev, span, v, n, 1)
        got = v.cpu().numpy()
        want = rhs.copy()
        want[bar:] = np.linalg.solve(op, rhs[bar:])
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-10, name
    assert sol.runCounters()["sweep_launches"] - before >= 2


def test_sweep_watchdog_ends_a_stuck_launch_and_retires_the_sweeps():
    """fault injection (bsp_test_set_fault kind 2): the spine of block 1 never publishes its x.  The
    launch must END BY ITSELF (bounded spins, 50 ms here), the next solve must report the failed
    call, and from then on the Solver takes the multi-launch path and is correct again."""
    sol = _wide_solver(1500, 0)
    data = spd_data(sol, 31)
    dev = to_dev(data)
    sol.factor(dev)
    _, A = dense_lower_chol(sol, data)
    n = sol.order()
    rhs = np.random.default_rng(1).standard_normal(n)
    sol._testSetFault(2)
    v = to_dev(rhs.copy())
    t0 = time.perf_counter()
    sol.solve(dev, v, n, 1)
    torch.cuda.synchronize()
    took = time.perf_counter() - t0
    assert took < 5.0, ("a stuck sweep must end by its watchdog", took)
    c = sol.runCounters()
    assert c["sweep_error_pending"] == 1 and c["sweeps_retired"] == 0, c
    sol._testSetFault(0)
    with pytest.raises(RuntimeError, match="timed out"):
        sol.solve(dev, to_dev(rhs.copy()), n, 1)
    c = sol.runCounters()
    assert c["sweeps_retired"] == 1 and c["sweep_timeouts"] == 1 and c["sweep_error_pending"] == 0, c
    launched = c["sweep_launches"]
    v = to_dev(rhs.copy())
    sol.solve(dev, v, n, 1)
    want = np.linalg.solve(A, rhs)
    assert np.linalg.norm(v.cpu().numpy() - want) / np.linalg.norm(want) < 1e-10
    assert sol.runCounters()["sweep_launches"] == launched


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("tail", [0, 420])
def test_matrix_core_sweep_several_right_hand_sides(monkeypatch, dtype, tail):
    """csrc/hip_sweep_mfma.h: from eight right-hand sides on (here: three), ONE set of workgroups carries up to 16 of
    them (products on v_mfma, x exchanged as [row][16]); 3 .. 16 right-hand sides in one group, 17 and 35
    in two and three, run widths around the panel / block boundaries, rows below the run"""
    monkeypatch.setenv("BSP_SWEEP_MIN_WIDTH", "128")
    monkeypatch.setenv("BSP_SWEEP_MFMA_MIN", "3")  # (product default: 8)
    monkeypatch.setenv("BSP_SWEEP_MFMA_MIN_WIDTH", "0")  # (product default: runs of at least 7000 columns)
    for W, nrhs in ((130, 3), (192, 5), (193, 16), (385, 4), (577, 10), (641, 17), (960, 7), (1101, 35)):
        sol = _wide_solver(W, tail, span=8 if W % 2 == 0 else 5)
        data = spd_data(sol, 5 + W, dtype=dtype)
        before = sol.runCounters()["sweep_mfma_launches"]
        _check_all_solves(sol, data, dtype, nrhs)
        assert sol.runCounters()["sweep_mfma_launches"] - before >= 4, "the matrix-core sweep was not the path taken"


def test_matrix_core_sweep_batched_and_default_threshold(monkeypatch):
    """ten right-hand sides, a batch of two; the default width threshold (7000 columns) keeps a 1500-wide run
    on the block path, with it lowered the matrix-core sweep takes it"""
    sol0 = _wide_solver(1500, 140)
    d0 = to_dev(spd_data(sol0, 20))
    sol0.factor(d0)
    v0 = to_dev(np.random.default_rng(0).standard_normal((10, sol0.order())).reshape(-1).copy())
    sol0.solve(d0, v0, sol0.order(), 10)
    assert sol0.runCounters()["sweep_mfma_launches"] == 0
    monkeypatch.setenv("BSP_SWEEP_MFMA_MIN_WIDTH", "1024")
    sol = _wide_solver(1500, 140)
    n, nrhs, batch = sol.order(), 10, 2
    mats, rhs, dense = [], [], []
    for q in range(batch):
        data = spd_data(sol, 20 + q)
        d = to_dev(data)
        sol.factor(d)
        mats.append(d)
        dense.append(dense_lower_chol(sol, data)[1])
        rhs.append(np.random.default_rng(q).standard_normal((nrhs, n)))
    before = sol.runCounters()["sweep_mfma_launches"]
    vecs = [to_dev(r.reshape(-1).copy()) for r in rhs]
    sol.solve(mats, vecs, n, nrhs)
    for q in range(batch):
        got = vecs[q].cpu().numpy().reshape(nrhs, n).T
        want = np.linalg.solve(dense[q], rhs[q].T)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-9, q
    assert sol.runCounters()["sweep_mfma_launches"] - before >= 2
