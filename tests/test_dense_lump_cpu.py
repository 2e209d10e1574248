"""Host-side checks of the dense-lump schedule (DenseLumpPlan, csrc/hip_plan.h): the plan of a wide
lump -- chain steps over a two-block window, hand-overs, block trsm and deadline-ordered bulk tiles
on a side stream -- is replayed symbolically by the library itself (bsp_test_verify_dense_lumps):
every 64 x 64 tile is solved once, receives every source panel exactly once, a diagonal tile is
complete before its potrf, and every pair of operations on different streams that meets on a tile
is ordered by an event.  No GPU involved.  Replaces the host-serial per-lump loop of
Solver.cpp:198-218 for lumps of several outer blocks."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T


def _dense(n, below=()):
    sizes = [1] * n + list(below)
    cols = [set(range(i, len(sizes))) for i in range(len(sizes))]
    return np.array(sizes, dtype=np.int64), T.columns_to_structure(cols)


@pytest.mark.parametrize("n", [257, 300, 320, 511, 512, 513, 577, 640, 767, 768, 769, 1000, 1024, 1025,
                               1100, 1279, 1700, 2047, 2048, 2100, 3000])
def test_single_dense_lump(n):
    sizes, ss = _dense(n)
    sol = B.create_solver(B.Settings(), sizes, ss)
    assert sol.numLumps() == 1
    assert sol._testVerifyDenseLumps() == 1


@pytest.mark.parametrize("n", [257, 321, 515, 700, 1030, 1600])
@pytest.mark.parametrize("below", [(40, 30), (3,), (200, 100, 70)])
def test_dense_lump_with_rows_below(n, below):
    """rows below the lump (boards): solved by every block's trsmBlock, updated by bulk tiles tiled from
    the first row below, and sources of the board-target updates"""
    sizes, ss = _dense(n, below)
    sol = B.create_solver(B.Settings(), sizes, ss)
    assert sol._testVerifyDenseLumps() >= 1


@pytest.mark.parametrize("ahead", ["0", "0.3", "1", "100"])
def test_budget_does_not_change_what_is_applied(monkeypatch, ahead):
    """BSP_BULK_AHEAD moves bulk tiles between the due and the optional launches only"""
    monkeypatch.setenv("BSP_BULK_AHEAD", ahead)
    sizes, ss = _dense(1900, (50,))
    sol = B.create_solver(B.Settings(), sizes, ss)
    assert sol._testVerifyDenseLumps() == 1
    n = 1900
    sizes, ss = _dense(n)
    sol = B.create_solver(B.Settings(), sizes, ss)
    dense = 2.0 * sum(64 * (j // 64) * (n - j) for j in range(n))
    assert abs(sol.planStats()["upd_flops"] - dense) <= 1e-9 * dense


def test_bundle_adjustment_shape_and_switch(monkeypatch):
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=130, num_pts=9000, band=30, seed=5)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 9000])
    assert sol._testVerifyDenseLumps() == 1      # the camera lump (1170 columns)
    monkeypatch.setenv("BSP_DENSE_LUMP", "0")
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 9000])
    assert sol._testVerifyDenseLumps() == 0      # level-by-level schedule of rounds 1-3


def test_random_structures():
    """the reference's flat generator at sizes whose top separators are wide lumps: every lump that
    qualifies (alone in its levels, more than one outer block) must verify"""
    seen = 0
    for seed in range(4):
        n = 900 + 100 * seed
        ss = T.gen_flat(n, 0.02, 50 + seed)
        sol = B.create_solver(B.Settings(), np.full(n, 3, dtype=np.int64), ss)
        seen += sol._testVerifyDenseLumps()
    assert seen >= 1
