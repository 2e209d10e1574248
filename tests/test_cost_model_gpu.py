"""GPU: the sample collection behind the cost-model fit (the reference's `bench -Z`
statistics, Bench.cpp:72-124, Solver::enableStats): collectOpStats() / opStats() over a factor driven
through the per-op boundary must hold exactly one sample per potrf / trsm / syrk-gemm / assemble call
the reference's loop makes (Solver.cpp:198-218) -- counts and sizes derived here from the skeleton
alone -- with positive times; and the vendor comparator of tools/ runs (tools only, never the product)."""
import os
import sys

import numpy as np
import pytest
import torch

import baspacho_amd as B
from baspacho_amd import testing as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected_ops(sol):
    sk = sol.skel()
    ranges = sol.sparseEliminationRanges()
    dense_from = int(ranges[-1]) if len(ranges) else 0
    n_lumps = sol.numLumps()
    ls = sk["lumpStart"]
    widths, with_rows, boards = [], 0, 0
    for l in range(dense_from, n_lumps):
        w = int(ls[l + 1] - ls[l])
        widths.append(w)
        c0, c1 = int(sk["chainColPtr"][l]), int(sk["chainColPtr"][l + 1])
        total_rows = int(sk["chainRowsTillEnd"][c1 - 1])
        with_rows += 1 if total_rows > w else 0
        r0, r1 = int(sk["boardRowPtr"][l]), int(sk["boardRowPtr"][l + 1])
        cols = sk["boardColLump"][r0:r1 - 1]          # last board of the row = the diagonal
        boards += int(((cols >= dense_from) & (cols < l)).sum())
    return sorted(widths), with_rows, boards


@pytest.mark.parametrize("kind", ["grid", "bal", "flat"])
def test_op_stats_hold_one_sample_per_reference_call(kind):
    if kind == "grid":
        sizes, ss, ranges = np.full(30 * 30, 3, dtype=np.int64), T.gen_grid(30, 30, 1.0, 2, 37), []
    elif kind == "bal":
        sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=60, num_pts=4000, band=8, seed=3)
        ranges = [0, 4000]
    else:
        sizes, ss, ranges = np.full(900, 3, dtype=np.int64), T.gen_flat(900, 6.0e-3, 37), []
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    data = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
    sol.damp(data, 0.0, sol.order() * 1.2)
    d = torch.from_numpy(data).cuda()
    sol.factorPerOp(d.clone())          # warm-up
    sol.collectOpStats(True)
    sol.factorPerOp(d)
    st = sol.opStats()
    sol.collectOpStats(False)
    widths, n_trsm, n_boards = _expected_ops(sol)
    assert sorted(int(v) for v in st["potrf"][:, 0]) == widths
    assert len(st["trsm"]) == n_trsm
    assert len(st["syge"]) == n_boards and len(st["asmbl"]) == n_boards
    for k, a in st.items():
        if len(a):
            assert (a[:, -1] > 0).all() and (a[:, -1] < 1.0).all(), k       # seconds, HIP-event timed
            assert (a[:, :-1] >= 0).all() and (a[:, :-1] == np.floor(a[:, :-1])).all(), k
    # trsm samples: n = a lump width, k = its rows below; syge: k = source lump width
    assert set(int(v) for v in st["trsm"][:, 0]) <= set(widths)
    if n_boards:
        assert set(int(v) for v in st["syge"][:, 2]) <= set(widths)
    # nothing is collected once switched off (the samples are kept or dropped, never extended)
    sol.factorPerOp(torch.from_numpy(data).cuda())
    assert all(len(v) in (0, len(st[k])) for k, v in sol.opStats().items())


def test_vendor_comparator_runs():
    """tools/vendor_compare.py (rocSOLVER dpotrf, rocBLAS dsyrk through ctypes): the numbers DESIGN.md
    sets beside the hand-written dense phase.  Only checks that it runs and returns sane rates."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import vendor_compare
    r = vendor_compare.run(n_list=(2048,), syrk=((4096, 256),), reps=2)
    assert r["rocsolver_dpotrf"][0]["n"] == 2048 and r["rocsolver_dpotrf"][0]["ms"] > 0
    assert r["rocsolver_dpotrf"][0]["rel_err"] < 1e-10
    assert r["rocblas_dsyrk"][0]["ms"] > 0
