"""CPU: the C-ABI library loads, exports every symbol declared in include/baspacho_amd.h, and its
host-only entry points (symbolic analysis, accessors, plan serialisation) behave."""
import os
import re

import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import _lib
from baspacho_amd import testing as T
from helpers import solver_random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    import glob
    hdrs = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert len(hdrs) >= 2   # the drop-in boundary + the testing hooks
    hdr = "".join(open(h).read() for h in hdrs)
    names = set(re.findall(r"\b(bsp_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) > 30
    lib = _lib.load()
    for n in sorted(names):
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.bsp_version()


def test_cpu_backends_are_refused():
    sol, ps, ss = solver_random(57)
    for backend in (B.BackendRef, B.BackendFast):
        with pytest.raises(RuntimeError, match="oracle"):
            B.create_solver(B.Settings(backend=backend), ps, ss)
    B.create_solver(B.Settings(backend=B.BackendCuda), ps, ss)  # alias of the HIP engine


def test_accessor_matches_skeleton():
    """AccessorTest: blockOffset/diagBlockOffset through the permutation, incl. flipped blocks"""
    sol, ps, ss = solver_random(63)
    sk = sol.skel()
    p2s = sol.paramToSpan()
    data = np.arange(sol.dataSize(), dtype=np.float64)
    dense = sol.densify(data, fill_upper_half=False)
    n = len(ps)
    checked = 0
    for r in range(n):
        for k in range(int(ss.ptrs[r]), int(ss.ptrs[r + 1])):
            c = int(ss.inds[k])
            off, stride, flipped = sol.blockOffset(r, c)
            sr, sc = int(p2s[r]), int(p2s[c])
            assert flipped == (sr < sc)
            lo, hi = min(sr, sc), max(sr, sc)
            assert data[off] == dense[int(sk["spanStart"][hi]), int(sk["spanStart"][lo])]
            assert stride == sk["lumpStart"][sk["spanToLump"][lo] + 1] - \
                sk["lumpStart"][sk["spanToLump"][lo]]
            checked += 1
        off, stride = sol.diagBlockOffset(r)
        s = int(p2s[r])
        assert data[off] == dense[int(sk["spanStart"][s]), int(sk["spanStart"][s])]
        assert sol.order() == int(ps.sum())
    assert checked > n
    with pytest.raises(RuntimeError):
        sol.blockOffset(n + 5, 0)


def test_permutation_and_fill_properties():
    """EliminationTreeTest / CreateSolverTest properties: the factor skeleton contains the
    permuted original pattern; lumps partition the spans; sizes are permuted consistently"""
    for seed in range(8):
        sol, ps, ss = solver_random(70 + seed, model="openblas" if seed % 2 else "hip")
        p2s = sol.paramToSpan()
        assert sorted(p2s.tolist()) == list(range(len(ps)))
        sk = sol.skel()
        sizes = np.diff(sk["spanStart"])
        assert np.array_equal(sizes[p2s], ps)
        for r in range(len(ps)):
            for k in range(int(ss.ptrs[r]), int(ss.ptrs[r + 1])):
                sol.blockOffset(r, int(ss.inds[k]))  # raises if the block is missing
        # elimination ranges are made of single-span lumps
        for l in range(int(sol.sparseEliminationRanges()[-1]) if len(sol.sparseEliminationRanges()) else 0):
            assert sk["lumpToSpan"][l + 1] - sk["lumpToSpan"][l] == 1


def test_plan_serialisation_roundtrip():
    sol, _, _ = solver_random(57, fill=0.03, elim=(0, 60))
    buf = sol.serialize_plan()
    clone = B.Solver.from_plan(buf)
    a, b = sol.skel(), clone.skel()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(sol.paramToSpan(), clone.paramToSpan())
    assert np.array_equal(sol.sparseEliminationRanges(), clone.sparseEliminationRanges())
    assert clone.canFactorUpToSpan() == sol.canFactorUpToSpan()
    assert clone.planStats() == sol.planStats()
    with pytest.raises(RuntimeError):
        B.Solver.from_plan(buf[:-3])


def test_flops_formula_and_plan_stats():
    sol, _, _ = solver_random(58)
    sk = sol.skel()
    flops = 0.0
    for l in range(sol.numLumps()):
        n = float(sk["lumpStart"][l + 1] - sk["lumpStart"][l])
        c0, c1 = int(sk["chainColPtr"][l]), int(sk["chainColPtr"][l + 1])
        r = float(sk["chainRowsTillEnd"][c1 - 1]) - n
        flops += n ** 3 / 3 + r * n * n + r * r * n
    assert abs(sol.factorFlops() - flops) <= 1e-9 * flops
    st = sol.planStats()
    assert abs(st["flops"] - flops) <= 1e-9 * flops
    assert st["num_panels"] >= sol.numLumps() - (int(sol.sparseEliminationRanges()[-1]) if len(
        sol.sparseEliminationRanges()) else 0)


@pytest.mark.parametrize("n", [700, 1700, 2049])
def test_lookahead_schedule_covers_every_update_once(monkeypatch, n):
    """host-side check of the device plan of one dense lump: however the lookahead units are
    scheduled (as late as allowed, default budget, as early as possible), the update work of the
    plan is the same and equals the dense count -- every (source columns -> target element) update
    is planned exactly once.  Dense count: element (i, j), j <= i, of the n x n lower triangle
    receives a rank-1 update from every column k that lies in an earlier 64-column panel than j
    (the columns of j's own panel act inside its potrf / trsm): 2 * sum_j 64 (j // 64) (n - j)."""
    ss = T.columns_to_structure([set(range(i, n)) for i in range(n)])
    dense = 2.0 * sum(64 * (j // 64) * (n - j) for j in range(n))
    seen = []
    for ahead in ("0", "0.6", "100"):
        monkeypatch.setenv("BSP_BULK_AHEAD", ahead)
        sol = B.create_solver(B.Settings(), np.ones(n, dtype=np.int64), ss)
        assert sol.numLumps() == 1
        st = sol.planStats()
        # (the updates inside a persistent tail launch are counted apart)
        planned = st["upd_flops"] + st["tail_upd_flops"]
        seen.append((planned, st["trsm_flops"], st["potrf_flops"]))
        assert abs(planned - dense) <= 1e-9 * dense, (ahead, planned, dense)
        # (wide rule: the last 6 of >= 6 outer blocks; narrow rule: all but the first block when that
        #  leaves at least six panels -- 700 columns: 7 panels)
        assert (st["num_tail_panels"] > 0) == (n - 256 > 5 * 64), st["num_tail_panels"]
    assert seen[0] == seen[1] == seen[2]
    assert seen[0][1] >= st["trsm_flops_merged"] > 0 and seen[0][2] >= st["potrf_flops_fused"] > 0


def test_block_tridiagonal_config_c1():
    """BASELINE config 0: 3334 x (3x3) block tridiagonal, symbolic analysis + oracle factor"""
    from oracle import cref
    ss = T.block_tridiagonal(3334)
    sol = B.create_solver(B.Settings(), np.full(3334, 3), ss)
    assert sol.order() == 10002
    data = T.random_data(sol.dataSize(), -1, 1, 37)
    sol.damp(data, 0.0, sol.order() * 1.2)
    A = sol.densify(data, fill_upper_half=True)
    cref.factor(sol.skel(), data, sol.sparseEliminationRanges())
    L = np.tril(sol.densify(data))
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) < 1e-12


def _chainlike(n, closures, seed):
    """a pose chain (i, i+1) with a few loop closures and short side branches"""
    rng = np.random.default_rng(seed)
    rows = list(range(1, n))
    cols = list(range(0, n - 1))
    for _ in range(closures):
        a, b = sorted(rng.choice(n, 2, replace=False))
        if a != b:
            rows.append(int(b))
            cols.append(int(a))
    return T.structure_from_pairs(n, np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64))


@pytest.mark.parametrize("closures", [0, 7])
def test_chain_contraction_gives_ranges_and_a_valid_factor(closures):
    """createSolver's ordering takes independent sets of pivots with <= 2 neighbours first
    (min_degree.cpp, contractChains): on a chain the elimination tree is then logarithmic, its
    height classes become sparse-elimination ranges, and the factor is still that of P A P^T.
    The public fillReducingPermutation (reference API, SparseStructure.h:96) stays plain min-degree."""
    from oracle import cref
    n = 900
    ss = _chainlike(n, closures, 3)
    sizes = np.full(n, 3, dtype=np.int64)
    sol = B.create_solver(B.Settings(), sizes, ss)
    ranges = sol.sparseEliminationRanges()
    assert len(ranges) >= 4 and ranges[1] >= n // 2 - closures * 3
    assert sorted(sol.paramToSpan().tolist()) == list(range(n))
    # far fewer dependent steps than lumps
    st = sol.planStats()
    assert st["num_levels"] <= 12 + 3 * closures
    data = T.random_data(sol.dataSize(), -1, 1, 37)
    sol.damp(data, 0.0, sol.order() * 1.2)
    A = sol.densify(data, fill_upper_half=True)
    cref.factor(sol.skel(), data, ranges)
    L = np.tril(sol.densify(data))
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) < 1e-12
    p = ss.fillReducingPermutation()
    assert sorted(np.asarray(p).tolist()) == list(range(n))


def _connect_ranges_literal(columns, b1, e1, b2, e2, fill, max_offset, seed):
    """the reference's connectRanges loop, statement for statement (TestingMatGen.cpp:23-50), on a
    list of sets; the keep/drop decision is the generator's stateless hash of (i, j)"""
    size = len(columns)
    if b1 > b2:
        return _connect_ranges_literal(columns, b2, e2, b1, e1, fill, max_offset, seed)
    if e1 > e2:
        _connect_ranges_literal(columns, b2, e2, e2, e1, fill, max_offset, seed)
    for i in range(b1, e1):
        d_begin = min(max_offset, max(b2 - i, 1))
        d_end = min(max_offset, e2 - i)
        for j in range(i + d_begin, i + d_end):
            u = T.hash_unit(seed, np.uint64(i) * np.uint64(size) + np.array([j], dtype=np.uint64))[0]
            if fill >= 1.0 or fill > u:
                columns[i].add(j)


def test_gen_meridians_is_the_reference_construction():
    """genMeridians (TestingMatGen.cpp:87-168) restated with numpy == the literal loops on a small
    instance; bands never exceed `band`; the size formula holds"""
    num, line_len, fill, band, hair_len, nh, sh, seed = 3, 40, 0.5, 7, 15, 2, 1, 5
    ss = T.gen_meridians(num, line_len, fill, band, hair_len, nh, sh, seed)
    size = line_len * num + hair_len * (nh + sh)
    assert ss.order() == size
    cols = [{i} for i in range(size)]
    end_m = line_len * num

    def conn(b1, e1, b2, e2):
        _connect_ranges_literal(cols, b1, e1, b2, e2, fill, band, seed)

    for i in range(num):
        conn(line_len * i, line_len * (i + 1), line_len * i, line_len * (i + 1))
    for h in range(nh + sh):
        b = end_m + hair_len * h
        conn(b, b + hair_len, b, b + hair_len)
    for i in range(num):
        ib = line_len * i
        for j in range(i):
            jb = line_len * j
            conn(ib, ib + band, jb, jb + band)
            conn(ib + line_len - band, ib + line_len, jb + line_len - band, jb + line_len)
    for i in range(num):
        ib = line_len * i
        for h in range(nh):
            hb = end_m + hair_len * h
            conn(ib, ib + band, hb, hb + band)
        for h in range(sh):
            hb = end_m + hair_len * (h + nh)
            conn(ib + line_len - band, ib + line_len, hb, hb + band)
    for h in range(nh):
        hb = end_m + hair_len * h
        for k in range(h):
            kb = end_m + hair_len * k
            conn(kb, kb + band, hb, hb + band)
    for h in range(sh):
        hb = end_m + hair_len * (h + nh)
        for k in range(h):
            conn(hb, hb + band, hb, hb + band)
    want = T.columns_to_structure(cols)
    assert np.array_equal(ss.ptrs, want.ptrs) and np.array_equal(ss.inds, want.inds)
    rows = np.repeat(np.arange(size), np.diff(ss.ptrs))
    assert (rows - ss.inds).max() < band
    # a genuinely connected case: adjacent tracks closer than the band DO get pole connections
    ss2 = T.gen_meridians(2, 8, 1.0, 8, 8, 1, 0, 1)
    rows2 = np.repeat(np.arange(ss2.order()), np.diff(ss2.ptrs))
    assert ((rows2 >= 8) & (ss2.inds < 8)).any()
