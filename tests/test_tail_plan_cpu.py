"""Plan-level rules of the persistent tail (csrc/hip_plan.cpp), checked without a GPU: which lumps hand
columns to the tail launch.  A tail level factors nothing but the tail, so its panels must be alone in
their levels (forests of wide roots: found on the reference's MERI family in round 6)."""
import numpy as np

import baspacho_amd as B
from baspacho_amd import testing as T


def _cliques(widths, span=8):
    cols, base = [], 0
    for w in widths:
        k = w // span
        cols += [list(range(base + c, base + k)) for c in range(k)]
        base += k
    return np.full(base, span, dtype=np.int64), T.columns_to_structure(cols)


def _tail_panels(widths):
    sizes, ss = _cliques(widths)
    sol = B.create_solver(B.Settings(findSparseEliminationRanges=False, hipOptions={"lazy_plan": 1}), sizes, ss, [])
    return sol.planStats()["num_tail_panels"]


def test_tail_only_for_panels_alone_in_their_levels():
    assert _tail_panels([1600]) == 21            # last 6 of 7 outer blocks (5 x 4 panels + 1)
    assert _tail_panels([1600, 1600]) == 0       # two roots on the same levels
    assert _tail_panels([1600, 3200]) == 22      # the wider root's last blocks are alone (levels 28 ..)
    assert _tail_panels([900, 900]) == 0


def test_narrow_root_lump_rule():
    assert _tail_panels([990]) == 12             # GRID 82x82's root: all but the first outer block
    assert _tail_panels([576]) == 0              # fewer than six panels of tail
    assert _tail_panels([640]) == 6


def test_whole_narrow_root_behind_children():
    """two blocks of 500, each coupled to 150 columns of a 900-column separator: the root follows other levels and is the tail as a
    whole (15 panels); alone at level 0 it keeps its first outer block (11 panels)"""
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
    sol = B.create_solver(B.Settings(hipOptions={"lazy_plan": 1}), np.ones(n, dtype=np.int64), T.columns_to_structure(cols))
    assert sol.planStats()["num_tail_panels"] == 15
    assert _tail_panels([904]) == 11
