"""Writes tests/golden/reference_known_answers.json.

The reference has no Python and cannot be compiled here, so the committed golden
vectors are the literal inputs / expected outputs that the reference's own unit tests hold
(data, not code), each with the file:line it was transcribed from, plus dense-Cholesky answers
computed by numpy.linalg.cholesky (the reference's tests use Eigen::LLT for the same purpose).
Run:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

G = {}

# ---- tests/CoalescedBlockMatrixTest.cpp:48-112 (BasicAssertions)
G["skeleton_9span"] = {
    "source": "baspacho/tests/CoalescedBlockMatrixTest.cpp:48-112",
    "spanStart": [0, 1, 2, 4, 5, 7, 9, 12, 14, 16],
    "lumpToSpan": [0, 1, 3, 4, 6, 7, 9],
    "columnParams": [[0, 1, 2, 5, 8], [1, 2, 3, 6, 7], [3, 4, 5, 8], [4, 5, 7], [6, 8], [7, 8]],
    "expected": {
        "spanToLump": [0, 1, 1, 2, 3, 3, 4, 5, 5, 6],
        "lumpStart": [0, 1, 4, 5, 9, 12, 16],
        "chainColPtr": [0, 5, 10, 14, 17, 19, 21],
        "chainRowSpan": [0, 1, 2, 5, 8, 1, 2, 3, 6, 7, 3, 4, 5, 8, 4, 5, 7, 6, 8, 7, 8],
        "chainData": [0, 1, 2, 4, 6, 8, 11, 17, 20, 29, 35, 36, 38, 40, 42, 50, 58, 66, 75, 81, 89,
                      97],
        "chainRowsTillEnd": [1, 2, 4, 6, 8, 1, 3, 4, 7, 9, 1, 3, 5, 7, 2, 4, 6, 3, 5, 2, 4],
        "boardColPtr": [0, 5, 10, 14, 17, 20, 22],
        "boardRowLump": [0, 1, 3, 5, -1, 1, 2, 4, 5, -1, 2, 3, 5, -1, 3, 5, -1, 4, 5, -1, 5, -1],
        "boardChainColOrd": [0, 1, 3, 4, 5, 0, 2, 3, 4, 5, 0, 1, 3, 4, 0, 2, 3, 0, 1, 2, 0, 2],
        "boardRowPtr": [0, 1, 3, 5, 8, 10, 16],
        "boardColLump": [0, 0, 1, 1, 2, 0, 2, 3, 1, 4, 0, 1, 2, 3, 4, 5],
        "boardColOrd": [0, 1, 0, 1, 0, 2, 1, 0, 2, 0, 3, 3, 2, 1, 1, 0],
    },
}

# ---- tests/CoalescedBlockMatrixTest.cpp:114-180 (Densify, Densify2): data = iota(13..)
G["densify_9span"] = {
    "source": "baspacho/tests/CoalescedBlockMatrixTest.cpp:114-180",
    "data_iota_start": 13,
    "expected_lower_16x16": [
        [13, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [14, 21, 22, 23, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [15, 24, 25, 26, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [16, 27, 28, 29, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [0, 30, 31, 32, 48, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 49, 55, 56, 57, 58, 0, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 50, 59, 60, 61, 62, 0, 0, 0, 0, 0, 0, 0],
        [17, 0, 0, 0, 51, 63, 64, 65, 66, 0, 0, 0, 0, 0, 0, 0],
        [18, 0, 0, 0, 52, 67, 68, 69, 70, 0, 0, 0, 0, 0, 0, 0],
        [0, 33, 34, 35, 0, 0, 0, 0, 0, 79, 80, 81, 0, 0, 0, 0],
        [0, 36, 37, 38, 0, 0, 0, 0, 0, 82, 83, 84, 0, 0, 0, 0],
        [0, 39, 40, 41, 0, 0, 0, 0, 0, 85, 86, 87, 0, 0, 0, 0],
        [0, 42, 43, 44, 0, 71, 72, 73, 74, 0, 0, 0, 94, 95, 96, 97],
        [0, 45, 46, 47, 0, 75, 76, 77, 78, 0, 0, 0, 98, 99, 100, 101],
        [19, 0, 0, 0, 53, 0, 0, 0, 0, 88, 89, 90, 102, 103, 104, 105],
        [20, 0, 0, 0, 54, 0, 0, 0, 0, 91, 92, 93, 106, 107, 108, 109]],
    "expected_full_from_span1_15x15": [
        [21, 24, 27, 30, 0, 0, 0, 0, 33, 36, 39, 42, 45, 0, 0],
        [24, 25, 28, 31, 0, 0, 0, 0, 34, 37, 40, 43, 46, 0, 0],
        [27, 28, 29, 32, 0, 0, 0, 0, 35, 38, 41, 44, 47, 0, 0],
        [30, 31, 32, 48, 49, 50, 51, 52, 0, 0, 0, 0, 0, 53, 54],
        [0, 0, 0, 49, 55, 59, 63, 67, 0, 0, 0, 71, 75, 0, 0],
        [0, 0, 0, 50, 59, 60, 64, 68, 0, 0, 0, 72, 76, 0, 0],
        [0, 0, 0, 51, 63, 64, 65, 69, 0, 0, 0, 73, 77, 0, 0],
        [0, 0, 0, 52, 67, 68, 69, 70, 0, 0, 0, 74, 78, 0, 0],
        [33, 34, 35, 0, 0, 0, 0, 0, 79, 82, 85, 0, 0, 88, 91],
        [36, 37, 38, 0, 0, 0, 0, 0, 82, 83, 86, 0, 0, 89, 92],
        [39, 40, 41, 0, 0, 0, 0, 0, 85, 86, 87, 0, 0, 90, 93],
        [42, 43, 44, 0, 71, 72, 73, 74, 0, 0, 0, 94, 98, 102, 106],
        [45, 46, 47, 0, 75, 76, 77, 78, 0, 0, 0, 98, 99, 103, 107],
        [0, 0, 0, 53, 0, 0, 0, 0, 88, 89, 90, 102, 103, 104, 108],
        [0, 0, 0, 54, 0, 0, 0, 0, 91, 92, 93, 106, 107, 108, 109]],
}

# ---- tests/SparseStructureTest.cpp:20-63 (Transpose, SymPermutation)
G["transpose"] = {
    "source": "baspacho/tests/SparseStructureTest.cpp:20-34",
    "ptrs": [0, 2, 4, 7, 9, 11], "inds": [0, 3, 2, 4, 0, 1, 4, 1, 2, 2, 4],
    "expected_ptrs": [0, 2, 4, 7, 8, 11], "expected_inds": [0, 2, 2, 3, 1, 3, 4, 0, 1, 2, 4],
}
G["sym_permutation"] = {
    "source": "baspacho/tests/SparseStructureTest.cpp:36-63",
    "ptrs": [0, 1, 2, 4, 6, 8, 12], "inds": [0, 1, 0, 1, 1, 3, 2, 4, 0, 1, 4, 5],
    "mapPerm": [4, 5, 2, 1, 0, 3],
    "expected_upper_ptrs": [0, 1, 2, 3, 5, 8, 12],
    "expected_upper_inds": [0, 1, 0, 0, 3, 2, 3, 4, 1, 2, 3, 5],
    "expected_lower_ptrs": [0, 3, 5, 7, 10, 11, 12],
    "expected_lower_inds": [0, 2, 3, 1, 5, 4, 5, 3, 4, 5, 4, 5],
}

# ---- tests/SparseStructureTest.cpp:117-152 (FillReducingPermutation): 24-node graph,
#      nnz(L) after ordering must be <= 130 ("should be 120")
G["amd_24"] = {
    "source": "baspacho/tests/SparseStructureTest.cpp:117-152",
    "ptrs": [0, 9, 15, 21, 27, 33, 39, 48, 57, 61, 70, 76, 82, 88, 94, 100, 106, 110, 119, 128,
             137, 143, 152, 156, 160],
    "inds": [0, 5, 6, 12, 13, 17, 18, 19, 21, 1, 8, 9, 13, 14, 17, 2, 6, 11, 20, 21, 22, 3, 7, 10,
             15, 18, 19, 4, 7, 9, 14, 15, 16, 0, 5, 6, 12, 13, 17, 0, 2, 5, 6, 11, 12, 19, 21, 23,
             3, 4, 7, 9, 14, 15, 16, 17, 18, 1, 8, 9, 14, 1, 4, 7, 8, 9, 13, 14, 17, 18, 3, 10, 18,
             19, 20, 21, 2, 6, 11, 12, 21, 23, 0, 5, 6, 11, 12, 23, 0, 1, 5, 9, 13, 17, 1, 4, 7, 8,
             9, 14, 3, 4, 7, 15, 16, 18, 4, 7, 15, 16, 0, 1, 5, 7, 9, 13, 17, 18, 19, 0, 3, 7, 9,
             10, 15, 17, 18, 19, 0, 3, 6, 10, 17, 18, 19, 20, 21, 2, 10, 19, 20, 21, 22, 0, 2, 6,
             10, 11, 19, 20, 21, 22, 2, 20, 21, 22, 6, 11, 12, 23],
    "max_fill_nnz": 130,
}

# ---- tests/FactorTest.cpp:43-65 (testCoalescedFactor): tiny coalesced factor, tol 1e-10 (f64)
tiny_cols = [[0, 3, 5], [1], [2, 4], [3], [4], [5]]
G["tiny_factor"] = {
    "source": "baspacho/tests/FactorTest.cpp:43-65",
    "colBlocks": tiny_cols,
    "spanStart": [0, 2, 5, 7, 10, 12, 15],
    "lumpToSpan": [0, 2, 4, 6],
    "data_iota_start": 13, "damp_alpha": 5, "damp_beta": 50,
    "tol_f64": 1e-10, "tol_f32": 1e-5,
}


def tiny_factor_answer():
    """dense Cholesky answer for the tiny case, computed with numpy (Eigen::LLT stand-in)"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import skel as OS, structure as ST
    cols = [set(c) for c in tiny_cols]
    ptrs, inds = ST.columns_to_csc(cols)
    rptr, rind = ST.transpose(ptrs, inds)           # csr lower
    fcols = ST.csr_to_columns(rptr, rind)
    ST.naive_add_elimination_entries(fcols, 0, len(fcols))   # == addFullEliminationFill
    grouped = ST.join_columns(fcols, G["tiny_factor"]["lumpToSpan"])
    gp, gi = ST.columns_to_csc(grouped)
    sk = OS.build_skeleton(G["tiny_factor"]["spanStart"], G["tiny_factor"]["lumpToSpan"], gp, gi)
    data = np.arange(13, 13 + OS.data_size(sk), dtype=np.float64)
    OS.damp(sk, data, 5.0, 50.0)
    A = OS.densify(sk, data, fill_upper_half=True)
    L = np.linalg.cholesky(A)
    return {"groupedPtrs": gp, "groupedInds": gi, "dataSize": OS.data_size(sk),
            "L_lower": L.tolist()}


G["tiny_factor"]["answer"] = tiny_factor_answer()

with open(os.path.join(HERE, "reference_known_answers.json"), "w") as f:
    json.dump(G, f, indent=1)
print("wrote", os.path.join(HERE, "reference_known_answers.json"))
