"""ORACLE (test infrastructure only): block-pattern helpers in plain Python sets/lists.

Restates the pieces of baspacho/baspacho/SparseStructure.cpp and baspacho/testing/TestingUtils.cpp
that the reference's tests use as ground truth:
  transpose                      SparseStructure.cpp:34-66
  symmetricPermutation           SparseStructure.cpp:109-159
  naiveAddEliminationEntries     TestingUtils.cpp:196-212   (the reference's own "gt" for fill)
  makeIndependentElimSet         TestingUtils.cpp:214-230
  columnsToCscStruct / csrStructToColumns / joinColums   TestingUtils.cpp:150-194
"""


def columns_to_csc(columns):
    ptrs, inds = [0], []
    for col in columns:
        inds.extend(sorted(col))
        ptrs.append(len(inds))
    return ptrs, inds


def csr_to_columns(ptrs, inds):
    n = len(ptrs) - 1
    cols = [set() for _ in range(n)]
    for i in range(n):
        for k in range(ptrs[i], ptrs[i + 1]):
            cols[inds[k]].add(i)
    return cols


def transpose(ptrs, inds):
    n = len(ptrs) - 1
    rows = [[] for _ in range(n)]
    for i in range(n):
        for k in range(ptrs[i], ptrs[i + 1]):
            rows[inds[k]].append(i)
    return columns_to_csc_lists(rows)


def columns_to_csc_lists(lists):
    ptrs, inds = [0], []
    for l in lists:
        inds.extend(l)
        ptrs.append(len(inds))
    return ptrs, inds


def symmetric_permutation(ptrs, inds, map_perm, lower_half=True):
    n = len(ptrs) - 1
    buckets = [[] for _ in range(n)]
    for i in range(n):
        ni = map_perm[i]
        for k in range(ptrs[i], ptrs[i + 1]):
            nj = map_perm[inds[k]]
            col = min(ni, nj) if lower_half else max(ni, nj)
            row = max(ni, nj) if lower_half else min(ni, nj)
            buckets[col].append(row)
    return columns_to_csc_lists([sorted(b) for b in buckets])


def naive_add_elimination_entries(columns, start, end):
    """in place: eliminating column i connects every pair of its below-diagonal rows"""
    for i in range(start, end):
        rows = sorted(columns[i])
        assert rows[0] == i
        for a in range(1, len(rows)):
            for b in range(a + 1, len(rows)):
                columns[rows[a]].add(rows[b])


def make_independent_elim_set(columns, start, end):
    out = []
    for i, col in enumerate(columns):
        if i < start or i >= end:
            out.append(set(col))
        else:
            out.append({i} | {c for c in col if c >= end})
    return out


def join_columns(columns, lump_start):
    out = []
    for a in range(len(lump_start) - 1):
        s = set()
        for i in range(lump_start[a], lump_start[a + 1]):
            s |= columns[i]
        out.append(s)
    return out
