"""shared helpers of the parity tests (problem construction mirrors the reference's tests)"""
import numpy as np

import baspacho_amd as B
from baspacho_amd import testing as T
from baspacho_amd.csrc_models import MODEL_OPENBLAS_I7


def columns_to_csc(columns):
    ptrs, inds = [0], []
    for col in columns:
        inds.extend(sorted(col))
        ptrs.append(len(inds))
    return ptrs, inds


def solver_random(seed, size=115, fill=0.037, elim=None, pmin=2, pmax=5, model="openblas",
                  ranges=(), policy=B.AddFillComplete, last_ids=(), psize_seed=47,
                  find_ranges=True):
    """random block pattern as in tests/FactorTest.cpp:75-107 / CreateSolverTest.cpp:75-140"""
    cols = T.random_cols(size, fill, seed)
    if elim is not None:
        cols = T.make_independent_elim_set(cols, elim[0], elim[1])
    ss = T.columns_to_structure(cols)
    ps = T.random_vec(size, pmin, pmax, psize_seed)
    st = B.Settings(findSparseEliminationRanges=find_ranges, addFillPolicy=policy,
                    computationModel=MODEL_OPENBLAS_I7 if model == "openblas" else None)
    return B.create_solver(st, ps, ss, ranges, last_ids), ps, ss


def spd_data(sol, seed, beta_factor=1.5, dtype=np.float64):
    data = T.random_data(sol.dataSize(), -1.0, 1.0, seed).astype(dtype)
    sol.damp(data, dtype(0), dtype(sol.order() * beta_factor))
    return data


def dense_lower_chol(sol, data):
    A = sol.densify(data.astype(np.float64), fill_upper_half=True)
    return np.linalg.cholesky(A), A


def lower_of(sol, data):
    return np.tril(sol.densify(np.asarray(data, dtype=np.float64)))


def to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()
