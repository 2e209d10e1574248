"""GPU: several Solvers of one process factoring and solving at the same time, each on a stream of its
own from a thread of its own.  The auxiliary (lookahead / elimination) streams and the event pool
are shared by every Solver of a device: their launches may interleave in any
order, the results may not change.  (The reference's Solver is "one factor() at a time per Solver",
Solver.h threading note; several Solvers side by side is what a caller like Theseus does.)"""
import threading

import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import dense_lower_chol, lower_of, spd_data

pytestmark = pytest.mark.gpu


def _dense_lump_solver(width):
    nparam = width // 8
    sizes = np.full(nparam, 8, dtype=np.int64)
    cols = [list(range(c, nparam)) for c in range(nparam)]
    return B.create_solver(B.Settings(), sizes, T.columns_to_structure(cols), [])


def _random_solver(seed, size, fill):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, 7, size=size).astype(np.int64)
    cols = T.make_independent_elim_set(T.random_cols(size, fill, 100 + seed), 0, size // 3)
    return B.create_solver(B.Settings(), sizes, T.columns_to_structure(cols), [0, size // 3])


def test_three_solvers_on_three_streams_from_three_threads():
    import torch
    dev = torch.device("cuda", 0)
    # wide dense lumps (chain + lookahead units on the shared side streams) and an elimination range
    sols = [_dense_lump_solver(1856), _random_solver(5, 700, 0.2), _dense_lump_solver(2112)]
    datas = [spd_data(s, 11 + i) for i, s in enumerate(sols)]
    refs = [dense_lower_chol(s, d) for s, d in zip(sols, datas)]
    errs = [[], [], []]
    failures = []

    def work(i):
        try:
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(device=dev)
            sol, (L, A) = sols[i], refs[i]
            sol.setStream(stream)
            n = sol.order()
            rhs = np.random.default_rng(i).standard_normal(n)
            X = np.linalg.solve(A, rhs)
            with torch.cuda.stream(stream):
                for it in range(10):
                    d = torch.from_numpy(datas[i]).to(dev, non_blocking=False)
                    v = torch.from_numpy(rhs.copy()).to(dev)
                    stream.wait_stream(torch.cuda.default_stream(dev))
                    sol.factor(d)
                    sol.solve(d, v, n, 1)
                    stream.synchronize()
                    got = lower_of(sol, d.cpu().numpy())
                    errs[i].append((np.linalg.norm(got - L) / np.linalg.norm(L),
                                    np.linalg.norm(v.cpu().numpy() - X) / np.linalg.norm(X)))
        except Exception as e:  # noqa: BLE001
            failures.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s in sols:
        s.setStream(None)
    assert not failures, failures
    for i in range(3):
        assert len(errs[i]) == 10
        for fe, se in errs[i]:
            assert fe < 1e-11 and se < 1e-9, (i, errs[i])
