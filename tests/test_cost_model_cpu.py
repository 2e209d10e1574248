"""CPU test of the cost-model fit pipeline (reference: bench -Z per-op CSVs,
benchmarking/Bench.cpp:72-124, fitted by examples/OptimizeCompModel.cpp:64-275).  The per-op samples
were dumped on an MI355X by tools/op_stats_dump.py (profiles/r06_opstats_*.csv); the fit itself is
host work and must reproduce the committed coefficients, and the fitted model -- with the constant
terms scaled by the level-batching share the GPU evaluation picked (profiles/r06_model_batch_sweep.txt) --
IS the built-in model_Hip_MI355X: both must drive the supernode merges to the same partitions."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import baspacho_amd as B  # noqa: E402
from baspacho_amd import testing as T  # noqa: E402
import fit_computation_model as F  # noqa: E402

PREFIX = os.path.join(ROOT, "profiles", "r06_opstats")   # (round 6: re-dumped on this round's kernels)
LEVEL_BATCHING_SHARE = 0.03   # = kLevelBatchingShare of csrc/computation_model.cpp


def _model(fit, s):
    m = []
    for k in ("potrf", "trsm", "syge", "asmbl"):
        p = [float(v) for v in fit[k][0]]
        p[0] *= s
        m += p
    return m


def test_fit_reproduces_committed_coefficients():
    fit = F.fit_all(PREFIX)
    want = json.load(open(os.path.join(ROOT, "profiles", "r06_model_fit.json")))
    for k in ("potrf", "trsm", "syge", "asmbl"):
        c, med, p90, n = fit[k]
        assert n == want[k]["samples"] and n > 1000
        assert np.allclose(c, want[k]["params"], rtol=1e-6, atol=1e-20), k
        assert np.all(c >= 0)
        assert med < 0.25, (k, med)         # the polynomial models describe the samples (r06: 0.03 .. 0.20)
    # ... also at the top of the sampled range (the widest fronts), not only on average
    smp = F.load(PREFIX + "_syge.csv", 4)
    top = smp[np.argsort(smp[:, 0] * smp[:, 1] * smp[:, 2])[-40:]]
    pred = F.basis_syge(top[:, 0], top[:, 1], top[:, 2]) @ fit["syge"][0]
    assert np.median(np.abs(pred - top[:, 3]) / top[:, 3]) < 0.5


def test_fitted_model_reproduces_todays_merges():
    fit = F.fit_all(PREFIX)
    model = _model(fit, LEVEL_BATCHING_SHARE)
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=120, num_pts=8000, band=16)
    probs = {"bal": (sizes, ss, [0, 8000]),
             "grid": (np.full(40 * 40, 3, dtype=np.int64), T.gen_grid(40, 40, 1.0, 2, 37), []),
             "flat": (np.full(3000, 3, dtype=np.int64), T.gen_flat(3000, 1.5e-3, 37), [])}
    for name, (sz, st, ranges) in probs.items():
        a = B.create_solver(B.Settings(), sz, st, ranges)
        b = B.create_solver(B.Settings(computationModel=model), sz, st, ranges)
        dense_a = a.numLumps() - (ranges[1] if ranges else 0)
        dense_b = b.numLumps() - (ranges[1] if ranges else 0)
        # (the built-in constants are this fit printed with 16 digits)
        assert dense_a == dense_b, (name, dense_a, dense_b)
        assert a.dataSize() == b.dataSize(), (name, a.dataSize(), b.dataSize())
    # bundle adjustment: the cameras stay a handful of supernodes (this small problem's camera graph
    # is banded, not complete; a complete one is ONE supernode by the dense-merge rule)
    a = B.create_solver(B.Settings(computationModel=model), *probs["bal"])
    assert a.numLumps() - 8000 <= 4
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=60, num_pts=20000, band=60, seed=3)
    b = B.create_solver(B.Settings(computationModel=model), sizes, ss, [0, 20000])
    assert b.numLumps() == 20001
