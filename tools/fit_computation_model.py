#!/usr/bin/env python
"""Least-squares fit of the supernode-merge cost model from per-op samples -- the role of the
reference's examples/OptimizeCompModel.cpp:64-275 (there a small Gauss-Newton optimizer; the models
are linear in their coefficients, so weighted linear least squares reaches the same minimum):

    potrf(n)       ~ a + b n + c n^2 + d n^3
    trsm(n, k)     ~ a + b n + c n^2 + (d + e n + f n^2) k
    syge(m, n, k)  ~ a + b u + c v + k (d + e u + f v),  u = m + n, v = m n
    asmbl(br, bc)  ~ a + b br + c bc + d br bc

residuals scaled by 1 / sqrt(t) as the reference does (OptimizeCompModel.cpp:82-83), coefficients
constrained to be non-negative (NNLS) so that the merge estimates stay monotone.
usage: python tools/fit_computation_model.py <prefix>      (reads <prefix>_{potrf,trsm,syge,asmbl}.csv)
prints the four parameter arrays (C++ initialiser form) and a JSON summary on the last line."""
import json
import sys

import numpy as np
from scipy.optimize import nnls


def load(path, ncols):
    a = np.loadtxt(path, ndmin=2)
    assert a.shape[1] == ncols, (path, a.shape)
    return a[a[:, -1] > 0]


def basis_potrf(n):
    return np.stack([np.ones_like(n), n, n * n, n ** 3], axis=1)


def basis_trsm(n, k):
    return np.stack([np.ones_like(n), n, n * n, k, n * k, n * n * k], axis=1)


def basis_syge(m, n, k):
    u, v = m + n, m * n
    return np.stack([np.ones_like(u), u, v, k, u * k, v * k], axis=1)


def basis_asmbl(br, bc):
    return np.stack([np.ones_like(br), br, bc, br * bc], axis=1)


def fit(Bm, t):
    w = 1.0 / np.sqrt(t)
    # column scaling keeps NNLS well conditioned (n^3 against 1)
    sc = np.abs(Bm).max(axis=0)
    sc[sc == 0] = 1.0
    c, _ = nnls(Bm * w[:, None] / sc, t * w)
    c = c / sc
    rel = np.abs(Bm @ c - t) / t
    return c, float(np.median(rel)), float(np.percentile(rel, 90))


def fit_all(prefix):
    out = {}
    p = load(prefix + "_potrf.csv", 2)
    out["potrf"] = fit(basis_potrf(p[:, 0]), p[:, 1]) + (len(p),)
    p = load(prefix + "_trsm.csv", 3)
    out["trsm"] = fit(basis_trsm(p[:, 0], p[:, 1]), p[:, 2]) + (len(p),)
    p = load(prefix + "_syge.csv", 4)
    out["syge"] = fit(basis_syge(p[:, 0], p[:, 1], p[:, 2]), p[:, 3]) + (len(p),)
    p = load(prefix + "_asmbl.csv", 3)
    out["asmbl"] = fit(basis_asmbl(p[:, 0], p[:, 1]), p[:, 2]) + (len(p),)
    return out


def main():
    res = fit_all(sys.argv[1])
    summary = {}
    for k in ("potrf", "trsm", "syge", "asmbl"):
        c, med, p90, n = res[k]
        print("%sParams = { %s };   // %d samples, rel. error median %.2f, p90 %.2f"
              % (k, ", ".join("%.6e" % v for v in c), n, med, p90))
        summary[k] = {"params": [float(v) for v in c], "samples": n, "rel_err_median": round(med, 3),
                      "rel_err_p90": round(p90, 3)}
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
