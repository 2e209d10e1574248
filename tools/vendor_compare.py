#!/usr/bin/env python
"""Vendor comparators for the dense phase (TOOLS ONLY -- nothing under baspacho_amd/ links or loads
rocSOLVER / rocBLAS; the reference ships the same kind of harness against CHOLMOD,
benchmarking/BenchCholmod.cpp, Bench.cpp:362-373): on the same GPU,
  * rocsolver_dpotrf of a dense n x n SPD matrix (n = 7839: the camera block of the BAL-871 stand-in,
    whose factorisation IS this library's dense phase; n = 14000: the root front of FLAT-50k),
  * rocblas_dsyrk C -= A A^T with k = 256 (the rank-256 lookahead update) on an n x n lower triangle.
Bound through ctypes; torch only holds the device buffers.
usage: python tools/vendor_compare.py [--json]"""
import ctypes
import json
import sys
import time

import torch

FILL_UPPER, FILL_LOWER = 121, 122
OP_NONE, OP_T = 111, 112


def _libs():
    rb = ctypes.CDLL("librocblas.so", mode=ctypes.RTLD_GLOBAL)
    rs = ctypes.CDLL("librocsolver.so", mode=ctypes.RTLD_GLOBAL)
    return rb, rs


def _time(fn, reps):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def run(n_list=(7839, 14000), syrk=((7839, 256), (14000, 256)), reps=5):
    rb, rs = _libs()
    handle = ctypes.c_void_p()
    assert rb.rocblas_create_handle(ctypes.byref(handle)) == 0
    out = {"device": torch.cuda.get_device_name(0), "rocsolver_dpotrf": [], "rocblas_dsyrk": []}
    for n in n_list:
        g = torch.Generator(device="cuda").manual_seed(1)
        M = torch.rand((n, n), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
        A0 = torch.tril(M) + torch.tril(M, -1).T + 1.2 * n * torch.eye(n, dtype=torch.float64, device="cuda")
        del M
        A = A0.clone()
        info = torch.zeros(1, dtype=torch.int32, device="cuda")

        def potrf():
            # column-major lower == row-major upper; the matrix is symmetric, either triangle works
            rc = rs.rocsolver_dpotrf(handle, ctypes.c_int(FILL_LOWER), ctypes.c_int(n),
                                     ctypes.c_void_p(A.data_ptr()), ctypes.c_int(n),
                                     ctypes.c_void_p(info.data_ptr()))
            assert rc == 0, rc
        potrf()                                   # warm-up (workspace, kernels)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            A.copy_(A0)
            ts.append(_time(potrf, 1))
        ts.sort()
        sec = ts[len(ts) // 2]
        # check: column-major lower of A (= row-major upper) is L^T stored row-major -> U with U^T U = A0
        U = torch.triu(A)
        x = torch.rand(n, dtype=torch.float64, device="cuda")
        err = float((U.T @ (U @ x) - A0 @ x).norm() / (A0 @ x).norm())
        out["rocsolver_dpotrf"].append({"n": n, "ms": round(sec * 1e3, 3), "TFs": round(n ** 3 / 3 / sec / 1e12, 2),
                                        "info": int(info.item()), "rel_err": err})
        del A, A0, U
        torch.cuda.empty_cache()
    for n, k in syrk:
        P = torch.rand((n, k), dtype=torch.float64, device="cuda")
        C = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        alpha, beta = ctypes.c_double(-1.0), ctypes.c_double(1.0)

        def dsyrk():
            # row-major P (n x k) = column-major k x n: C -= P P^T is op = T on the column-major view
            rc = rb.rocblas_dsyrk(handle, ctypes.c_int(FILL_UPPER), ctypes.c_int(OP_T), ctypes.c_int(n),
                                  ctypes.c_int(k), ctypes.byref(alpha), ctypes.c_void_p(P.data_ptr()),
                                  ctypes.c_int(k), ctypes.byref(beta), ctypes.c_void_p(C.data_ptr()),
                                  ctypes.c_int(n))
            assert rc == 0, rc
        dsyrk()
        sec = _time(dsyrk, reps)
        out["rocblas_dsyrk"].append({"n": n, "k": k, "ms": round(sec * 1e3, 3),
                                     "TFs": round(n * n * k / sec / 1e12, 2)})   # n^2 k flops (one triangle)
        del P, C
        torch.cuda.empty_cache()
    rb.rocblas_destroy_handle(handle)
    return out


if __name__ == "__main__":
    r = run()
    if "--json" in sys.argv:
        print(json.dumps(r))
    else:
        for e in r["rocsolver_dpotrf"]:
            print("rocsolver_dpotrf n=%d: %.3f ms = %.2f TF/s (info %d, rel err %.1e)" % (
                e["n"], e["ms"], e["TFs"], e["info"], e["rel_err"]))
        for e in r["rocblas_dsyrk"]:
            print("rocblas_dsyrk n=%d k=%d: %.3f ms = %.2f TF/s" % (e["n"], e["k"], e["ms"], e["TFs"]))
