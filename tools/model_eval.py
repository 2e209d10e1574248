#!/usr/bin/env python
"""factor() time of the benchmark structures under candidate merge models: the built-in
model_Hip_MI355X against a fitted model (JSON line of tools/fit_computation_model.py) whose constant
terms are scaled by s -- a level of the fused path batches the ops of all its lumps into one launch,
so an op's marginal fixed cost there is a fraction of what it costs as a launch of its own (which is
what the per-op samples measure).
usage: python tools/model_eval.py fit.json [workloads...]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import baspacho_amd as B  # noqa: E402
import bench  # noqa: E402
from baspacho_amd import testing as T  # noqa: E402


def model_from_fit(fit, s):
    m = []
    for k in ("potrf", "trsm", "syge", "asmbl"):
        p = list(fit[k]["params"])
        p[0] *= s
        m += p
    return m


def run(name, sizes, ss, ranges, model, label, batch=1):
    sol = B.create_solver(B.Settings(computationModel=model), sizes, ss, ranges)
    sol.setStream(torch.cuda.current_stream())
    h = T.random_data(sol.dataSize(), -1, 1, 37)
    sol.damp(h, 0.0, sol.order() * 1.2)
    A = torch.from_numpy(h).cuda()
    mk = (lambda: [A.clone() for _ in range(batch)]) if batch > 1 else (lambda: A.clone())
    bufs = [mk() for _ in range(8)]
    sol.factor(bufs[0])
    sol.factor(bufs[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, 8):
        sol.factor(bufs[i])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    print("%-10s %-14s %9.3f ms  lumps %6d  data %8.1f MB  flops %9.2f GF" % (
        name, label, ms, sol.numLumps(), sol.dataSize() * 8 / 1e6, sol.factorFlops() / 1e9), flush=True)
    return ms


def main():
    fit = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    names = sys.argv[2:] or ["tridiag", "grid82", "grid82x16", "flat50k", "bal871"]
    for name in names:
        batch = 1
        base = name
        if "x" in name and name.split("x")[-1].isdigit():
            base, batch = name.rsplit("x", 1)[0], int(name.rsplit("x", 1)[1])
        sizes, ss, ranges, _, _ = bench.build_problem(base)
        run(name, sizes, ss, ranges, None, "built-in", batch)
        for s in (1.0, 0.3, 0.1, 0.03, 0.01):
            run(name, sizes, ss, ranges, model_from_fit(fit, s), "fit, a*%g" % s, batch)


if __name__ == "__main__":
    main()
