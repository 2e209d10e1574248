#!/usr/bin/env python
"""Per-op timing dump for the supernode-merge cost model (the reference's `bench -Z`,
benchmarking/Bench.cpp:72-124: one CSV row per potrf / trsm / syrk-gemm / assemble call with its
sizes and seconds).  Drives factor() through the per-op NumericCtx boundary with stats on, over a
spread of problems, and writes <out>_{potrf,trsm,syge,asmbl}.csv (tab separated, sizes then
seconds, as the reference's files).  Run on the GPU box:
    python tools/op_stats_dump.py gpurun_out/opstats
then fit with tools/fit_computation_model.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baspacho_amd as B  # noqa: E402
from baspacho_amd import testing as T  # noqa: E402
from baspacho_amd.csrc_models import MODEL_OPENBLAS_I7  # noqa: E402


def problems():
    # structures of the reference's bench suite in small (Bench.cpp:279-330) + BAL-like + chains;
    # two merge models so that both narrow and wide fronts are sampled
    yield "flat", np.full(2500, 3, dtype=np.int64), T.gen_flat(2500, 2.0e-3, 37), []
    yield "flat_b", np.full(1500, 8, dtype=np.int64), T.gen_flat(1500, 3.0e-3, 38), []
    yield "grid", np.full(60 * 60, 3, dtype=np.int64), T.gen_grid(60, 60, 1.0, 2, 37), []
    yield "grid_b", np.full(40 * 40, 6, dtype=np.int64), T.gen_grid(40, 40, 1.0, 3, 39), []
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=200, num_pts=20000, band=24, seed=3)
    yield "bal", sizes, ss, [0, 20000]
    yield "tridiag", np.full(1200, 3, dtype=np.int64), T.block_tridiagonal(1200), []
    # wide fronts, so that the flop terms (n^3, n^2 k, m n k) are sampled and not only the fixed costs
    yield "flat_wide", np.full(6000, 3, dtype=np.int64), T.gen_flat(6000, 1.5e-3, 41), []
    yield "grid_wide", np.full(110 * 110, 3, dtype=np.int64), T.gen_grid(110, 110, 1.0, 3, 43), []
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=420, num_pts=60000, band=40, seed=5)
    yield "bal_wide", sizes, ss, [0, 60000]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/opstats"
    acc = {"potrf": [], "trsm": [], "syge": [], "asmbl": []}
    for name, sizes, ss, ranges in problems():
        for model in (None, MODEL_OPENBLAS_I7):
            sol = B.create_solver(B.Settings(computationModel=model), sizes, ss, ranges)
            data = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
            sol.damp(data, 0.0, sol.order() * 1.2)
            d = torch.from_numpy(data).cuda()
            sol.factorPerOp(d.clone())          # warm-up (code objects, allocator)
            sol.collectOpStats(True)
            sol.factorPerOp(d)
            st = sol.opStats()
            sol.collectOpStats(False)
            for k in acc:
                acc[k].append(st[k])
            print(name, "model" if model is None else "openblas", {k: len(v) for k, v in st.items()},
                  file=sys.stderr)
    for k, parts in acc.items():
        a = np.concatenate(parts, axis=0)
        with open("%s_%s.csv" % (out, k), "w") as f:
            for row in a:
                f.write("\t".join("%d" % v for v in row[:-1]) + "\t%.9e\n" % row[-1])
        print(k, a.shape, file=sys.stderr)


if __name__ == "__main__":
    main()
