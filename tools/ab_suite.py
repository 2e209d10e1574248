"""A/B of schedule switches over the reference's eleven benchmark families (Bench.cpp:290-367, first
instance): for every family and every setting (environment overrides read at Solver creation) the
median of `--reps` warm factor() calls on pristine copies, the vector residual probe, and what the
plan holds.  Usage (GPU box): python tools/ab_suite.py "A=1 B=2" "A=3" ... ('-' = the product)"""
import json
import os
import statistics
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import torch

import baspacho_amd as B
from baspacho_amd import testing as T
import bench


def main():
    reps = 9
    args = [a for a in sys.argv[1:]]
    if args and args[0].startswith("--reps="):
        reps = int(args.pop(0).split("=")[1])
    flt = ""
    if args and args[0].startswith("--filter="):
        flt = args.pop(0).split("=", 1)[1]
    batch = 1
    if args and args[0].startswith("--batch="):
        batch = int(args.pop(0).split("=")[1])
    settings = [dict(kv.split("=") for kv in a.split()) if a != "-" else {} for a in (args or ["-"])]
    device = torch.device("cuda:0")
    probs = dict(bench.ref_suite_problems())
    probs["grid82"] = lambda sd: (np.full(82 * 82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, sd))
    import re
    rx = re.compile(flt)
    for name, make in probs.items():
        if not rx.search(name):
            continue
        sizes, ss = make(37)
        line = []
        for env in settings:
            for k, v in env.items():
                os.environ[k] = v
            sol = B.create_solver(B.Settings(findSparseEliminationRanges=True), sizes, ss)
            for k in env:
                del os.environ[k]
            sol.setStream(torch.cuda.current_stream(device))
            h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
            sol.damp(h, 0.0, sol.order() * 1.2)
            A = torch.from_numpy(h).to(device)
            if batch > 1:  # the same matrix `batch` times per call (timing only; per-matrix time reported)
                bufs = [[A.clone() for _ in range(batch)] for _ in range(reps + 2)]
            else:
                bufs = [A.clone() for _ in range(reps + 2)]
            sol.factor(bufs[0])
            sol.factor(bufs[1])
            it = iter(bufs[2:])
            t, _ = bench._timed(device, lambda: sol.factor(next(it)), reps)
            t /= batch
            res = bench.residual_probe(sol, h, bufs[-1][-1] if batch > 1 else bufs[-1], nprobe=1)
            st = sol.planStats()
            line.append((t * 1e3, res, st["num_tail_panels"], st["num_levels"], st["num_launches"]))
            del sol, A, bufs, it
            torch.cuda.empty_cache()
        base = line[0][0]
        print("%-46s" % name[:46] + "  ".join(
            "%8.4f ms (%+5.1f %%) tail %2d lv %3d ln %4d r %.0e" % (l[0], 100 * (l[0] / base - 1), l[2], l[3], l[4], l[1])
            for l in line), flush=True)


if __name__ == "__main__":
    main()
