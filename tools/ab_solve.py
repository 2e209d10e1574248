"""A/B of solve() schedule switches (environment overrides read at Solver creation) on tree-structured
problems: median of `reps` solve() calls with 1 and 10 right-hand sides on a factor computed once, the
error against the first setting's solution, and the run counters.  Usage (GPU box):
python tools/ab_solve.py "BSP_SOLVE_FUSED=0" -"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import torch

import baspacho_amd as B
from baspacho_amd import testing as T
import bench


def main():
    reps = 25
    settings = [dict(kv.split("=") for kv in a.split()) if (a != "-" and not a.startswith("--")) else {} for a in (sys.argv[1:] or ["-"])]
    device = torch.device("cuda:0")
    which = None
    if sys.argv[1:] and sys.argv[1].startswith("--probs="):
        which = sys.argv[1].split("=", 1)[1].split(",")
        settings = settings[1:]
    probs = {"grid82": lambda sd: (np.full(82 * 82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, sd), None)}
    for k, v in bench.ref_suite_problems().items():
        probs[k[:2]] = (lambda f: (lambda sd: f(sd) + (None,)))(v)
    for w in ("flat50k", "bal871", "bal1723", "bal-small"):
        probs[w] = (lambda name: (lambda sd: bench.build_problem(name)[:3]))(w)
    if which is None:
        which = ["grid82", "10", "21", "30", "31", "33", "40"]
    probs = {k: probs[k] for k in which}
    for name, make in probs.items():
        sizes, ss, ranges = make(37)
        ref_x = {}
        line = []
        for env in settings:
            for k, v in env.items():
                os.environ[k] = v
            if ranges is None:
                sol = B.create_solver(B.Settings(findSparseEliminationRanges=True), sizes, ss)
            else:
                sol = B.create_solver(B.Settings(), sizes, ss, ranges)
            for k in env:
                del os.environ[k]
            sol.setStream(torch.cuda.current_stream(device))
            n = sol.order()
            h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
            sol.damp(h, 0.0, n * 1.2)
            L = torch.from_numpy(h).to(device)
            sol.factor(L)
            out = []
            for nrhs in (1, 10):
                rhs = torch.from_numpy(T.random_data(nrhs * n, -1, 1, 38)).to(device)
                work = rhs.clone()
                sol.solve(L, work, n, nrhs)

                def one():
                    work.copy_(rhs)
                    sol.solve(L, work, n, nrhs)
                t, _ = bench._timed(device, one, reps)
                x = work.cpu().numpy()
                if nrhs not in ref_x:
                    ref_x[nrhs] = x
                err = float(np.linalg.norm(x - ref_x[nrhs]) / np.linalg.norm(ref_x[nrhs]))
                out.append((t * 1e3, err))
            c = sol.runCounters()
            line.append((out, c["solve_wide_launches"], c["sweep_launches"]))
            del sol, L
            torch.cuda.empty_cache()
        b1, b10 = line[0][0][0][0], line[0][0][1][0]
        print("%-40s" % name[:40] + "   ".join(
            "1: %.4f ms (%+5.1f %%) 10: %.4f ms (%+5.1f %%) diff %.0e/%.0e wide %d sweeps %d" % (
                o[0][0], 100 * (o[0][0] / b1 - 1), o[1][0], 100 * (o[1][0] / b10 - 1), o[0][1], o[1][1], f, sw)
            for o, f, sw in line), flush=True)


if __name__ == "__main__":
    main()
