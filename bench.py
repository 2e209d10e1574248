#!/usr/bin/env python
"""bench.py -- fp64 factor() throughput of the MI355X backend on the BAL-871-shaped Schur problem.

A "step" = one Solver::factor() of one matrix resident in HBM (per rank).  Protocol follows the
reference's BAL_bench (benchmarking/BaAtLargeBench.cpp:44-97,155-171): structure only from the
problem (points first, size 3; cameras after, size 9; one block per observation), sparse
elimination range {0, numPts}, numeric data = uniform(-1,1) seed 37 then damp(0, 1.2*order),
warm-up factor, timed factor on device-resident data (H2D excluded).  The BAL file is not
available offline, so the structure comes from the committed synthetic generator
(baspacho_amd/testing.py: gen_bal_synthetic, 871 cameras / 527480 points / 2.79 M observations).

N>1: one process per GPU; rank 0 runs the symbolic analysis and broadcasts the flat symbolic
plan over RCCL; every rank then factors its own matrix of that structure (batched mode sharded
over the GPUs, no collective inside a factorisation) -> weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import baspacho_amd as B  # noqa: E402
from baspacho_amd import testing as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DT = "double"
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix (= vector) peak, AMD datasheet
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_problem(name):
    if name == "bal871":
        sizes, ss, cam, pt = T.gen_bal_synthetic()
        return sizes, ss, [0, 527480], "synthetic BAL-871 stand-in: 871 cams x 9, 527480 pts x 3, %d obs" % len(cam)
    if name == "bal-small":
        sizes, ss, cam, pt = T.gen_bal_synthetic(num_cams=120, num_pts=40000, band=16)
        return sizes, ss, [0, 40000], "synthetic BAL-like: 120 cams, 40000 pts, %d obs" % len(cam)
    if name == "flat50k":
        ss = T.gen_flat(16667, 3.0e-4, 37)
        return np.full(16667, 3, dtype=np.int64), ss, [], "FLAT 16667 x 3, fill 3e-4"
    if name == "grid82":
        ss = T.gen_grid(82, 82, 1.0, 2, 37)
        return np.full(82 * 82, 3, dtype=np.int64), ss, [], "GRID 82x82 conn 2 x 3"
    if name == "tridiag":
        return np.full(3334, 3, dtype=np.int64), T.block_tridiagonal(3334), [], "block-tridiagonal 3334 x 3"
    raise ValueError(name)


def residual_probe(sol, host_A, L_dev, nprobe=2, seed=5):
    """size-independent check of L L^T = A at FULL size: ||L (L^T x) - A x|| / ||A x|| for random
    x, A and L applied as block-sparse operators through the skeleton (checker = oracle/)"""
    from oracle import cref
    L = L_dev.cpu().numpy()
    skh = cref.SkelHandle(sol.skel())
    worst = 0.0
    for p in range(nprobe):
        x = T.random_data(sol.order(), -1, 1, seed + p)
        worst = max(worst, float(cref.probe_residual(skh, host_A, L, x)))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bal871")
    ap.add_argument("--batch", type=int, default=0,
                    help="total number of identical-structure matrices (sharded over the GPUs); "
                         "0 = one matrix per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, "launch with torchrun --nproc-per-node == --gpus"

    # ---- symbolic analysis on rank 0, plan broadcast over RCCL ------------------------------
    t_sym = 0.0
    desc = ""
    sol = None
    if rank == 0:
        sizes, ss, ranges, desc = build_problem(args.workload)
        t0 = time.time()
        sol = B.create_solver(B.Settings(), sizes, ss, ranges)
        t_sym = time.time() - t0
    if world > 1:
        from baspacho_amd.distributed import broadcast_solver
        sol = broadcast_solver(sol, src=0, device=device)
    sol.setStream(torch.cuda.current_stream(device))

    # ---- numeric data: K+W pristine copies of this rank's matrices resident in HBM ------------
    order, flops = sol.order(), sol.factorFlops()
    from baspacho_amd.distributed import shard_batch
    total_batch = args.batch if args.batch > 0 else world
    q0, q1 = shard_batch(total_batch, world, rank)
    n_local = q1 - q0
    # seeds / damping as the reference's batched bench (Bench.cpp:227-233): seed 37+q
    hosts = []
    for q in range(q0, q1):
        h = T.random_data(sol.dataSize(), -1.0, 1.0, 37 + q)
        sol.damp(h, 0.0, order * (1.2 if args.batch == 0 else 1.3))
        hosts.append(h)
    host = hosts[0]
    A_devs = [torch.from_numpy(h).to(device) for h in hosts]
    A_dev = A_devs[0]
    n_buf = args.steps + args.warmup
    if args.batch == 0:
        bufs = [A_dev.clone() for _ in range(n_buf)]
    else:
        bufs = [[a.clone() for a in A_devs] for _ in range(n_buf)]   # batched factor() per step
    torch.cuda.synchronize(device)

    for i in range(args.warmup):
        sol.factor(bufs[i])
    torch.cuda.synchronize(device)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        sol.factor(bufs[args.warmup + i])
    torch.cuda.synchronize(device)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = 1e3 * elapsed / args.steps
    value = total_batch * flops * args.steps / elapsed / 1e9

    out = {
        "metric": "factor_gflops_fp64_bal871_schur" if args.workload == "bal871"
        else "factor_gflops_fp64_" + args.workload,
        "value": round(value, 2), "unit": "GF/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "description": desc, "order": order,
                   "data_MB": round(sol.dataSize() * 8 / 1e6, 1), "matrices_per_gpu": n_local,
                   "batch": total_batch,
                   "factor_GF_per_matrix": round(flops / 1e9, 3),
                   "parallelism": "batch-shard x%d (plan broadcast over RCCL)" % world},
    }

    if rank == 0:
        # ---- parity at full size: residual probe of the last timed factor -------------------
        last = bufs[-1] if args.batch == 0 else bufs[-1][0]
        out["residual_probe"] = residual_probe(sol, host, last)
        st = sol.planStats()
        out["plan"] = {k: st[k] for k in ("num_launches", "num_levels", "num_panels",
                                          "num_upd_tasks", "num_atomic_upd_tasks")}
        out["analysis_s"] = round(t_sym, 3)

        # ---- roofline of the dominant kernel, HIP events on the execution stream ------------
        if not args.no_profile:
            work = A_dev.clone()
            prof = sol.factorProfiled(work)
            tot = sum(v[0] for v in prof.values())
            out["kernel_ms"] = {k: [round(v[0], 4), v[1]] for k, v in prof.items()}
            # algorithmic work of every kernel class, from the plan (DESIGN.md "Kernels"):
            #   update / chain_update: 2 K per lower-trapezoid target element (symmetric update)
            #   elim_update: the source blocks of every pair read once (the operands of the
            #     products) + one read-modify-write of every target element
            #   elim_factor: every element of the eliminated columns read and written once
            pair_src_bytes = 8.0 * 2.0 * st["elim_pair_operand_elems"]
            work_of = {
                "update": ("updateTileBulk|updateTile<%s>" % DT, "mfma", st["upd_flops"] - st["upd_flops_direct"]),
                # (one-panel levels: update tiles + the next panel's potrf, and inside an outer block
                #  also the panel's trsm, in one launch)
                "chain_update": ("chainStep|updateTileDirectPotrf<%s>" % DT, "mfma",
                                 st["upd_flops_direct"] + st["trsm_flops_merged"] + st["potrf_flops_fused"]),
                "elim_update": ("elimGatherMfma<%s>" % DT, "hbm",
                                pair_src_bytes + 16.0 * st["elim_target_elems"]),
                "elim_factor": ("elimFactorTiny|elimFactorSmall<%s>" % DT, "hbm", 16.0 * st["elim_col_elems"]),
                "trsm": ("trsmPanel<%s>" % DT, "mfma", st["trsm_flops"] - st["trsm_flops_merged"]),
                "potrf": ("potrfPanel<%s>" % DT, "mfma", st["potrf_flops"] - st["potrf_flops_fused"]),
            }
            # dominant class = largest event time; classes within 5 % of it count as tied and the
            # one carrying more of the algorithmic work wins (all classes are in kernel_rates)
            tmax = max(v[0] for v in prof.values())
            share = {k: work_of[k][2] / (flops if work_of[k][1] == "mfma" else 16.0 * sol.dataSize())
                     for k in prof}
            dom = max((k for k in prof if prof[k][0] >= 0.95 * tmax), key=lambda k: share[k])
            ms, launches = prof[dom]
            kname, bound, amount = work_of[dom]
            rate = amount / (ms * 1e-3) / (1e12 if bound == "mfma" else 1e9)
            peak = PEAK_FP64_MFMA_TFLOPS if bound == "mfma" else PEAK_HBM_GBS
            roof = {"kernel": kname, "bound": bound, "achieved": round(rate, 3), "peak": peak,
                    "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                    "frac": round(rate / peak, 4), "traffic": None, "launches": launches,
                    "avg_launch_ms": round(ms / max(launches, 1), 5),
                    ("algorithmic_flops_per_launch" if bound == "mfma"
                     else "algorithmic_bytes_per_launch"): amount / max(launches, 1),
                    "share_of_factor": round(ms / tot, 3)}
            # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as
            # MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as reported), same workload
            try:
                with open(os.path.join(HERE, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f)
                names = kname.split("<")[0].split("|")
                ents = [pmc.get(args.workload, {}).get(n) for n in names]
                ents = [e for e in ents if e]
                if ents:
                    ent = {k: sum(e[k] for e in ents) for k in ("fetch_KB_per_factor", "write_KB_per_factor")}
                    roof["traffic"] = (2.0 * ent["fetch_KB_per_factor"] +
                                       ent["write_KB_per_factor"]) * 1024.0 / max(launches, 1)
                    roof["traffic_source"] = pmc.get("_source", "profiles/pmc_traffic.json")
            except (OSError, ValueError):
                pass
            if bound == "mfma":
                # what back-to-back fp64 MFMAs sustain on this very GPU (register-only probe)
                probe = B.probe_mfma_f64_tflops()
                roof["measured_mfma_probe"] = round(probe, 2)
                roof["frac_of_probe"] = round(rate / probe, 4)
            out["roofline"] = roof
            # every kernel class, for the record
            sec = {}
            for k, (nm, bd, amt) in work_of.items():
                if prof[k][0] > 0 and amt > 0:
                    r = amt / (prof[k][0] * 1e-3) / (1e12 if bd == "mfma" else 1e9)
                    sec[k] = {"kernel": nm, "rate": round(r, 3 if bd == "mfma" else 1),
                              "unit": "TFLOP/s" if bd == "mfma" else "GB/s"}
            out["kernel_rates"] = sec
            # whole-factor roofline: max(flops / MFMA peak, compulsory bytes / HBM peak)
            t_roof = max(flops / (PEAK_FP64_MFMA_TFLOPS * 1e12),
                         16.0 * sol.dataSize() / (PEAK_HBM_GBS * 1e9))
            out["factor_roofline_frac"] = round(t_roof / (ms_per_step * 1e-3), 4)

        # ---- device solve() on the last factor (not part of the metric; BENCHMARK_RESULTS.md of the
        #      reference reports solve-1 timings separately too)
        try:
            rhs = torch.from_numpy(T.random_data(order, -1, 1, 38)).to(device)
            work = rhs.clone()
            sol.solve(last, work, order, 1)  # warm-up
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            work.copy_(rhs)
            sol.solve(last, work, order, 1)
            torch.cuda.synchronize(device)
            out["solve1_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
        except Exception as e:
            out["solve1_ms"] = "error: %s" % e

        # ---- CPU baseline: the BackendFast restatement (oracle/blas_factor.c) on host cores --
        if world == 1 and not args.no_cpu_baseline:
            try:
                from oracle import cref
                # small problems run slower on a wide BLAS team (per-call fork/join on tiny fronts):
                # about one thread per 0.5 GF of work, all cores for the headline workload
                cores = max(1, min(os.cpu_count() or 1, int(flops / 0.5e9)))
                _, blas_desc = cref.blas_lib(cores)
                hostA = host.copy()
                skh = cref.SkelHandle(sol.skel())
                t0 = time.perf_counter()
                elim_s = cref.blas_factor(skh, hostA, sol.sparseEliminationRanges(), cores)
                dt = time.perf_counter() - t0
                out["cpu_baseline"] = {
                    "value": round(flops / dt / 1e9, 2), "unit": "GF/s", "cores": cores,
                    "kind": "port", "seconds": round(dt, 3), "elim_seconds": round(elim_s, 3),
                    "sample": "1 full factor() of the same matrix (same plan, same data)",
                    "blas": blas_desc}
                out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 2)
            except Exception as e:  # the baseline is a report, never a reason to fail the bench
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))

    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
