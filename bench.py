#!/usr/bin/env python
"""bench.py -- fp64 factor() throughput of the MI355X backend.

A "step" = one Solver::factor() of this rank's matrices, resident in HBM.  Protocol follows the
reference's benches (benchmarking/BaAtLargeBench.cpp:44-97,155-171, Bench.cpp:150-160,227-233):
structure only from the problem, numeric data = uniform(-1,1) seed 37 (+q for matrix q of a batch)
then damp(0, 1.2*order) (1.3 for batches), warm-up factors, timed factors on device-resident data
(H2D excluded), every timed step on a pristine copy.

--gpus 1 (default): the headline config, BAL-871 Schur problem (BASELINE.json configs[2]):
  points first (size 3), cameras after (size 9), one block per observation, sparse elimination
  range {0, numPts}.  The BAL file is not available offline: the structure comes from the committed
  synthetic generator (baspacho_amd/testing.py: gen_bal_synthetic, 871 cameras / 527480 points /
  2.79 M observations) unless --bal-file PATH supplies a real one (loader: baspacho_amd/bal.py).
  Extra workloads measured in the same run and reported inside the same JSON line: "batched" =
  BASELINE configs[3] (64 x GRID 82x82 x 3, one batched factor() per step) and "c5" = BAL-1723
  shaped fp32 factor + fp64 iterative refinement (configs[4]).
--gpus N > 1 (torchrun, one process per GPU; or plain `python bench.py --gpus N`, which spawns its
  ranks): the SAME headline metric, one BAL-871 matrix per GPU -- a single factorisation is not
  sharded (no exchange step: replicas), "scaling": "weak", value = N matrices' flops / max-over-ranks
  time.  The batched config of BASELINE's metric ("batch throughput at 1/2/4/8 GPUs") rides along in
  "batched" at every N: rank 0 runs the symbolic analysis, broadcasts the flat plan over RCCL, the 64
  matrices are sharded contiguously (no collective inside a factorisation), "scaling": "strong";
  "batched_1gpu" = rank 0 alone factoring all 64 in the same job (the 1-GPU point of that curve).
  `--workload grid82 --batch 64` makes the batched config the headline line instead.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import baspacho_amd as B  # noqa: E402
from baspacho_amd import testing as T  # noqa: E402
from baspacho_amd.distributed import broadcast_solver, shard_batch  # noqa: E402

HERE = ROOT
DT = "double"
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix (= vector) peak, AMD datasheet
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_problem(name, bal_file=None):
    """returns (paramSizes, SparseStructure, sparseElimRanges, description, data source)"""
    if name == "bal871" and bal_file:
        from baspacho_amd import bal
        prob = bal.load_bal(bal_file)
        sizes, ss, ranges = bal.bal_structure(prob)
        return sizes, ss, ranges, "BAL file %s: %d cams x 9, %d pts x 3, %d obs" % (
            os.path.basename(bal_file), prob.num_cams, prob.num_pts, len(prob.obs_cam)), "file"
    if name == "bal871":
        sizes, ss, cam, pt = T.gen_bal_synthetic()
        return sizes, ss, [0, 527480], ("synthetic BAL-871 stand-in: 871 cams x 9, 527480 pts x 3, "
                                        "%d obs" % len(cam)), "synthetic"
    if name == "bal871-clustered":   # generator sensitivity (round 6): clustered co-visibility
        sizes, ss, cam, pt = T.gen_bal_clustered()
        return sizes, ss, [0, 527480], ("synthetic BAL-871 stand-in, CLUSTERED co-visibility (clusters of 8-30 "
                                        "cameras, power-law camera degree): %d obs" % len(cam)), "synthetic"
    if name == "bal871-banded":      # ... and a narrow band with few loop closures
        sizes, ss, cam, pt = T.gen_bal_synthetic(band=16, far_prob=0.005)
        return sizes, ss, [0, 527480], ("synthetic BAL-871 stand-in, band +-16, far_prob 0.005: %d obs"
                                        % len(cam)), "synthetic"
    if name == "bal1723":
        sizes, ss, cam, pt = T.gen_bal_synthetic(num_cams=1723, num_pts=156502, mean_track=4.4,
                                                 band=24, far_prob=0.02, seed=41)
        return sizes, ss, [0, 156502], ("synthetic BAL-1723 stand-in: 1723 cams x 9, 156502 pts x 3, "
                                        "%d obs" % len(cam)), "synthetic"
    if name == "bal-small":
        sizes, ss, cam, pt = T.gen_bal_synthetic(num_cams=120, num_pts=40000, band=16)
        return sizes, ss, [0, 40000], "synthetic BAL-like: 120 cams, 40000 pts, %d obs" % len(cam), "synthetic"
    if name == "flat50k":
        ss = T.gen_flat(16667, 3.0e-4, 37)
        return np.full(16667, 3, dtype=np.int64), ss, [], "FLAT 16667 x 3, fill 3e-4", "synthetic"
    if name == "grid82":
        ss = T.gen_grid(82, 82, 1.0, 2, 37)
        return np.full(82 * 82, 3, dtype=np.int64), ss, [], "GRID 82x82 conn 2 x 3", "synthetic"
    if name == "tridiag":
        return (np.full(3334, 3, dtype=np.int64), T.block_tridiagonal(3334), [],
                "block-tridiagonal 3334 x 3", "synthetic")
    raise ValueError(name)


def residual_probe(sol, host_A, L_dev, nprobe=2, seed=5):
    """size-independent check of L L^T = A at FULL size: the VECTOR probe ||L (L^T x) - A x|| /
    ||A x|| for random x, A and L applied as block-sparse operators through the skeleton
    (checker = oracle/; the Frobenius form ||L L^T - A||_F / ||A||_F is what the dense-size tests
    in tests/ check)"""
    from oracle import cref
    L = L_dev.cpu().numpy()
    skh = cref.SkelHandle(sol.skel())
    worst = 0.0
    for p in range(nprobe):
        x = T.random_data(sol.order(), -1, 1, seed + p)
        worst = max(worst, float(cref.probe_residual(skh, host_A, L, x)))
    return worst


class Runner:
    """this rank's share of a workload: solver, pristine matrices in HBM, timed factor steps"""

    def __init__(self, ctx, workload, total_batch, batched, bal_file=None, participate=True):
        self.ctx, self.workload, self.batched = ctx, workload, batched
        rank, world, device = ctx["rank"], ctx["world"], ctx["device"]
        self.t_sym, self.desc, self.source = 0.0, "", "synthetic"
        sol = None
        if rank == 0:
            sizes, ss, ranges, self.desc, self.source = build_problem(workload, bal_file)
            t0 = time.time()
            sol = B.create_solver(B.Settings(), sizes, ss, ranges)
            self.t_sym = time.time() - t0
        self.bcast = None
        if world > 1 and participate:
            self.bcast = {}
            sol = broadcast_solver(sol, src=0, device=device, stats=self.bcast)
        self.sol = sol
        if sol is None:
            return
        sol.setStream(torch.cuda.current_stream(device))
        self.order, self.flops = sol.order(), sol.factorFlops()
        self.total_batch = total_batch
        self.q0, self.q1 = shard_batch(total_batch, world, rank) if participate else (0, total_batch)
        # seeds / damping as the reference's benches: seed 37 (+q), damp 1.2 (1.3 for batches)
        self.hosts = []
        for q in range(self.q0, self.q1):
            h = T.random_data(sol.dataSize(), -1.0, 1.0, 37 + q)
            sol.damp(h, 0.0, self.order * (1.3 if batched else 1.2))
            self.hosts.append(h)
        self.A_devs = [torch.from_numpy(h).to(device) for h in self.hosts]

    def run(self, steps, warmup, sync_ranks=True):
        """W untimed + K timed factor() calls, barrier + synchronize on both sides, max over ranks"""
        dist, device = (self.ctx["dist"] if sync_ranks else None), self.ctx["device"]
        n_buf = steps + warmup
        if self.batched:
            bufs = [[a.clone() for a in self.A_devs] for _ in range(n_buf)]
        else:
            bufs = [self.A_devs[0].clone() for _ in range(n_buf)]
        torch.cuda.synchronize(device)
        for i in range(warmup):
            self.sol.factor(bufs[i])
        torch.cuda.synchronize(device)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            self.sol.factor(bufs[warmup + i])
        torch.cuda.synchronize(device)
        own = time.perf_counter() - t0   # this rank's K steps, before it waits for the others
        if dist:
            dist.barrier()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        rank_ms = None
        if dist:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            # per-rank view (so that a first multi-GPU curve can be read without a second run): every
            # rank's own time for its K steps, and what it spent on the plan broadcast
            mine = [own] + [float((self.bcast or {}).get(k, 0.0)) for k in ("serialize_s", "broadcast_s", "rebuild_s")]
            allv = [torch.zeros(4, dtype=torch.float64, device=device) for _ in range(self.ctx["world"])]
            dist.all_gather(allv, torch.tensor(mine, dtype=torch.float64, device=device))
            rows = [[float(x) for x in v.tolist()] for v in allv]
            per = [round(1e3 * r[0] / steps, 4) for r in rows]
            rank_ms = {"min": min(per), "max": max(per), "all": per}
            if self.bcast is not None:
                rank_ms["plan_broadcast"] = {
                    "plan_MB": self.bcast.get("plan_MB"), "plan_int64_words": self.bcast.get("plan_int64_words"),
                    "serialize_s_rank0": round(rows[0][1], 4),
                    "broadcast_s_max": round(max(r[2] for r in rows), 4),
                    "rebuild_s_max": round(max(r[3] for r in rows), 4)}
        self.last = bufs[-1][0] if self.batched else bufs[-1]
        if sync_ranks:   # every rank took part: the whole batch, or one matrix per rank
            n_mat = self.total_batch if self.batched else self.ctx["world"]
        else:            # this rank alone
            n_mat = self.total_batch if self.batched else 1
        ms = 1e3 * elapsed / steps
        res = {"value": round(n_mat * self.flops * steps / elapsed / 1e9, 2), "unit": "GF/s",
               "ms_per_step": round(ms, 4), "matrices_per_step": n_mat,
               "matrices_per_s": round(n_mat * steps / elapsed, 2)}
        if rank_ms is not None:
            res["rank_ms_per_step"] = rank_ms
        return res


def physical_cores():
    """(physical cores, hardware threads, model name) of the host"""
    cores, model, threads = set(), "", 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                threads += 1
            elif line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
                cores.add((phys, core))
    except OSError:
        pass
    return (len(cores) or os.cpu_count() or 1), (threads or os.cpu_count() or 1), model


def socket_cpus():
    """{physical id: [first hardware thread of every core]} and the highest clock the host reports (MHz)"""
    socks, seen, mhz = {}, set(), 0.0
    try:
        cpu = phys = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":", 1)[1])
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("cpu MHz"):
                mhz = max(mhz, float(line.split(":", 1)[1]))
            elif line.startswith("core id"):
                key = (phys, line.split(":", 1)[1].strip())
                if key not in seen:
                    seen.add(key)
                    socks.setdefault(phys, []).append(cpu)
    except (OSError, ValueError):
        pass
    try:
        mhz = max(mhz, float(open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq").read()) / 1e3)
    except (OSError, ValueError):
        pass
    return socks, mhz


def _cpu_baseline_once(sol, host, flops):
    """the BackendFast restatement (oracle/blas_factor.c: same call sequence and syrk/gemm rule as
    MatOpsFast.cpp) on the host cores: 1 warm-up + median of 5 full factors of the same matrix, the
    BLAS team pinned to one hardware thread per core of ONE socket when it fits there (run-to-run
    spread of the unpinned team on a two-socket host: 173-210 GF/s in round 3)"""
    from oracle import cref
    n_phys, n_threads, model = physical_cores()
    socks, mhz = socket_cpus()
    # small problems run slower on a wide BLAS team (per-call fork/join on tiny fronts): about one
    # thread per 0.5 GF of work; the headline workload gets every physical core the BLAS accepts
    want = max(1, min(n_phys, int(flops / 0.5e9)))
    # pin BEFORE the BLAS is loaded (it sizes its team from the affinity mask and its threads inherit
    # it): one hardware thread per core of the largest socket -- north_star's "single-socket CPU"
    # baseline; restored afterwards
    # (OPT-IN, CPU_BASELINE_PIN=1: measured on the GPU box -- 2 x EPYC 9575F, socket 0 = CPUs 0-63 -- the
    #  pinned team ran 0.8 ... 10 s per factor against 0.8-1.0 s unpinned: the box is shared, other
    #  tenants' processes run on those cores, and only the free scheduler finds the idle ones)
    pinned, old_aff = None, None
    try:
        best = max(socks.values(), key=len) if socks else []
        if os.environ.get("CPU_BASELINE_PIN") == "1" and best and hasattr(os, "sched_setaffinity"):
            old_aff = os.sched_getaffinity(0)
            allowed = [c for c in best if c in old_aff][:want]
            if allowed:
                os.sched_setaffinity(0, allowed)
                want = len(allowed)
                pinned = "one socket, %d cores, one hardware thread each" % want
    except OSError:
        pinned = None
    _, blas_desc = cref.blas_lib(want)
    cap = want
    for tok in blas_desc.split():
        if tok.startswith("MAX_THREADS="):
            cap = int(tok.split("=")[1])
    used = min(want, cap)
    skh = cref.SkelHandle(sol.skel())
    ranges = sol.sparseEliminationRanges()
    times, elims = [], []
    for it in range(6):
        hostA = host.copy()
        t0 = time.perf_counter()
        es = cref.blas_factor(skh, hostA, ranges, want)
        dt = time.perf_counter() - t0
        if it > 0:      # (first run = warm-up: page faults of the copy, BLAS thread start-up)
            times.append(dt)
            elims.append(es)
        if dt > 20.0:
            break
    if not times:
        times, elims = [dt], [es]
    if old_aff is not None:
        try:
            os.sched_setaffinity(0, old_aff)
        except OSError:
            pass
    dt = statistics.median(times)
    # fp64 peak of the cores used: 2 x 512-bit FMA pipes per core (Zen 4c/5, Skylake-X class) = 32
    # flops per cycle at the highest clock the host reports -- an upper bound, stated beside the number
    peak = used * 32.0 * mhz / 1e3 if mhz > 0 else None
    return {"value": round(flops / dt / 1e9, 2), "unit": "GF/s", "cores": used, "kind": "port",
            "seconds": round(dt, 3), "seconds_all": [round(t, 3) for t in times],
            "spread_pct": round(100.0 * (max(times) - min(times)) / dt, 1),
            "pinned": pinned or "no (team larger than a socket, or affinity not available)",
            "host_peak_gflops": round(peak, 0) if peak else None,
            "host_peak_note": "cores x 32 fp64 flops/cycle (2 x 512-bit FMA) x %.2f GHz" % (mhz / 1e3),
            "frac_of_host_peak": round(flops / dt / 1e9 / peak, 4) if peak else None,
            "elim_seconds": round(statistics.median(elims), 3),
            "sample": "full factor() of the same matrix (same plan, same data): 1 warm-up + median of %d"
                      % len(times),
            "host": {"model": model, "physical_cores": n_phys, "hardware_threads": n_threads},
            "blas": blas_desc, "blas_thread_cap": cap,
            "blas_coretype_env": os.environ.get("OPENBLAS_CORETYPE", "(auto)")}


def cpu_baseline_child(args):
    """child process of cpu_baseline(): OPENBLAS_CORETYPE must be in the environment before the BLAS
    is loaded, so every kernel set is timed in a process of its own (host only: no GPU call)"""
    workload = args.cpu_baseline_child
    sizes, ss, ranges, _, _ = build_problem(workload, args.bal_file)
    settings = B.Settings()
    if os.environ.get("CPU_BASELINE_PLAN") == "reference":
        # the plan the reference would build for its CPU backend: its own OpenBLAS merge model
        # (ComputationModel.cpp:12-21) and none of this build's extra merge rules (the parent sets
        # BSP_DENSE_MERGE_OFF=1: elimination_tree.cpp reads it when the solver is created)
        from baspacho_amd.csrc_models import MODEL_OPENBLAS_I7
        settings = B.Settings(computationModel=MODEL_OPENBLAS_I7)
    sol = B.create_solver(settings, sizes, ss, ranges)
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
    sol.damp(h, 0.0, sol.order() * 1.2)
    res = _cpu_baseline_once(sol, h, sol.factorFlops())
    lump_start = sol.skel()["lumpStart"]
    res["plan_lumps"] = int(sol.numLumps())
    res["plan_widest_lump"] = int(np.max(np.diff(lump_start)))
    res["plan_GF"] = round(sol.factorFlops() / 1e9, 2)
    res["plan_data_MB"] = round(sol.dataSize() * 8 / 1e6, 1)
    print("CPU_BASELINE_CHILD " + json.dumps(res))
    return 0


def cpu_baseline(sol, host, flops, value, workload=None, bal_file=None):
    """CPU baseline = the BLAS restatement of the reference's BackendFast on this host.  The only
    BLAS in the image is the DYNAMIC_ARCH OpenBLAS of the scipy / numpy wheels (MAX_THREADS=64); which
    kernel set it picks is decided at load time, so besides its own choice the sets named in
    CPU_BASELINE_CORETYPES are timed in child processes (OPENBLAS_CORETYPE=...) and the FASTEST
    is reported, with every attempt listed.  In-process when no workload name is given."""
    import subprocess
    out = _cpu_baseline_once(sol, host, flops)
    tried = [{"coretype": "(auto)", "GF/s": out["value"], "blas": out["blas"]}]
    if workload is not None and not bal_file:
        for ct in os.environ.get("CPU_BASELINE_CORETYPES", "ZEN SKYLAKEX").split():
            try:
                env = dict(os.environ, OPENBLAS_CORETYPE=ct, OPENBLAS_VERBOSE="0")
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", workload],
                                   env=env, capture_output=True, text=True, timeout=240)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("CPU_BASELINE_CHILD ")]
                if not line:
                    tried.append({"coretype": ct, "error": (r.stderr or "no output")[-200:]})
                    continue
                c = json.loads(line[-1][len("CPU_BASELINE_CHILD "):])
                tried.append({"coretype": ct, "GF/s": c["value"], "blas": c["blas"]})
                if c["value"] > out["value"]:
                    out = c
            except Exception as e:  # noqa: BLE001
                tried.append({"coretype": ct, "error": repr(e)[:200]})
    out["kernel_sets_tried"] = tried
    # Round 5: the same restatement on the plan a CPU would get.  The numbers above factor the
    # MI355X-tuned partition (one 7839-wide camera lump: 96 % of the work is one dpotrf-shaped block);
    # the reference's CPU backend would merge by its own OpenBLAS model.  Both are reported, the
    # FASTER (in seconds per factor of the same matrix) is `value`, quoted at the headline's flop
    # count so that value / cpu value = the ratio of the two times.
    lump_start = sol.skel()["lumpStart"]
    gpu_plan = {"plan": "MI355X-tuned (model_Hip_MI355X + dense-merge rule; the plan the GPU factors)",
                "lumps": int(sol.numLumps()), "widest_lump": int(np.max(np.diff(lump_start))),
                "plan_GF": round(flops / 1e9, 2), "seconds": out["seconds"], "GF/s": out["value"],
                "frac_of_host_peak": out.get("frac_of_host_peak")}
    plans = [gpu_plan]
    if workload is not None and not bal_file:
        try:
            best_ct = max((t for t in tried if "GF/s" in t), key=lambda t: t["GF/s"])["coretype"]
            env = dict(os.environ, OPENBLAS_VERBOSE="0", CPU_BASELINE_PLAN="reference", BSP_DENSE_MERGE_OFF="1")
            if best_ct != "(auto)":
                env["OPENBLAS_CORETYPE"] = best_ct
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", workload],
                               env=env, capture_output=True, text=True, timeout=400)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("CPU_BASELINE_CHILD ")]
            if line:
                c = json.loads(line[-1][len("CPU_BASELINE_CHILD "):])
                job_rate = round(flops / c["seconds"] / 1e9, 2)   # the headline's flops over ITS time
                peak = c.get("host_peak_gflops")
                plans.append({"plan": "reference's OpenBLAS merge model (ComputationModel.cpp:12-21), no "
                                      "dense-merge rule: what its CPU backend would factor",
                              "lumps": c["plan_lumps"], "widest_lump": c["plan_widest_lump"],
                              "plan_GF": c["plan_GF"], "seconds": c["seconds"],
                              "GF/s_own_flops": c["value"], "GF/s": job_rate,
                              "frac_of_host_peak": round(c["value"] / peak, 4) if peak else None,
                              "spread_pct": c.get("spread_pct"), "blas_coretype_env": c.get("blas_coretype_env")})
                if c["seconds"] < out["seconds"]:
                    for k in ("seconds", "seconds_all", "spread_pct", "elim_seconds", "blas", "blas_coretype_env",
                              "host_peak_gflops", "host_peak_note"):
                        if k in c:
                            out[k] = c[k]
                    out["value"] = job_rate
                    out["frac_of_host_peak"] = round(c["value"] / peak, 4) if peak else None
                    out["sample"] = c["sample"] + " -- on the reference's CPU plan (cpu_baseline.plans[1])"
            else:
                plans.append({"plan": "reference's OpenBLAS merge model", "error": (r.stderr or "no output")[-300:]})
        except Exception as e:  # noqa: BLE001
            plans.append({"plan": "reference's OpenBLAS merge model", "error": repr(e)[:300]})
    out["plans"] = plans
    out["plan_of_value"] = min((p for p in plans if "seconds" in p), key=lambda p: p["seconds"])["plan"]
    out["note"] = ("cores = BLAS threads actually used (the wheels' OpenBLAS is built with MAX_THREADS=%d); "
                   "value = the fastest kernel set of kernel_sets_tried (OPENBLAS_CORETYPE, one process "
                   "each); no system BLAS / MKL / AOCL exists in this image" % out["blas_thread_cap"])
    out["speedup_gpu_over_cpu"] = round(value / out["value"], 2)
    return out


def roofline_block(sol, A_dev, flops, ms_per_step, workload):
    """per-kernel-class times with HIP events on the stream each launch runs on, (a) in the real
    two-stream schedule (in situ: what rocprofv3's kernel trace shows) and (b) serialised; the
    dominant class is the one with the largest IN-SITU time"""
    st = sol.planStats()
    out = {}
    prof_iso = sol.factorProfiled(A_dev.clone(), in_situ=False)
    prof3 = sol.factorProfiled(A_dev.clone(), in_situ=True, busy=True)
    prof = {k: (v[0], v[1]) for k, v in prof3.items()}
    busy = {k: v[2] for k, v in prof3.items()}
    out["kernel_ms"] = {k: [round(v[0], 4), v[1]] for k, v in prof.items()}
    # time during which at least one launch of the class was running (launches of a class that go
    # to different streams overlap: their durations add up to more than this)
    out["kernel_busy_ms"] = {k: round(v, 4) for k, v in busy.items()}
    out["kernel_ms_isolated"] = {k: [round(v[0], 4), v[1]] for k, v in prof_iso.items()}
    # algorithmic work of every kernel class, from the plan (DESIGN.md "Kernels"):
    #   update / chain_update: 2 K per lower-trapezoid target element (symmetric update)
    #   elim_update: every below-diagonal source block read ONCE (distinct source bytes) + one
    #     read-modify-write of every distinct target element
    #   elim_factor: every element of the eliminated columns read and written once
    sk = sol.skel()
    ranges = sol.sparseEliminationRanges()
    diag_elems = 0.0
    for r in range(len(ranges) - 1):
        w = np.diff(sk["lumpStart"][int(ranges[r]):int(ranges[r + 1]) + 1]).astype(np.float64)
        diag_elems += float((w * w).sum())
    elim_src_bytes = 8.0 * (st["elim_col_elems"] - diag_elems)
    work_of = {
        "update": ("updateTileBulk|updateTile<%s>" % DT, "mfma", st["upd_flops"] - st["upd_flops_direct"]),
        # (one-panel levels: update tiles + the next panel's potrf, and inside an outer block also
        #  the panel's trsm, in one launch)
        "chain_update": ("chainStep|updateTileDirectPotrf<%s>" % DT, "mfma",
                         st["upd_flops_direct"] + st["trsm_flops_merged"] + st["potrf_flops_fused"]),
        "elim_update": ("elimGatherMfma<%s>" % DT, "hbm", elim_src_bytes + 16.0 * st["elim_target_elems"]),
        "elim_factor": ("elimFactorTiny|elimFactorSmall<%s>" % DT, "hbm", 16.0 * st["elim_col_elems"]),
        "trsm": ("trsmPanel|trsmPanelPotrf<%s>" % DT, "mfma", st["trsm_flops"] - st["trsm_flops_merged"]),
        "potrf": ("potrfPanel<%s>" % DT, "mfma", st["potrf_flops"] - st["potrf_flops_fused"]),
    }
    dom = max(prof, key=lambda k: prof[k][0])
    ms, launches = prof[dom]
    kname, bound, amount = work_of[dom]
    scale = 1e12 if bound == "mfma" else 1e9
    rate = amount / (ms * 1e-3) / scale
    peak = PEAK_FP64_MFMA_TFLOPS if bound == "mfma" else PEAK_HBM_GBS
    iso_ms = prof_iso[dom][0]
    roof = {"kernel": kname, "bound": bound, "achieved": round(rate, 3), "peak": peak,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "frac": round(rate / peak, 4),
            "frac_isolated": round(amount / (iso_ms * 1e-3) / scale / peak, 4) if iso_ms > 0 else None,
            # the class's launches overlap one another when they run on two auxiliary streams: work
            # over the time during which at least one of them is running
            "frac_busy": round(amount / (busy[dom] * 1e-3) / scale / peak, 4) if busy[dom] > 0 else None,
            "busy_ms": round(busy[dom], 4),
            "traffic": None, "launches": launches,
            "avg_launch_ms": round(ms / max(launches, 1), 5),
            "timing": "HIP events around every launch on the stream it runs on, lookahead "
                      "schedule on (in situ); frac_isolated = same launches serialised",
            ("algorithmic_flops_per_launch" if bound == "mfma" else "algorithmic_bytes_per_launch"):
                amount / max(launches, 1),
            ("algorithmic_flops" if bound == "mfma" else "algorithmic_bytes"): amount,
            "share_of_kernel_time": round(ms / sum(v[0] for v in prof.values()), 3)}
    # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as
    # MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as reported), same workload
    try:
        from baspacho_amd import _lib
        with open(os.path.join(HERE, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        sha = _lib.kernel_source_sha16()
        if pmc.get("_kernel_source_sha16") != sha:
            # counters of another build say nothing about this one's kernels: report none
            roof["traffic_stale"] = ("profiles/pmc_traffic.json was collected on sources %s, this run is %s: "
                                     "re-run profiles/collect_pmc.sh" % (pmc.get("_kernel_source_sha16"), sha))
            pmc = {}
        names = kname.split("<")[0].split("|")
        ents = [pmc.get(workload, {}).get(n) for n in names]
        ents = [e for e in ents if e]
        if ents:
            ent = {k: sum(e[k] for e in ents) for k in ("fetch_KB_per_factor", "write_KB_per_factor")}
            roof["traffic"] = (2.0 * ent["fetch_KB_per_factor"] +
                               ent["write_KB_per_factor"]) * 1024.0 / max(launches, 1)
            roof["traffic_source"] = pmc.get("_source", "profiles/pmc_traffic.json")
    except (OSError, ValueError):
        pass
    # the same fraction recomputed from the committed rocprofv3 kernel stats (avg_ns of the class)
    try:
        with open(os.path.join(HERE, "profiles", "rocprof_roofline.json")) as f:
            rdoc = json.load(f)
        from baspacho_amd import _lib
        rr = rdoc.get(workload, {}) if rdoc.get("_kernel_source_sha16") == _lib.kernel_source_sha16() else {}
        if dom in rr:
            roof["rocprof"] = rr[dom]
    except (OSError, ValueError):
        pass
    if bound == "mfma":
        probe = B.probe_mfma_f64_tflops()   # back-to-back fp64 MFMAs on this very GPU
        roof["measured_mfma_probe"] = round(probe, 2)
        roof["frac_of_probe"] = round(rate / probe, 4)
    out["roofline"] = roof
    sec = {}
    for k, (nm, bd, amt) in work_of.items():
        if prof[k][0] > 0 and amt > 0:
            sc = 1e12 if bd == "mfma" else 1e9
            pk = PEAK_FP64_MFMA_TFLOPS if bd == "mfma" else PEAK_HBM_GBS
            r = amt / (prof[k][0] * 1e-3) / sc
            ent = {"kernel": nm, "rate": round(r, 3 if bd == "mfma" else 1),
                   "unit": "TFLOP/s" if bd == "mfma" else "GB/s", "frac": round(r / pk, 4),
                   "ms": round(prof[k][0], 4), "launches": prof[k][1],
                   ("algorithmic_flops" if bd == "mfma" else "algorithmic_bytes"): amt}
            if prof_iso[k][0] > 0:
                ent["rate_isolated"] = round(amt / (prof_iso[k][0] * 1e-3) / sc, 3 if bd == "mfma" else 1)
            sec[k] = ent
    out["kernel_rates"] = sec
    # whole-factor roofline: max(flops / MFMA peak, compulsory bytes / HBM peak)
    t_roof = max(flops / (PEAK_FP64_MFMA_TFLOPS * 1e12), 16.0 * sol.dataSize() / (PEAK_HBM_GBS * 1e9))
    out["factor_roofline_frac"] = round(t_roof / (ms_per_step * 1e-3), 4)
    # (the same number inside the roofline object: fraction of the bound for the WHOLE factor(),
    #  driver-timed ms_per_step, not only the dominant kernel)
    roof["frac_whole_factor"] = out["factor_roofline_frac"]
    return out


def c5_block(device):
    """BASELINE configs[4]: fp32 factor + fp64 iterative refinement on a BAL-1723 shaped problem,
    all numerics in the library (factor<float>, solve<float>, addMvFrom<double>)"""
    from baspacho_amd.refine import solve_refined
    sizes, ss, ranges, desc, _ = build_problem("bal1723")
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    sol.setStream(torch.cuda.current_stream(device))
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
    sol.damp(h, 0.0, sol.order() * 1.2)
    A = torch.from_numpy(h).to(device)
    b = torch.from_numpy(T.random_data(sol.order(), -1, 1, 38)).to(device)
    tm = {}
    solve_refined(sol, A, b, tol=1e-10)                 # warm-up (plans, solve lists)
    x, iters, hist = solve_refined(sol, A, b, tol=1e-10, timings=tm)
    return {"workload": desc, "order": sol.order(), "factor_f32_ms": round(tm["factor_f32_ms"], 3),
            "refine_ms": round(tm["refine_ms"], 3), "iterations": iters,
            "final_rel_residual": hist[-1], "factor_GF": round(sol.factorFlops() / 1e9, 2),
            "factor_f32_GFs": round(sol.factorFlops() / tm["factor_f32_ms"] / 1e6, 1)}


# ---- the reference's benchmark suite (benchmarking/Bench.cpp:277-409) -----------------------------
def ref_suite_problems():
    """name -> builder(seed) -> (paramSizes, SparseStructure); names, generators and arguments as in
    Bench.cpp:290-367 (where a name and the call disagree -- 21_...schurfill=0.2 passes 0.0002,
    33_GRID_size=200x200 builds 150x150, 4x_MERI...fill=0.1 passes 0.5 -- the CALL is followed)"""
    def fixed(ss, b):
        return np.full(ss.order(), b, dtype=np.int64), ss

    def flat_schur(seed, schur, sfill):
        return fixed(T.add_schur_set(T.gen_flat(1000, 0.1, seed), schur, sfill, seed + 1000), 3)

    def flat_2_5(seed):
        ss = T.gen_flat(2000, 0.03, seed)
        return T.random_vec(ss.order(), 2, 5, seed + 2000), ss

    return {
        "10_FLAT_size=1000_fill=0.1_bsize=3": lambda sd: fixed(T.gen_flat(1000, 0.1, sd), 3),
        "11_FLAT_size=4000_fill=0.01_bsize=3": lambda sd: fixed(T.gen_flat(4000, 0.01, sd), 3),
        "12_FLAT_size=2000_fill=0.03_bsize=2-5": flat_2_5,
        "20_FLAT+SCHUR_size=1000_fill=0.1_bsize=3_schursize=50000_schurfill=0.02":
            lambda sd: flat_schur(sd, 50000, 0.02),
        "21_FLAT+SCHUR_size=1000_fill=0.1_bsize=3_schursize=5000_schurfill=0.2":
            lambda sd: flat_schur(sd, 5000, 0.0002),
        "30_GRID_size=100x100_fill=1.0_conn=2_bsize=3": lambda sd: fixed(T.gen_grid(100, 100, 1.0, 2, sd), 3),
        "31_GRID_size=150x150_fill=1.0_conn=2_bsize=3": lambda sd: fixed(T.gen_grid(150, 150, 1.0, 2, sd), 3),
        "32_GRID_size=200x200_fill=0.25_conn=2_bsize=3": lambda sd: fixed(T.gen_grid(200, 200, 0.25, 2, sd), 3),
        "33_GRID_size=200x200_fill=0.05_conn=3_bsize=3": lambda sd: fixed(T.gen_grid(150, 150, 0.05, 3, sd), 3),
        "40_MERI_size=1500_n=4_hairlen=600_hairs=2_band=120_fill=0.1_bsize=3":
            lambda sd: fixed(T.gen_meridians(4, 1500, 0.5, 120, 600, 2, 2, sd), 3),
        "41_MERI_size=1500_n=7_hairlen=600_hairs=2_band=120_fill=0.1_bsize=3":
            lambda sd: fixed(T.gen_meridians(7, 1500, 0.5, 120, 600, 2, 2, sd), 3),
    }


def _timed(device, fn, reps):
    """median wall time of fn() in seconds, synchronised on both sides (as the reference's hrc::now()
    around a synchronous call)"""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(device)
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def suite_ref(args, device):
    """the reference's `bench` on its eleven problem families with the protocol of benchmarkSolver /
    benchmarkSolverBatched (Bench.cpp:126-268): solver {findSparseEliminationRanges = true, GPU
    backend}, data = uniform(-1,1) seed 37 (+q), damp(0, 1.2 order) (1.3 for batches), factor, solve
    with nRHS 1 / 2 / 10 after a heat-up, batches of 4 / 8 / 16 reported per matrix.  The reference
    times ONE cold factor() per problem; here `factor_first_s` is that number and `factor_s` the
    median of 5 warm calls on pristine copies.  Every factor is checked with the residual probe.
    The published timings (profiles/published_reference_results.json, other hardware -- stated in
    the output) are printed beside ours; no claim is made across hardware."""
    import re
    pub = {}
    try:
        with open(os.path.join(HERE, "profiles", "published_reference_results.json")) as f:
            pub = json.load(f)
    except (OSError, ValueError):
        pass
    rx = re.compile(args.suite_filter)
    rows = []
    iters = max(1, args.suite_iters)

    def one_instance(name, make, seed, with_batches):
        sizes, ss = make(seed)
        t0 = time.perf_counter()
        sol = B.create_solver(B.Settings(findSparseEliminationRanges=True), sizes, ss)
        analysis_s = time.perf_counter() - t0
        sol.setStream(torch.cuda.current_stream(device))
        order, flops = sol.order(), sol.factorFlops()
        h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
        sol.damp(h, 0.0, order * 1.2)
        A = torch.from_numpy(h).to(device)
        bufs = [A.clone() for _ in range(6)]
        first, _ = _timed(device, lambda: sol.factor(bufs[0]), 1)
        it = iter(bufs[1:])
        fac, fac_all = _timed(device, lambda: sol.factor(next(it)), 5)
        L = bufs[-1]
        row = {"seed": seed, "order": order, "data_MB": round(sol.dataSize() * 8 / 1e6, 1),
               "factor_GF": round(flops / 1e9, 3), "analysis_s": round(analysis_s, 3),
               "factor_first_s": round(first, 6), "factor_s": round(fac, 6),
               "factor_GFs": round(flops / fac / 1e9, 1),
               "residual_probe": residual_probe(sol, h, L, nprobe=1),
               "sparse_elim_ranges": [int(v) for v in sol.sparseEliminationRanges()]}
        for nrhs in (1, 2, 10):
            rhs = torch.from_numpy(T.random_data(nrhs * order, -1, 1, 38)).to(device)
            work = rhs.clone()
            sol.solve(L, work, order, nrhs)          # heat up (Bench.cpp:165-167)

            def one():
                work.copy_(rhs)
                sol.solve(L, work, order, nrhs)
            row["solve-%d_s" % nrhs] = round(_timed(device, one, 3)[0], 6)
        del bufs, it
        for bs in ((4, 8, 16) if with_batches else ()):
            if sol.dataSize() * 8 * bs * 3 > 60e9:
                continue
            hosts = []
            for q in range(bs):
                hq = T.random_data(sol.dataSize(), -1.0, 1.0, 37 + q)
                sol.damp(hq, 0.0, order * 1.3)
                hosts.append(torch.from_numpy(hq).to(device))
            sets = [[a.clone() for a in hosts] for _ in range(3)]
            sol.factor(sets[0])
            it2 = iter(sets[1:])
            t, _ = _timed(device, lambda: sol.factor(next(it2)), 2)
            row["factor_batch%d_s_per_matrix" % bs] = round(t / bs, 6)
            if bs == 16:
                row["residual_probe_batch16_last"] = residual_probe(
                    sol, hosts[-1].cpu().numpy(), sets[-1][-1], nprobe=1)
            del hosts, sets, it2
            torch.cuda.empty_cache()
        del sol, A, L
        torch.cuda.empty_cache()
        return row

    for name, make in ref_suite_problems().items():
        if not rx.search(name):
            continue
        # the reference's five instances per family: seed 37 + it * 1000000 (Bench.cpp:458-461);
        # batches on the first instance only (they take most of the suite's time)
        inst = [one_instance(name, make, 37 + it * 1000000, it == 0) for it in range(iters)]
        row = {"problem": name, "instances": len(inst)}
        for k in inst[0]:
            vals = [r[k] for r in inst if k in r]
            if k in ("seed", "sparse_elim_ranges"):
                row[k + "s"] = vals if k == "seed" else vals[:1]
            elif k == "residual_probe" or k.startswith("residual_probe"):
                row[k] = max(vals)
            elif isinstance(vals[0], (int, float)):
                row[k] = round(statistics.median(vals), 6)
        row["factor_first_s_all"] = [r["factor_first_s"] for r in inst]
        row["factor_s_all"] = [r["factor_s"] for r in inst]
        row["first_over_warm"] = round(row["factor_first_s"] / row["factor_s"], 2)
        p = pub.get("problems", {}).get(name, {})
        row["published"] = {op: {k: v["median_s"] for k, v in d.items()} for op, d in p.items()}
        rows.append(row)
        print(json.dumps(row), flush=True)
    doc = {"suite": "ref (benchmarking/Bench.cpp:290-367)", "dtype": "f64", "device": torch.cuda.get_device_name(device),
           "published_hardware": pub.get("_hardware"), "published_source": pub.get("_source"),
           "protocol": "%d instances per family (seeds 37 + it * 1000000, Bench.cpp:458-461), medians over the "
                       "instances; per instance: factor_first_s = the first call (what the reference times), "
                       "factor_s = median of 5 warm calls, solves after a heat-up; batches per matrix on the "
                       "first instance; residual probe (worst instance) on every factor" % iters,
           "rows": rows}
    if args.suite_out:
        with open(args.suite_out, "w") as f:
            json.dump(doc, f, indent=1)
    worst = max((r["residual_probe"] for r in rows), default=0.0)
    print(json.dumps({"suite": "ref", "problems": len(rows), "worst_residual_probe": worst}))
    return 0


def spawn_ranks(n):
    """launcher mode of `python bench.py --gpus N` (no torchrun): N child processes, one per GPU,
    rendezvous on 127.0.0.1; rank 0's JSON line goes to this process's stdout; returns the exit code"""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("BSP_BENCH_SHARE_GPU") != "1":
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (n, have), file=sys.stderr)
        return 2
    # The rendezvous port is picked by binding port 0 and closing the socket (rank 0's store has to
    # bind it itself), so another process can take it in between: a launch whose ranks die within
    # the rendezvous window is retried on a fresh port.
    import time
    rc = 0
    for attempt in range(3):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        t0 = time.time()
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        rc = 0
        try:
            for p in procs:
                rc = p.wait() or rc
        finally:
            for p in procs:          # a rank that died leaves the others in a collective: end them
                if p.poll() is None:
                    p.kill()
        if rc == 0 or time.time() - t0 > 60:   # (a late failure is not a rendezvous problem)
            break
        print("bench.py: ranks failed within %.0f s (rc %d), retrying on a new port" % (time.time() - t0, rc),
              file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None,
                    help="default: bal871 at --gpus 1, grid82 (batch 64, sharded) at --gpus N > 1")
    ap.add_argument("--batch", type=int, default=None,
                    help="total number of identical-structure matrices (sharded over the GPUs); "
                         "0 = one matrix per GPU, each its own factor() (replicas)")
    ap.add_argument("--bal-file", default=None, help="BAL text file for the bal871 workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--suite", default=None, choices=["ref"],
                    help="ref: the reference's benchmark families (Bench.cpp:290-409): factor, solve-1, "
                         "solve-10 per problem, JSON lines to stdout, one summary line at the end")
    ap.add_argument("--suite-filter", default="", help="regex on the problem names of --suite")
    ap.add_argument("--suite-iters", type=int, default=5,
                    help="instances per family of --suite ref (the reference's numIterations: 5)")
    ap.add_argument("--suite-out", default=None, help="write the suite's JSON document here too")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_child:
        return cpu_baseline_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU,
        # the same environment contract as torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    # BSP_BENCH_SHARE_GPU=1 (TESTING the N > 1 control flow on a one-GPU box): every rank on GPU 0,
    # gloo instead of RCCL -- the timings mean nothing, the line's structure and the collectives'
    # order are what such a run checks
    share_gpu = os.environ.get("BSP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (under torchrun use --nproc-per-node == "
                         "--gpus; without torchrun bench.py spawns its own ranks)" % (args.gpus, world))
    ctx = {"rank": rank, "world": world, "device": device, "dist": dist}
    if args.suite == "ref":
        return suite_ref(args, device)

    # the headline line is the SAME metric at every N: BAL-871, one matrix per GPU (a single
    # factorisation is not sharded: replicas, weak scaling); the batched config rides along in
    # "batched" at every N (64 matrices sharded over the N GPUs: strong scaling)
    workload = args.workload or "bal871"
    batch = 0 if args.batch is None else args.batch
    batched = batch > 0
    total_batch = batch if batched else world

    main_run = Runner(ctx, workload, total_batch, batched, args.bal_file)
    res = main_run.run(args.steps, args.warmup)
    sol = main_run.sol

    metric = {"bal871": "factor_gflops_fp64_bal871_schur"}.get(workload, "factor_gflops_fp64_" + workload)
    if batched:
        metric = "batched_" + metric
    out = {
        "metric": metric, "value": res["value"], "unit": "GF/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if batched else "weak", "vs_baseline": None,
        "dtype": "f64", "data": main_run.source if rank == 0 else "synthetic",
        "config": {"workload": workload, "description": main_run.desc, "order": main_run.order,
                   "data_MB": round(sol.dataSize() * 8 / 1e6, 1),
                   "matrices_per_gpu": main_run.q1 - main_run.q0, "batch": total_batch,
                   "matrices_per_s": res["matrices_per_s"],
                   "factor_GF_per_matrix": round(main_run.flops / 1e9, 3),
                   "numeric_data": "mock: uniform(-1,1) seed 37(+q), damp(0, %.1f*order), as the "
                                   "reference's benches" % (1.3 if batched else 1.2),
                   "parallelism": ("batch of %d sharded over %d GPU(s), plan broadcast over RCCL"
                                   % (total_batch, world)) if batched else
                                  ("one matrix per GPU x%d (replicas; a single factorisation is not sharded)" % world)},
    }
    if "rank_ms_per_step" in res:   # N > 1: every rank's own time + what the plan broadcast cost
        out["rank_ms_per_step"] = res["rank_ms_per_step"]
        out["scaling_note"] = ("`metric` / `value` = one BAL-871 factorisation per GPU (replicas, weak scaling: no "
                               "exchange step exists inside a factorisation, x N by construction); the metric's "
                               "batched half -- 64 matrices sharded over the GPUs, STRONG scaling -- is at the top "
                               "level as `batched_metric` / `batched_value` / `batched_strong_efficiency`, its "
                               "1-GPU point measured in this job is `batched_1gpu`")

    if rank == 0:
        # ---- parity at full size: vector residual probe of the last timed factor ---------------
        out["residual_probe"] = residual_probe(sol, main_run.hosts[0], main_run.last)
        out["residual_probe_kind"] = "vector probe ||L(L^T x) - A x|| / ||A x||, 2 random x (not the Frobenius norm)"
        st = sol.planStats()
        out["plan"] = {k: st[k] for k in ("num_launches", "num_levels", "num_panels",
                                          "num_upd_tasks", "num_atomic_upd_tasks", "num_tail_panels")}
        out["analysis_s"] = round(main_run.t_sym, 3)
        if not args.no_profile:
            out.update(roofline_block(sol, main_run.A_devs[0], main_run.flops, res["ms_per_step"]
                                      if not batched else res["ms_per_step"] / max(1, main_run.q1 - main_run.q0),
                                      workload))
        # ---- device solve() on the last factor (not part of the metric)
        try:
            order = main_run.order
            rhs = torch.from_numpy(T.random_data(order, -1, 1, 38)).to(device)
            work = rhs.clone()
            sol.solve(main_run.last, work, order, 1)  # warm-up
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            work.copy_(rhs)
            sol.solve(main_run.last, work, order, 1)
            torch.cuda.synchronize(device)
            out["solve1_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
            # ten right-hand sides in one call (column-major, leading dimension = order), as the
            # reference's bench times solve-1 and solve-10 (Bench.cpp)
            rhs10 = torch.from_numpy(T.random_data(order * 10, -1, 1, 39)).to(device)
            work10 = rhs10.clone()
            sol.solve(main_run.last, work10, order, 10)  # warm-up
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            work10.copy_(rhs10)
            sol.solve(main_run.last, work10, order, 10)
            torch.cuda.synchronize(device)
            out["solve10_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
        except Exception as e:
            out["solve1_ms"] = "error: %s" % e

    # ---- extra workloads (every rank takes part in the collective ones) ------------------------
    if not args.no_extras and args.workload is None and args.batch is None and not args.bal_file:
        # the headline workload's device buffers go first, and torch's cached blocks with them: a
        # batch carved out of the cached 1.1 GB blocks ran 20 % slower than one in fresh allocations
        # (tools/exp_order.py)
        main_run.A_devs = main_run.last = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        try:
            if world == 1:
                r = Runner(ctx, "grid82", 64, True)
                e = r.run(3, 1)
                e.update({"workload": "64 x GRID 82x82 conn 2 x 3 (BASELINE configs[3]), one batched "
                                      "factor() per step", "n_gpus": 1, "scaling": "strong"})
                out["batched"] = e
                del r
                torch.cuda.empty_cache()
                # one GPU's SHARE of the 8-GPU run of that config: a batch of 8.  What one GPU says about
                # the strong-scaling curve before any multi-GPU run: efficiency at 8 GPUs = t(64 on 1) /
                # (8 x t(8 on 1)) -- the plan broadcast is paid once, outside the timed steps
                r8 = Runner(ctx, "grid82", 8, True)
                e8 = r8.run(5, 2)
                e8.update({"workload": "8 x GRID 82x82: one rank's share of the 64-batch on 8 GPUs", "n_gpus": 1})
                out["batched_share_of_8"] = e8
                out["predicted_8gpu_strong_efficiency"] = round(e["ms_per_step"] / (8.0 * e8["ms_per_step"]), 3)
                del r8
                torch.cuda.empty_cache()
                out["c5"] = c5_block(device)
                # vendor comparators on the same GPU (tools/vendor_compare.py -- tools only, never
                # the product): rocSOLVER's dense Cholesky of a matrix as wide as the camera block,
                # whose factorisation is this library's dense phase, and rocBLAS's rank-256 syrk
                try:
                    sys.path.insert(0, os.path.join(HERE, "tools"))
                    import vendor_compare
                    cmpr = vendor_compare.run(n_list=(7839,), syrk=((7839, 256),), reps=3)
                    km = out.get("kernel_ms", {})
                    if km:
                        elim = km.get("elim_factor", [0])[0] + km.get("elim_update", [0])[0]
                        cmpr["this_library_dense_phase_ms"] = round(out["ms_per_step"] - elim, 3)
                        cmpr["note"] = ("dense phase = ms_per_step minus the two sparse-elimination kernels: "
                                        "the Cholesky of the 7839-wide camera block incl. its scatter into "
                                        "skeleton layout; rocsolver_dpotrf factors a plain dense 7839^2 matrix")
                    out["comparators"] = cmpr
                except Exception as e:  # noqa: BLE001
                    out["comparators"] = {"error": repr(e)[:200]}
            else:
                r = Runner(ctx, "grid82", 64, True)   # (collective: plan broadcast, every rank its shard)
                e = r.run(3, 1)
                e.update({"workload": "64 x GRID 82x82 conn 2 x 3 (BASELINE configs[3]) sharded over %d GPUs, "
                                      "plan broadcast over RCCL, one batched factor() per rank and step" % world,
                          "n_gpus": world, "scaling": "strong"})
                out["batched"] = e
                del r
                torch.cuda.empty_cache()
                if rank == 0:   # the 1-GPU point of the batched curve, measured in this very job
                    r1 = Runner(ctx, "grid82", 64, True, participate=False)
                    e1 = r1.run(3, 1, sync_ranks=False)
                    e1.update({"workload": "all 64 matrices on rank 0 alone", "n_gpus": 1})
                    out["batched_1gpu"] = e1
                # (no barrier here: the other ranks wait at the final one; a barrier inside this
                #  try block would be skipped by a rank whose rider raised)
        except Exception as e:  # extras never fail the bench
            out["extras_error"] = repr(e)

    if rank == 0 and isinstance(out.get("batched"), dict) and "value" in out["batched"]:
        # BASELINE's metric has two halves -- "factor() GF/s on BAL-871" and "batch throughput at 1/2/4/8
        # GPUs".  `metric` / `value` stay the first at every N (replicas at N > 1: a single factorisation
        # is not sharded), so that the N = 1 line is the same line everywhere; the second half is at the TOP
        # level too, under its own names, and it is the STRONG-scaling one: 64 matrices sharded over the N GPUs
        b = out["batched"]
        out["batched_metric"] = "batched_factor_gflops_fp64_grid82x64"
        out["batched_value"] = b["value"]
        out["batched_unit"] = "GF/s"
        out["batched_ms_per_step"] = b["ms_per_step"]
        out["batched_scaling"] = "strong"
        if world > 1 and isinstance(out.get("batched_1gpu"), dict) and "ms_per_step" in out["batched_1gpu"]:
            sp = out["batched_1gpu"]["ms_per_step"] / b["ms_per_step"]
            out["batched_speedup_vs_1gpu"] = round(sp, 3)
            out["batched_strong_efficiency"] = round(sp / world, 3)
    if rank == 0:
        # ---- CPU baseline: the BackendFast restatement (oracle/blas_factor.c) on host cores ----
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sol, main_run.hosts[0], main_run.flops,
                                                   out["value"] / total_batch,
                                                   workload if not batched else None, args.bal_file)
            except Exception as e:  # the baseline is a report, never a reason to fail the bench
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))

    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
