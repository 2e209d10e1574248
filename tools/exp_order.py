"""Experiment: does a BAL-871 run leave the process / GPU in a state that slows the next workload?"""
import gc
import sys
import time

import torch

sys.path.insert(0, ".")
import bench

ctx = {"rank": 0, "world": 1, "device": torch.device("cuda", 0), "dist": None}
mode = sys.argv[1]
if mode in ("A", "B", "C"):
    m = bench.Runner(ctx, "bal871", 1, False)
    print("bal871", m.run(5, 2)["ms_per_step"])
    if mode == "B":
        del m
        gc.collect()
        torch.cuda.empty_cache()
    if mode == "C":
        torch.cuda.synchronize()
        time.sleep(3.0)
r = bench.Runner(ctx, "grid82", 64, True)
print(mode, "grid82x64", r.run(3, 1)["ms_per_step"], r.run(3, 1)["ms_per_step"])
