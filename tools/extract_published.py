"""Transcribe the reference's published timings (BENCHMARK_RESULTS.md, OpenBLAS / CUDA 11.5 sections:
factor at :31-208, solve-1/2/10 at :303-767) into profiles/published_reference_results.json: per
problem type, operation and solver the five printed times in seconds and their median.  Numbers
only -- DATA, not source.  Runs in the build container (the reference tree is not on the GPU box);
bench.py --suite ref prints these beside its own measurements.
Hardware of the published numbers (BENCHMARK_RESULTS.md:10-13): ThinkStation P720, Xeon Silver 4214
@ 2.2 GHz, Quadro RTX 5000."""
import json
import os
import re
import statistics
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/BENCHMARK_RESULTS.md"
lines = open(src).read().splitlines()
out, section, prob, op, solver = {}, None, None, None, None


def secs(tok):
    m = re.match(r"([0-9.]+)(ms|s)$", tok)
    return float(m.group(1)) * (1e-3 if m.group(2) == "ms" else 1.0)


for ln in lines:
    if ln.startswith("## "):
        section = ln[3:].strip()
        continue
    if section is None or "Intel-MKL" in section or "Analysis" in section:
        continue
    m = re.match(r"Problem type: (\S+)", ln)
    if m:
        prob = m.group(1)
        continue
    m = re.match(r"Operation: (\S+)", ln)
    if m:
        op = m.group(1)
        continue
    m = re.match(r"- (\S+) \(", ln)
    if m:
        solver = m.group(1)
        continue
    if prob and op and solver and ln.startswith("    "):
        vals = [secs(t) for t in re.findall(r"([0-9.]+m?s)\b", ln)]
        if vals:
            out.setdefault(prob, {}).setdefault(op, {})[solver] = {
                "seconds": vals, "median_s": statistics.median(vals)}
        solver = None
doc = {"_source": "facebookresearch/baspacho BENCHMARK_RESULTS.md, sections 'Factor (OpenBLAS/Cuda 11.5)' and "
                  "'Solve (OpenBLAS/Cuda 11.5, nRHS = 1, 2, 10)'",
       "_hardware": "ThinkStation P720: Intel Xeon Silver 4214 @ 2.20 GHz, 128 GB, Quadro RTX 5000 (CUDA 11.5)",
       "_note": "batched solvers report seconds PER MATRIX (Bench.cpp:238,262); five random problems per type",
       "problems": out}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "published_reference_results.json")
json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
print(len(out), "problem types;", {k: sorted(v) for k, v in list(out.items())[:2]})
