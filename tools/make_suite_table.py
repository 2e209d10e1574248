"""README table of `python bench.py --suite ref` (profiles/r05_ref_suite.json: medians over five instances per family) beside the reference's
published timings (profiles/published_reference_results.json).  usage: python tools/make_suite_table.py [suite.json]"""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "profiles", "r05_ref_suite.json")
d = json.load(open(path))
ms = lambda s: "%.2f" % (1e3 * s) if s is not None else "-"
print("| problem (Bench.cpp:290-367) | order | GF | first / warm factor ms | TF/s | batch-16 ms/matrix | solve-1 / solve-10 ms | "
      "published factor ms: CUDA / CUDA batch-16 / BLAS 16 thr / CHOLMOD | published solve-1 / solve-10 ms (CUDA) | probe |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in d["rows"]:
    p = r.get("published", {})
    pf, p1, p10 = p.get("factor", {}), p.get("solve-1", {}), p.get("solve-10", {})
    name = r["problem"].split("_", 1)
    print("| %s | %d | %.1f | %s / %s | %.1f | %s | %s / %s | %s / %s / %s / %s | %s / %s | %.0e |" % (
        r["problem"][:44], r["order"], r["factor_GF"], ms(r.get("factor_first_s")), ms(r["factor_s"]), r["factor_GFs"] / 1e3,
        ms(r.get("factor_batch16_s_per_matrix")), ms(r["solve-1_s"]), ms(r["solve-10_s"]),
        ms(pf.get("3_BaSpaCho_CUDA")), ms(pf.get("6_BaSpaCho_CUDA_batchsize=16")),
        ms(pf.get("2_BaSpaCho_BLAS_numthreads=16")), ms(pf.get("1_CHOLMOD")),
        ms(p1.get("3_BaSpaCho_CUDA")), ms(p10.get("3_BaSpaCho_CUDA")), r["residual_probe"]))
print()
print("ours: %s; published: %s" % (d.get("device"), d.get("published_hardware")))
