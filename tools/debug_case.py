"""where a stress case's device factor differs from the dense one: python tools/debug_case.py SEED [big]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import dense_lower_chol, lower_of, spd_data, to_dev

seed = int(sys.argv[1])
big = len(sys.argv) > 2
rng = np.random.default_rng(seed)
size = int(rng.integers(300, 900)) if big else int(rng.integers(20, 260))
fill = float(rng.choice([0.05, 0.15, 0.5])) if big else float(rng.choice([0.01, 0.03, 0.08, 0.3]))
pmax = int(rng.choice([3, 6, 9])) if big else int(rng.choice([1, 3, 5, 9, 23]))
sizes = rng.integers(1, pmax + 1, size=size).astype(np.int64)
cols = T.random_cols(size, fill, 100 + seed)
ranges = []
if rng.random() < 0.5:
    k = int(rng.integers(5, max(6, size // 2)))
    cols = T.make_independent_elim_set(cols, 0, k)
    if rng.random() < 0.5:
        ranges = [0, k]
ss = T.columns_to_structure(cols)
st = B.Settings(findSparseEliminationRanges=bool(rng.random() < 0.7))
dtype = np.float64 if rng.random() < 0.6 else np.float32
sol = B.create_solver(st, sizes, ss, ranges)
n = sol.order()
nb = int(rng.choice([1, 1, 2, 5]))
print("order", n, "batch", nb, "dtype", dtype.__name__, "ranges", ranges, "elim ranges", sol.sparseEliminationRanges())
sk = sol.skel()
ls = np.asarray(sk["lumpStart"])
print("lumps", len(ls) - 1, "widths (last 8):", np.diff(ls)[-8:], "starts (last 8):", ls[-9:])
datas = [spd_data(sol, 7 * seed + q, dtype=dtype) for q in range(nb)]
devs = [to_dev(d) for d in datas]
sol.factor(devs if nb > 1 else devs[0])
L, A = dense_lower_chol(sol, datas[0])
got = lower_of(sol, devs[0].cpu().numpy())
E = np.abs(got - L)
print("rel err", np.linalg.norm(got - L) / np.linalg.norm(L), "max abs", E.max())
idx = np.argwhere(E > 1e-9 * np.abs(L).max())
print("entries off:", len(idx))
if len(idx):
    r, c = idx[:, 0], idx[:, 1]
    print("rows", r.min(), r.max(), "cols", c.min(), c.max())
    for lo in range(int(c.min()) // 16 * 16, int(c.max()) + 1, 16):
        m = (c >= lo) & (c < lo + 16)
        if m.any():
            print("  cols %5d..%5d: %6d entries, rows %d..%d, max err %.2e" % (lo, lo + 15, m.sum(), r[m].min(), r[m].max(), E[r[m], c[m]].max()))
