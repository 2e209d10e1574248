"""dense single-lump factor of width W (params of size 8, full structure): device factor against numpy"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import baspacho_amd as B
from baspacho_amd import testing as T
from helpers import dense_lower_chol, lower_of, spd_data, to_dev

for W in [int(a) for a in sys.argv[1:]]:
    nparam = W // 8
    sizes = np.full(nparam, 8, dtype=np.int64)
    cols = [list(range(c, nparam)) for c in range(nparam)]
    ss = T.columns_to_structure(cols)
    sol = B.create_solver(B.Settings(), sizes, ss, [])
    d = spd_data(sol, 3, dtype=np.float64)
    dev = to_dev(d)
    sol.factor(dev)
    L, A = dense_lower_chol(sol, d)
    got = lower_of(sol, dev.cpu().numpy())
    E = np.abs(got - L)
    idx = np.argwhere(E > 1e-9 * np.abs(L).max())
    lumps = np.diff(np.asarray(sol.skel()["lumpStart"]))
    print("W %5d lumps %s rel err %.2e  off entries %d %s" % (W, lumps[-3:], np.linalg.norm(got - L) / np.linalg.norm(L), len(idx),
          ("rows %d..%d cols %d..%d" % (idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max())) if len(idx) else ""))

    if len(idx) and W % 256 == 64:
        r0 = W - 64
        Gl, Ll = got[r0:, r0:], L[r0:, r0:]
        M = Gl @ Gl.T - Ll @ Ll.T   # error of the Schur complement tile before its potrf (got - exact)
        S = [L[r0:, 64 * p:64 * p + 64] @ L[r0:, 64 * p:64 * p + 64].T for p in range(r0 // 64)]
        Amat = np.stack([s.ravel() for s in S], axis=1)
        c, res, *_ = np.linalg.lstsq(Amat, M.ravel(), rcond=None)
        print("   |M| %.3e; least squares over panel terms X_p X_p^T: coefficients" % np.abs(M).max(), np.round(c, 4),
              "residual %.2e" % (np.linalg.norm(Amat @ c - M.ravel()) / np.linalg.norm(M.ravel())))
        # 16-column pieces of the last block's own predecessors
        S16 = [L[r0:, 16 * q:16 * q + 16] @ L[r0:, 16 * q:16 * q + 16].T for q in range(r0 // 16)]
        A16 = np.stack([s.ravel() for s in S16], axis=1)
        c16, *_ = np.linalg.lstsq(A16, M.ravel(), rcond=None)
        nz = [(q, round(float(v), 4)) for q, v in enumerate(c16) if abs(v) > 0.02]
        print("   16-column pieces with |coefficient| > 0.02:", nz, "residual %.2e" % (np.linalg.norm(A16 @ c16 - M.ravel()) / np.linalg.norm(M.ravel())))
