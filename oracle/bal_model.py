"""TEST INFRASTRUCTURE (oracle): host restatement of the BAL caller pipeline.

* reprojection residual + Jacobians of the 9-parameter BAL camera (Rodrigues rotation, translation,
  f, k1, k2 -- the file format of benchmarking/BaAtLarge.cpp:140-150) with forward-mode dual
  numbers in numpy, vectorised over the observations;
* Hessian / gradient assembly exactly as computeStep does it (benchmarking/BaAtLargeOptimizer.cpp:
  100-131): per observation  accessor.diagBlock(pt) += Jp^T Jp,  accessor.diagBlock(cam) += Jc^T Jc,
  accessor.block(cam, pt) += Jc^T Jp,  grad += J^T err,  then  diag *= 1 + lambda; diag += 1e-3
  lambda -- through the HOST accessor (Solver::accessor(), Accessor.h:145-166).
Only tests/ and the checker legs of bench.py may import this file."""
import numpy as np

ND = 12


class Dual:
    """value (n,) + 12 partials (n, 12): camera parameters 0..8, point 9..11"""

    def __init__(self, v, d):
        self.v, self.d = v, d

    @staticmethod
    def const(v):
        v = np.asarray(v, dtype=np.float64)
        return Dual(v, np.zeros(v.shape + (ND,)))

    @staticmethod
    def var(v, idx):
        r = Dual.const(v)
        r.d[..., idx] = 1.0
        return r

    def __add__(self, o):
        return Dual(self.v + o.v, self.d + o.d)

    def __sub__(self, o):
        return Dual(self.v - o.v, self.d - o.d)

    def __mul__(self, o):
        return Dual(self.v * o.v, self.d * o.v[..., None] + self.v[..., None] * o.d)

    def __truediv__(self, o):
        inv = 1.0 / o.v
        v = self.v * inv
        return Dual(v, (self.d - v[..., None] * o.d) * inv[..., None])

    def fn(self, s, ds):
        return Dual(s, ds[..., None] * self.d)


def linearize(cams, pts, xy):
    """cams (n, 9), pts (n, 3), xy (n, 2) per observation -> res (n, 2), Jc (n, 2, 9), Jp (n, 2, 3)"""
    n = len(cams)
    w = [Dual.var(cams[:, i], i) for i in range(3)]
    t = [Dual.var(cams[:, 3 + i], 3 + i) for i in range(3)]
    X = [Dual.var(pts[:, i], 9 + i) for i in range(3)]
    f, k1, k2 = Dual.var(cams[:, 6], 6), Dual.var(cams[:, 7], 7), Dual.var(cams[:, 8], 8)
    th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
    assert np.all(th2.v > 1e-20), "oracle: zero rotations are not part of the test scenes"
    th = np.sqrt(th2.v)
    theta = th2.fn(th, 0.5 / th)
    c, s = theta.fn(np.cos(th), -np.sin(th)), theta.fn(np.sin(th), np.cos(th))
    k = [wi / theta for wi in w]
    kx = [k[1] * X[2] - k[2] * X[1], k[2] * X[0] - k[0] * X[2], k[0] * X[1] - k[1] * X[0]]
    one = Dual.const(np.ones(n))
    kdot = (k[0] * X[0] + k[1] * X[1] + k[2] * X[2]) * (one - c)
    P = [X[i] * c + kx[i] * s + k[i] * kdot + t[i] for i in range(3)]
    zero = Dual.const(np.zeros(n))
    px, py = (zero - P[0]) / P[2], (zero - P[1]) / P[2]
    r2 = px * px + py * py
    dist = one + r2 * (k1 + k2 * r2)
    r0 = f * dist * px - Dual.const(xy[:, 0])
    r1 = f * dist * py - Dual.const(xy[:, 1])
    res = np.stack([r0.v, r1.v], axis=1)
    J = np.stack([r0.d, r1.d], axis=1)     # (n, 2, 12)
    return res, J[:, :, :9].copy(), J[:, :, 9:].copy()


def fill_hessian_host(sol, prob, Jc, Jp, res, lam):
    """computeStep's assembly through the host accessor; returns (data, grad, offsets) where
    offsets (n, 7) = what the accessor answered per observation (block offset, stride, flipped,
    camera diag offset, stride, point diag offset, stride)"""
    npt = prob.num_pts
    data = np.zeros(sol.dataSize())
    grad = np.zeros(sol.order())
    perm = sol.paramToSpan()
    span_start = sol.skel()["spanStart"]
    offs = np.zeros((len(prob.obs_cam), 7), dtype=np.int64)

    def add_block(off, stride, M, flipped=False):
        r, c = M.shape
        for i in range(r):
            for j in range(c):
                data[off + (j * stride + i if flipped else i * stride + j)] += M[i, j]

    for o, (c, p) in enumerate(zip(prob.obs_cam, prob.obs_pt)):
        cam_id, pt_id = npt + int(c), int(p)
        off, stride, flipped = sol.blockOffset(cam_id, pt_id)
        dco, dcs = sol.diagBlockOffset(cam_id)
        dpo, dps = sol.diagBlockOffset(pt_id)
        offs[o] = (off, stride, 1 if flipped else 0, dco, dcs, dpo, dps)
        add_block(dpo, dps, Jp[o].T @ Jp[o])
        add_block(dco, dcs, Jc[o].T @ Jc[o])
        add_block(off, stride, Jc[o].T @ Jp[o], flipped)
        ps, cs = int(span_start[perm[pt_id]]), int(span_start[perm[cam_id]])
        grad[ps:ps + 3] += Jp[o].T @ res[o]
        grad[cs:cs + 9] += Jc[o].T @ res[o]
    if lam != 0.0:
        sizes = np.concatenate([np.full(npt, 3), np.full(prob.num_cams, 9)])
        for i, n in enumerate(sizes):
            off, stride = sol.diagBlockOffset(i)
            for e in range(n):
                q = off + e * (stride + 1)
                data[q] = data[q] * (1.0 + lam) + lam * 1e-3
    return data, grad, offs
