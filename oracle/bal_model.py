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


# ---- the reference's parameterisation (BaAtLarge.h:56-150): SE3 tangent + fixed calibration ----
def _rodrigues(w):
    """rotation matrices (n, 3, 3) of axis-angle vectors (n, 3)"""
    th = np.linalg.norm(w, axis=1)
    K = np.zeros((len(w), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    small = th < 1e-10
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(ths) / ths)
    b = np.where(small, 0.5, (1 - np.cos(ths)) / ths ** 2)
    return np.eye(3)[None] + a[:, None, None] * K + b[:, None, None] * (K @ K)


def residual_se3(R, t, calib, X, xy):
    """Cost::compute_residual (BaAtLarge.h:56-72) on arrays: R (n,3,3), t (n,3), calib (n,3)"""
    P = np.einsum("nij,nj->ni", R, X) + t
    bad = P[:, 2] > 0.01
    z = np.where(bad, -1.0, P[:, 2])
    p = -P[:, :2] / z[:, None]
    sq = np.sum(p * p, axis=1)
    r = 1.0 + (calib[:, 1] + calib[:, 2] * sq) * sq
    res = (calib[:, 0] * r)[:, None] * p - xy
    res[bad] = (25.0, 0.0)
    return res


def linearize_se3(cams, pts, xy):
    """Cost::compute_residual with Jacobians (BaAtLarge.h:74-147), formula for formula: per
    observation cams (n, 9) [Rodrigues r, t, f, k1, k2], pts (n, 3), xy (n, 2) ->
    res (n, 2), Jc (n, 2, 6) [translation, rotation of a left perturbation], Jp (n, 2, 3)"""
    R = _rodrigues(cams[:, 0:3])
    P = np.einsum("nij,nj->ni", R, pts) + cams[:, 3:6]
    f, k1, k2 = cams[:, 6], cams[:, 7], cams[:, 8]
    bad = P[:, 2] > 0.01
    z = np.where(bad, -1.0, P[:, 2])
    x, y = P[:, 0], P[:, 1]
    p = np.stack([-x / z, -y / z], axis=1)
    sq = np.sum(p * p, axis=1)
    r = 1.0 + (k1 + k2 * sq) * sq
    res = (f * r)[:, None] * p - xy
    g = f * (k1 + k2 * 2.0 * sq)
    denum = -1.0 / (z * z)
    Dp = np.stack([(z[:, None] * R[:, 0, :] - x[:, None] * R[:, 2, :]) * denum[:, None],
                   (z[:, None] * R[:, 1, :] - y[:, None] * R[:, 2, :]) * denum[:, None]], axis=1)   # (n,2,3)
    dz = 1.0 / z
    xdz, ydz = x * dz, y * dz
    zero = np.zeros_like(z)
    Dc = np.stack([np.stack([-dz, zero, xdz * dz, xdz * ydz, -1 - xdz * xdz, ydz], axis=1),
                   np.stack([zero, -dz, ydz * dz, 1 + ydz * ydz, -xdz * ydz, -xdz], axis=1)], axis=1)  # (n,2,6)

    def full(D):
        dsq = 2.0 * np.einsum("ni,nic->nc", p, D)
        return (f * r)[:, None, None] * D + p[:, :, None] * dsq[:, None, :] * g[:, None, None]
    Jp, Jc = full(Dp), full(Dc)
    res[bad] = (25.0, 0.0)
    Jp[bad] = 0.0
    Jc[bad] = 0.0
    return res, Jc, Jp


def fill_hessian_host_se3(sol, prob, Jc, Jp, res, lam):
    """computeStep's assembly (BaAtLargeOptimizer.cpp:100-131) with the optimizer's 6-wide camera
    blocks, through the host accessor; returns (data, grad)"""
    npt = prob.num_pts
    data = np.zeros(sol.dataSize())
    grad = np.zeros(sol.order())
    perm = sol.paramToSpan()
    span_start = sol.skel()["spanStart"]

    def add_block(off, stride, M, flipped=False):
        r, c = M.shape
        idx = (np.arange(c)[None, :] * stride + np.arange(r)[:, None]) if flipped else \
              (np.arange(r)[:, None] * stride + np.arange(c)[None, :])
        data[off + idx] += M

    for o, (c, p) in enumerate(zip(prob.obs_cam, prob.obs_pt)):
        cam_id, pt_id = npt + int(c), int(p)
        off, stride, flipped = sol.blockOffset(cam_id, pt_id)
        dco, dcs = sol.diagBlockOffset(cam_id)
        dpo, dps = sol.diagBlockOffset(pt_id)
        add_block(dpo, dps, Jp[o].T @ Jp[o])
        add_block(dco, dcs, Jc[o].T @ Jc[o])
        add_block(off, stride, Jc[o].T @ Jp[o], flipped)
        ps, cs = int(span_start[perm[pt_id]]), int(span_start[perm[cam_id]])
        grad[ps:ps + 3] += Jp[o].T @ res[o]
        grad[cs:cs + 6] += Jc[o].T @ res[o]
    if lam != 0.0:
        sizes = np.concatenate([np.full(npt, 3), np.full(prob.num_cams, 6)])
        for i, n in enumerate(sizes):
            off, stride = sol.diagBlockOffset(i)
            q = off + np.arange(n) * (stride + 1)
            data[q] = data[q] * (1.0 + lam) + lam * 1e-3
    return data, grad
