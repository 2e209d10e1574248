"""Bundle-adjustment-at-large caller pipeline: BAL text files, the solver structure the reference's
BAL_bench builds from them, and the device-side linearisation / Hessian assembly.

Mirrors benchmarking/BaAtLarge.cpp:81-182 (Data::load: header `numCams numPts numObs`, one line
`cam pt x y` per observation, 9 numbers per camera -- Rodrigues rotation, translation, f, k1, k2 --
3 per point; same error behaviour: a bad index or a short file raises), :184-230 (save),
BaAtLargeBench.cpp:44-73 (structure: points first, size 3, cameras after, size 9, one block per
observation, sparse elimination range {0, numPts}) and BaAtLargeOptimizer.cpp:100-131 (computeStep:
Hessian / gradient assembly through the accessor + LM damping), the last on the GPU through
Solver::deviceAccessor() (csrc/bal_pipeline.hip)."""
import bz2
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _lib
from .testing import structure_from_pairs


@dataclass
class BalProblem:
    cams: np.ndarray      # (numCams, 9): r(3), t(3), f, k1, k2
    pts: np.ndarray       # (numPts, 3)
    obs_cam: np.ndarray   # (numObs,) int64
    obs_pt: np.ndarray    # (numObs,) int64
    obs_xy: np.ndarray    # (numObs, 2)

    @property
    def num_cams(self):
        return len(self.cams)

    @property
    def num_pts(self):
        return len(self.pts)


def load_bal(path):
    """Data::load (BaAtLarge.cpp:81-182); .bz2 files (as the BAL site ships them) are accepted"""
    try:
        if str(path).endswith(".bz2"):
            with bz2.open(path, "rt") as f:
                tok = np.array(f.read().split(), dtype=np.float64)
        else:
            tok = np.fromfile(path, dtype=np.float64, sep=" ")
    except OSError as e:
        raise RuntimeError("Cannot open file `%s`" % path) from e
    if len(tok) < 3:
        raise RuntimeError("Cannot open file `%s`" % path)
    nc, npt, nobs = (int(v) for v in tok[:3])
    need = 3 + 4 * nobs + 9 * nc + 3 * npt
    body = tok[3:]
    if len(tok) < 3 + 4 * nobs:
        raise RuntimeError("File error loading %dth observation!" % ((len(tok) - 3) // 4))
    obs = body[:4 * nobs].reshape(nobs, 4)
    cam_idx, pt_idx = obs[:, 0].astype(np.int64), obs[:, 1].astype(np.int64)
    bad = np.nonzero((cam_idx >= nc) | (cam_idx < 0))[0]
    if len(bad):
        raise RuntimeError("Error loading %dth observation, invalid camera index: %d"
                           % (bad[0], cam_idx[bad[0]]))
    bad = np.nonzero((pt_idx >= npt) | (pt_idx < 0))[0]
    if len(bad):
        raise RuntimeError("Error loading %dth observation, invalid point index: %d"
                           % (bad[0], pt_idx[bad[0]]))
    if len(tok) < 3 + 4 * nobs + 9 * nc:
        raise RuntimeError("File error loading %dth camera!" % ((len(tok) - 3 - 4 * nobs) // 9))
    if len(tok) < need:
        raise RuntimeError("File error loading %dth point!" % ((len(tok) - 3 - 4 * nobs - 9 * nc) // 3))
    cams = body[4 * nobs:4 * nobs + 9 * nc].reshape(nc, 9).copy()
    pts = body[4 * nobs + 9 * nc:4 * nobs + 9 * nc + 3 * npt].reshape(npt, 3).copy()
    return BalProblem(cams, pts, cam_idx, pt_idx, obs[:, 2:4].copy())


def save_bal(path, prob):
    """Data::save (BaAtLarge.cpp:184-230): the same text layout, %.16e precision"""
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (prob.num_cams, prob.num_pts, len(prob.obs_cam)))
        for c, p, (x, y) in zip(prob.obs_cam, prob.obs_pt, prob.obs_xy):
            f.write("%d %d %.16e %.16e\n" % (c, p, x, y))
        for v in prob.cams.reshape(-1):
            f.write("%.16e\n" % v)
        for v in prob.pts.reshape(-1):
            f.write("%.16e\n" % v)


def bal_structure(prob, cam_size=9):
    """(paramSizes, SparseStructure, sparseElimRanges) as testSolvers builds them
    (BaAtLargeBench.cpp:44-73): points first, then cameras; block (numPts + cam, pt) per observation.
    cam_size = 9: the camera blocks of BAL_bench (all parameters of a BAL file); cam_size = 6: the
    blocks of the LM optimizer BAL_opt (SE3 tangent, BaAtLargeOptimizer.cpp:33-52)"""
    npt, nc = prob.num_pts, prob.num_cams
    sizes = np.concatenate([np.full(npt, 3, dtype=np.int64), np.full(nc, cam_size, dtype=np.int64)])
    key = np.unique(prob.obs_pt * nc + prob.obs_cam)   # duplicate observations share a block
    ss = structure_from_pairs(npt + nc, npt + key % nc, key // nc)
    return sizes, ss, [0, npt]


def project(cams, pts):
    """the BAL camera model on arrays: p = -P.xy / P.z with P = R(r) X + t, then f (1 + k1 r^2 +
    k2 r^4) p (numpy; used to synthesise observations)"""
    w, t = cams[:, 0:3], cams[:, 3:6]
    th = np.linalg.norm(w, axis=1, keepdims=True)
    k = w / np.maximum(th, 1e-300)
    c, s = np.cos(th), np.sin(th)
    P = pts * c + np.cross(k, pts) * s + k * (np.sum(k * pts, axis=1, keepdims=True)) * (1 - c) + t
    p = -P[:, :2] / P[:, 2:3]
    r2 = np.sum(p * p, axis=1, keepdims=True)
    return cams[:, 6:7] * (1 + r2 * (cams[:, 7:8] + cams[:, 8:9] * r2)) * p


def synth_scene(num_cams=12, num_pts=200, mean_track=4, seed=3, noise=0.5, perturb=1e-2):
    """a small geometrically consistent scene: cameras on an arc looking at a point cloud, pixel
    noise on the observations, parameters perturbed off the truth (so that an LM step has work)"""
    rng = np.random.default_rng(seed)
    ang = np.linspace(-0.6, 0.6, num_cams)
    cams = np.zeros((num_cams, 9))
    cams[:, 1] = ang                                   # rotation about y
    cams[:, 0] = 0.05 * rng.standard_normal(num_cams)
    cams[:, 3] = 2.0 * np.sin(ang)
    cams[:, 5] = -12.0 + rng.uniform(-0.5, 0.5, num_cams)   # scene in front of the camera (z < 0)
    cams[:, 6] = 800.0 + rng.uniform(-20, 20, num_cams)
    cams[:, 7] = 1e-2 * rng.standard_normal(num_cams)
    cams[:, 8] = 1e-4 * rng.standard_normal(num_cams)
    pts = rng.uniform(-2.0, 2.0, (num_pts, 3))
    oc, op = [], []
    for p in range(num_pts):
        k = int(min(num_cams, max(2, rng.poisson(mean_track))))
        c0 = rng.integers(0, num_cams)
        sel = np.unique(np.clip(c0 + rng.integers(-3, 4, size=k), 0, num_cams - 1))
        if len(sel) < 2:
            sel = np.array([c0, (c0 + 1) % num_cams])
        oc.extend(sel.tolist())
        op.extend([p] * len(sel))
    oc, op = np.array(oc, dtype=np.int64), np.array(op, dtype=np.int64)
    xy = project(cams[oc], pts[op]) + noise * rng.standard_normal((len(oc), 2))
    cams = cams + perturb * rng.standard_normal(cams.shape) * np.array([1, 1, 1, 1, 1, 1, 100, 1e-2, 1e-4])
    pts = pts + perturb * rng.standard_normal(pts.shape)
    return BalProblem(cams, pts, oc, op, xy)


def synth_scene_for(num_cams, num_pts, obs_cam, obs_pt, seed=3, noise=0.5, perturb=1e-2):
    """a geometrically consistent scene for a GIVEN observation list (e.g. the co-visibility of
    testing.gen_bal_synthetic at BAL-871 / BAL-1723 size): cameras on a wide arc around a point cloud,
    every listed observation projected through the BAL camera model + pixel noise, parameters
    perturbed off the truth.  Vectorised (no per-point loop); the Hessian J^T J of the result is a
    real bundle-adjustment Hessian, not a diagonally dominant mock."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(-1.2, 1.2, num_cams)
    cams = np.zeros((num_cams, 9))
    cams[:, 1] = ang
    cams[:, 0] = 0.05 * rng.standard_normal(num_cams)
    cams[:, 3] = 2.0 * np.sin(ang)
    cams[:, 5] = -12.0 + rng.uniform(-0.5, 0.5, num_cams)
    cams[:, 6] = 800.0 + rng.uniform(-20, 20, num_cams)
    cams[:, 7] = 1e-2 * rng.standard_normal(num_cams)
    cams[:, 8] = 1e-4 * rng.standard_normal(num_cams)
    pts = rng.uniform(-2.0, 2.0, (num_pts, 3))
    oc = np.ascontiguousarray(obs_cam, dtype=np.int64)
    op = np.ascontiguousarray(obs_pt, dtype=np.int64)
    xy = project(cams[oc], pts[op]) + noise * rng.standard_normal((len(oc), 2))
    cams = cams + perturb * rng.standard_normal(cams.shape) * np.array([1, 1, 1, 1, 1, 1, 100, 1e-2, 1e-4])
    pts = pts + perturb * rng.standard_normal(pts.shape)
    return BalProblem(cams, pts, oc, op, xy)


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def so3_exp(w):
    th = float(np.linalg.norm(w))
    K = _skew(w)
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def so3_log(R):
    c = min(1.0, max(-1.0, 0.5 * (np.trace(R) - 1.0)))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    if np.pi - th < 1e-6:   # near pi: axis from the symmetric part
        A = 0.5 * (R + np.eye(3))
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[:, k] / ax[k]
        return th * ax / np.linalg.norm(ax)
    return th / (2.0 * np.sin(th)) * v


def se3_exp(delta):
    """Sophus::SE3d::exp for delta = (translation part, rotation part): (R, t)"""
    u, w = np.asarray(delta[:3], dtype=np.float64), np.asarray(delta[3:], dtype=np.float64)
    th = float(np.linalg.norm(w))
    K = _skew(w)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * K + K @ K / 6.0
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    return so3_exp(w), V @ u


def apply_step_se3(prob, step, solver):
    """applyStep of the reference's optimizer (BaAtLargeOptimizer.cpp:170-184): points -= step,
    T_W_C <- exp(-step) T_W_C; `step` is in the solver's internal order (paramToSpan /
    spanVectorOffset); cameras keep the BAL storage (Rodrigues rotation, translation).  Returns a
    new BalProblem."""
    out = BalProblem(prob.cams.copy(), prob.pts.copy(), prob.obs_cam, prob.obs_pt, prob.obs_xy)
    perm = solver.paramToSpan()
    ss = solver.skel()["spanStart"]
    npt = prob.num_pts
    pt_off = ss[perm[:npt]]
    out.pts -= step[pt_off[:, None] + np.arange(3)[None, :]]
    for i in range(prob.num_cams):
        o = int(ss[perm[npt + i]])
        Rd, td = se3_exp(-step[o:o + 6])
        R = so3_exp(prob.cams[i, 0:3])
        out.cams[i, 0:3] = so3_log(Rd @ R)
        out.cams[i, 3:6] = Rd @ prob.cams[i, 3:6] + td
    return out


def total_cost(prob):
    """computeCost (BaAtLargeOptimizer.cpp:54-63) with the reference's behind-the-camera rule"""
    w, t = prob.cams[prob.obs_cam, 0:3], prob.cams[prob.obs_cam, 3:6]
    X = prob.pts[prob.obs_pt]
    th = np.linalg.norm(w, axis=1, keepdims=True)
    k = w / np.maximum(th, 1e-300)
    P = X * np.cos(th) + np.cross(k, X) * np.sin(th) + k * np.sum(k * X, axis=1, keepdims=True) * (1 - np.cos(th)) + t
    bad = P[:, 2] > 0.01
    err = project(prob.cams[prob.obs_cam], X) - prob.obs_xy
    err[bad] = (25.0, 0.0)
    return 0.5 * float(np.sum(err * err))


def lm_optimize(prob, solver, max_iters=50, lam=1e-5, log=None):
    """the Levenberg-Marquardt loop of BAL_opt (BaAtLargeOptimizer.cpp:186-234), every numeric stage
    on the device: linearise (SE3 tangent) -> Hessian through deviceAccessor() -> factor -> solve ->
    exp-map update; same lambda schedule, same acceptance and convergence rules.  `solver` must have
    6-wide camera blocks (bal_structure(prob, cam_size=6)).  Returns (problem, cost history)."""
    import torch
    hist = []
    last_failed = last_ok = last_good = 0
    for i in range(max_iters):
        pipe = DevicePipeline(prob, solver, param="se3")
        pipe.linearize()
        cost = 0.5 * float((pipe.res * pipe.res).sum())
        H = torch.zeros(solver.dataSize(), dtype=torch.float64, device=pipe.res.device)
        g = torch.zeros(solver.order(), dtype=torch.float64, device=pipe.res.device)
        pipe.fill_hessian(H, g, lam)
        step = g.clone()
        solver.factor(H)
        solver.solve(H, step, solver.order(), 1)
        cand = apply_step_se3(prob, step.cpu().numpy(), solver)
        model_red = 0.5 * float(torch.dot(g, step))
        new_cost = total_cost(cand)
        bad, good = new_cost > cost, new_cost < cost * 0.999
        hist.append((cost, new_cost, lam))
        if log:
            log("[%d] cost %.3e -> %.3e lambda %.1e" % (i, cost, new_cost, lam))
        if bad:
            last_failed = i
            if lam > 1e8 or i > last_ok + 15:
                break
            lam *= 3
            continue
        prob = cand
        last_ok = i
        if good:
            last_good = i
        rel_red = (cost - new_cost) / model_red if model_red != 0 else 0.0
        if rel_red > 0.6:
            lam *= 0.6
        elif rel_red < 0.4:
            lam *= 1.3
        if i >= last_good + 3 and i >= last_failed + 3:
            break
    return prob, hist


class DevicePipeline:
    """problem data resident on the GPU + the two device stages of a Gauss-Newton / LM iteration.
    param = "bal9": cameras are the 9 parameters of a BAL file (BAL_bench's block size, derivatives
    by dual numbers); param = "se3": the reference optimizer's parameterisation (6-wide SE3-tangent
    camera blocks, calibration fixed, closed-form Jacobians of BaAtLarge.h:56-150)"""

    def __init__(self, prob, solver, device="cuda", param="bal9"):
        import torch
        assert param in ("bal9", "se3")
        self.param = param
        self.cam_size = 9 if param == "bal9" else 6
        self.prob, self.solver = prob, solver
        self.lib = _lib.load()
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
        self.obs_cam, self.obs_pt = t(prob.obs_cam, np.int64), t(prob.obs_pt, np.int64)
        self.obs_xy = t(prob.obs_xy, np.float64)
        self.cams, self.pts = t(prob.cams, np.float64), t(prob.pts, np.float64)
        n = len(prob.obs_cam)
        self.res = torch.empty(2 * n, dtype=torch.float64, device=device)
        self.Jc = torch.empty(2 * self.cam_size * n, dtype=torch.float64, device=device)
        self.Jp = torch.empty(6 * n, dtype=torch.float64, device=device)
        self.n_obs = n

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(0 if t is None else t.data_ptr())

    def linearize(self):
        """residuals and Jacobians of every observation (bsp_bal_linearize_f64)"""
        from . import _check
        fn = self.lib.bsp_bal_linearize_f64 if self.param == "bal9" else self.lib.bsp_bal_linearize_se3_f64
        _check(fn(
            ctypes.c_int64(self.n_obs), self._p(self.obs_cam), self._p(self.obs_pt), self._p(self.obs_xy),
            self._p(self.cams), self._p(self.pts), self._p(self.res), self._p(self.Jc), self._p(self.Jp),
            ctypes.c_void_p(0)))
        return self.res

    def fill_hessian(self, data, grad=None, lam=0.0, dbg=None):
        """data (zeroed by the caller) += J^T J through the device accessor, grad += J^T r,
        LM damping (bsp_bal_fill_hessian_*)"""
        import torch
        from . import _check
        f32 = data.dtype == torch.float32
        lam_c = ctypes.c_float(lam) if f32 else ctypes.c_double(lam)
        head = (self.solver._h, ctypes.c_int64(self.prob.num_pts), ctypes.c_int64(self.prob.num_cams),
                ctypes.c_int64(self.n_obs), self._p(self.obs_cam), self._p(self.obs_pt), self._p(self.Jc),
                self._p(self.Jp), self._p(self.res), lam_c, self._p(data), self._p(grad))
        if self.param == "se3":
            fn = self.lib.bsp_bal_fill_hessian_se3_f32 if f32 else self.lib.bsp_bal_fill_hessian_se3_f64
            _check(fn(*head, ctypes.c_void_p(0)))
        else:
            fn = self.lib.bsp_bal_fill_hessian_f32 if f32 else self.lib.bsp_bal_fill_hessian_f64
            _check(fn(*head, self._p(dbg), ctypes.c_void_p(0)))
        return data
