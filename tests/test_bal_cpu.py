"""CPU tests of the BAL caller pipeline's host side: the text loader (format and error behaviour of
benchmarking/BaAtLarge.cpp:81-182), the solver structure of BaAtLargeBench.cpp:44-73, and the
oracle's dual-number Jacobians against central finite differences."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import bal
from oracle import bal_model


def test_bal_round_trip_and_structure(tmp_path):
    prob = bal.synth_scene(num_cams=9, num_pts=120, seed=5)
    path = tmp_path / "problem-9-120-pre.txt"
    bal.save_bal(path, prob)
    back = bal.load_bal(path)
    assert back.num_cams == 9 and back.num_pts == 120
    assert np.array_equal(back.obs_cam, prob.obs_cam) and np.array_equal(back.obs_pt, prob.obs_pt)
    # %.16e round-trips fp64 bit-exactly
    assert np.array_equal(back.cams, prob.cams) and np.array_equal(back.pts, prob.pts)
    assert np.array_equal(back.obs_xy, prob.obs_xy)
    sizes, ss, ranges = bal.bal_structure(back)
    assert list(ranges) == [0, 120] and len(sizes) == 129
    assert np.all(sizes[:120] == 3) and np.all(sizes[120:] == 9)
    # lower-triangle block-CSR: row (120 + cam) lists the points it observes and itself
    for c in range(9):
        row = ss.inds[ss.ptrs[120 + c]:ss.ptrs[121 + c]]
        want = sorted(set(prob.obs_pt[prob.obs_cam == c].tolist()) | {120 + c})
        assert list(row) == want
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    assert sol.order() == 3 * 120 + 9 * 9
    assert list(sol.sparseEliminationRanges())[:2] == [0, 120]
    # bz2, as the BAL site ships the problems
    import bz2
    with bz2.open(str(path) + ".bz2", "wt") as f:
        f.write(open(path).read())
    z = bal.load_bal(str(path) + ".bz2")
    assert np.array_equal(z.cams, prob.cams)


def test_bal_loader_errors(tmp_path):
    """invalid indices and truncated files raise, as Data::load does"""
    prob = bal.synth_scene(num_cams=4, num_pts=20, seed=1)
    bad = bal.BalProblem(prob.cams, prob.pts, prob.obs_cam.copy(), prob.obs_pt.copy(), prob.obs_xy)
    bad.obs_cam[3] = 4
    p = tmp_path / "bad_cam.txt"
    bal.save_bal(p, bad)
    with pytest.raises(RuntimeError, match="3th observation, invalid camera index: 4"):
        bal.load_bal(p)
    bad.obs_cam[3] = 0
    bad.obs_pt[5] = -1
    bal.save_bal(p, bad)
    with pytest.raises(RuntimeError, match="5th observation, invalid point index: -1"):
        bal.load_bal(p)
    good = tmp_path / "good.txt"
    bal.save_bal(good, prob)
    text = open(good).read().split("\n")
    short = tmp_path / "short.txt"
    open(short, "w").write("\n".join(text[:len(text) - 10]))
    with pytest.raises(RuntimeError, match="th point!"):
        bal.load_bal(short)
    with pytest.raises(RuntimeError, match="Cannot open file"):
        bal.load_bal(tmp_path / "missing.txt")


def test_oracle_jacobians_vs_finite_differences():
    prob = bal.synth_scene(num_cams=6, num_pts=40, seed=2)
    cams, pts, xy = prob.cams[prob.obs_cam], prob.pts[prob.obs_pt], prob.obs_xy
    res, Jc, Jp = bal_model.linearize(cams, pts, xy)
    # residual = projection - observation, against the plain (non-dual) camera model
    assert np.allclose(res, bal.project(cams, pts) - xy, rtol=0, atol=1e-9)
    for i in range(9):
        h = 1e-6 * max(1.0, np.abs(cams[:, i]).max())
        cp, cm = cams.copy(), cams.copy()
        cp[:, i] += h
        cm[:, i] -= h
        fd = (bal.project(cp, pts) - bal.project(cm, pts)) / (2 * h)
        assert np.allclose(Jc[:, :, i], fd, rtol=2e-5, atol=1e-5 * np.abs(fd).max()), i
    for i in range(3):
        h = 1e-6
        pp, pm = pts.copy(), pts.copy()
        pp[:, i] += h
        pm[:, i] -= h
        fd = (bal.project(cams, pp) - bal.project(cams, pm)) / (2 * h)
        assert np.allclose(Jp[:, :, i], fd, rtol=2e-5, atol=1e-5 * np.abs(fd).max()), i


def test_host_fill_equals_dense_normal_equations():
    """the accessor-driven assembly (computeStep) == J^T J + damping built densely"""
    prob = bal.synth_scene(num_cams=5, num_pts=30, seed=4)
    sizes, ss, ranges = bal.bal_structure(prob)
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    res, Jc, Jp = bal_model.linearize(prob.cams[prob.obs_cam], prob.pts[prob.obs_pt], prob.obs_xy)
    lam = 0.3
    data, grad, _ = bal_model.fill_hessian_host(sol, prob, Jc, Jp, res, lam)
    n = sol.order()
    perm, sstart = sol.paramToSpan(), sol.skel()["spanStart"]
    J = np.zeros((2 * len(res), n))
    for o, (c, p) in enumerate(zip(prob.obs_cam, prob.obs_pt)):
        ps, cs = int(sstart[perm[p]]), int(sstart[perm[prob.num_pts + c]])
        J[2 * o:2 * o + 2, ps:ps + 3] = Jp[o]
        J[2 * o:2 * o + 2, cs:cs + 9] = Jc[o]
    H = J.T @ J
    H[np.diag_indices(n)] = np.diag(H) * (1 + lam) + 1e-3 * lam
    got = sol.densify(data, fill_upper_half=False)
    assert np.allclose(np.tril(got), np.tril(H), rtol=1e-12, atol=1e-9 * np.abs(H).max())
    assert np.allclose(grad, J.T @ res.reshape(-1), rtol=1e-12, atol=1e-9)


def test_se3_jacobians_are_the_derivatives_of_the_left_perturbation():
    """the oracle's restatement of Cost::compute_residual (BaAtLarge.h:74-147: SE3 tangent of a left
    perturbation, translation first; fixed calibration) against central finite differences of the
    residual under T <- exp(delta) T and X <- X + dX -- the pin of the oracle for the reference's
    parameterisation -- and the reference's behind-the-camera rule (residual (25, 0), zero Jacobians)"""
    from baspacho_amd import bal
    from oracle import bal_model
    prob = bal.synth_scene(num_cams=6, num_pts=40, seed=5)
    cams, pts, xy = prob.cams[prob.obs_cam], prob.pts[prob.obs_pt], prob.obs_xy
    res, Jc, Jp = bal_model.linearize_se3(cams, pts, xy)
    # the residual itself is the BAL model's
    assert np.allclose(res, bal.project(cams, pts) - xy, rtol=1e-12, atol=1e-10)
    R = bal_model._rodrigues(cams[:, 0:3])
    t, calib = cams[:, 3:6], cams[:, 6:9]
    h = 1e-6
    for c in range(6):
        d = np.zeros(6)
        d[c] = h
        Rp, tp = bal.se3_exp(d)
        Rm, tm = bal.se3_exp(-d)
        rp = bal_model.residual_se3(Rp[None] @ R, (Rp @ t.T).T + tp, calib, pts, xy)
        rm = bal_model.residual_se3(Rm[None] @ R, (Rm @ t.T).T + tm, calib, pts, xy)
        fd = (rp - rm) / (2 * h)
        assert np.allclose(fd, Jc[:, :, c], rtol=2e-6, atol=2e-5), c
    for c in range(3):
        d = np.zeros(3)
        d[c] = h
        fd = (bal_model.residual_se3(R, t, calib, pts + d, xy) -
              bal_model.residual_se3(R, t, calib, pts - d, xy)) / (2 * h)
        assert np.allclose(fd, Jp[:, :, c], rtol=2e-6, atol=2e-5), c
    # a point on the wrong side of the image plane
    cams2 = cams.copy()
    cams2[0, 5] += 100.0
    res2, Jc2, Jp2 = bal_model.linearize_se3(cams2, pts, xy)
    assert tuple(res2[0]) == (25.0, 0.0) and not Jc2[0].any() and not Jp2[0].any()
    # exp / log round trip of the update rule
    w = np.array([0.3, -0.2, 0.5])
    assert np.allclose(bal.so3_log(bal.so3_exp(w)), w, atol=1e-12)
