"""Synthetic problem generators (host side), the counterpart of the reference's
baspacho/testing/{TestingUtils,TestingMatGen}.cpp.  The reference seeds std::mt19937 and
libstdc++ distributions (not reproducible elsewhere); here every generator is a pure function
of a fully specified counter-based RNG (splitmix64), so fixtures are reproducible anywhere.
"""
import numpy as np

from . import SparseStructure

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(x):
    """splitmix64 finaliser on a uint64 array"""
    with np.errstate(over="ignore"):
        z = (x + _GOLDEN).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def hash_u64(seed, idx):
    """stateless: uint64 hash of (seed, idx)"""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = _mix(np.asarray([seed], dtype=np.uint64) * _GOLDEN)[0]
        return _mix(idx * _GOLDEN + s)


def hash_unit(seed, idx):
    """stateless uniform [0,1) doubles"""
    return (hash_u64(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class Rng:
    """sequential stream on top of the stateless hash"""

    def __init__(self, seed):
        self.seed = int(seed)
        self.counter = 0

    def unit(self, n):
        out = hash_unit(self.seed, np.arange(self.counter, self.counter + n, dtype=np.uint64))
        self.counter += n
        return out

    def uniform(self, n, lo, hi):
        return lo + (hi - lo) * self.unit(n)

    def integers(self, n, lo, hi):
        """inclusive bounds, like randomVec (TestingUtils.cpp:30-38)"""
        return (lo + np.floor(self.unit(n) * (hi - lo + 1))).astype(np.int64)


def random_data(size, lo, hi, seed, dtype=np.float64):
    """randomData (TestingUtils.cpp:40-52)"""
    return Rng(seed).uniform(size, lo, hi).astype(dtype)


def random_vec(size, lo, hi, seed):
    return Rng(seed).integers(size, lo, hi)


# ---- block patterns ------------------------------------------------------------------------
def structure_from_pairs(n, rows, cols):
    """lower-triangular block CSR (diagonal included) from arbitrary (row, col) block pairs"""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    r = np.concatenate([np.maximum(rows, cols), np.arange(n, dtype=np.int64)])
    c = np.concatenate([np.minimum(rows, cols), np.arange(n, dtype=np.int64)])
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    ptrs = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptrs, r + 1, 1)
    ptrs = np.cumsum(ptrs)
    return SparseStructure(ptrs, c)


def columns_to_structure(columns):
    """columnsToCscStruct(columns).transpose() (TestingUtils.cpp:186-194): columns[i] lists the
    rows >= i of column i; result = CSR of the lower triangle"""
    n = len(columns)
    rows = np.fromiter((r for col in columns for r in col), dtype=np.int64)
    cols = np.fromiter((i for i, col in enumerate(columns) for _ in col), dtype=np.int64)
    return structure_from_pairs(n, rows, cols)


def structure_to_columns(ss):
    n = ss.order()
    cols = [set() for _ in range(n)]
    for i in range(n):
        for k in range(ss.ptrs[i], ss.ptrs[i + 1]):
            cols[int(ss.inds[k])].add(i)
    return cols


def random_cols(size, fill, seed):
    """randomCols (TestingUtils.cpp:139-152): column i = {i} + each j>i with probability fill"""
    cols = []
    for i in range(size):
        j = np.arange(i + 1, size, dtype=np.int64)
        keep = hash_unit(seed, np.uint64(i) * np.uint64(size) + j.astype(np.uint64)) < fill
        cols.append({i} | set(int(x) for x in j[keep]))
    return cols


def make_independent_elim_set(columns, start, end):
    """makeIndependentElimSet (TestingUtils.cpp:214-230)"""
    out = []
    for i, col in enumerate(columns):
        if i < start or i >= end:
            out.append(set(col))
        else:
            out.append({i} | {c for c in col if c >= end})
    return out


def gen_flat_pairs(size, fill, seed, chunk=2048):
    """SparseMatGenerator::genFlat (TestingMatGen.cpp:72-76): every pair i<j with prob. fill"""
    rows, cols = [], []
    thr = fill
    for i0 in range(0, size, chunk):
        i = np.arange(i0, min(size, i0 + chunk), dtype=np.uint64)
        j = np.arange(size, dtype=np.uint64)
        u = hash_unit(seed, (i[:, None] * np.uint64(size) + j[None, :]).ravel()).reshape(len(i), size)
        mask = (u < thr) & (j[None, :] > i[:, None])
        ii, jj = np.nonzero(mask)
        rows.append(jj.astype(np.int64))
        cols.append(ii.astype(np.int64) + i0)
    return np.concatenate(rows), np.concatenate(cols)


def gen_flat(size, fill, seed=37):
    r, c = gen_flat_pairs(size, fill, seed)
    return structure_from_pairs(size, r, c)


def gen_grid(width, height, fill=1.0, conn=2, seed=37):
    """SparseMatGenerator::genGrid (TestingMatGen.cpp:170-200): params on a grid, each connected
    to the neighbours within Chebyshev distance `conn` (with probability `fill`)"""
    n = width * height
    i, j = np.meshgrid(np.arange(width), np.arange(height), indexing="ij")
    i, j = i.ravel(), j.ravel()
    off = i * height + j
    rows, cols = [], []
    for di in range(-conn, conn + 1):
        for dj in range(-conn, conn + 1):
            if di == 0 and dj == 0:
                continue
            i2, j2 = i + di, j + dj
            ok = (i2 >= 0) & (i2 < width) & (j2 >= 0) & (j2 < height)
            o1, o2 = off[ok], (i2 * height + j2)[ok]
            if fill < 1.0:
                keep = hash_unit(seed, (o1 * n + o2).astype(np.uint64)) < fill
                o1, o2 = o1[keep], o2[keep]
            rows.append(o1)
            cols.append(o2)
    return structure_from_pairs(n, np.concatenate(rows), np.concatenate(cols))


def block_tridiagonal(n):
    """column i holds blocks {i, i+1}"""
    i = np.arange(n - 1, dtype=np.int64)
    return structure_from_pairs(n, i + 1, i)


def add_schur_set(ss, size, fill, seed):
    """SparseMatGenerator::addSchurSet (TestingMatGen.cpp:52-70): prepend `size` independent
    params, each connected to every old param with probability `fill`"""
    n_old = ss.order()
    n = n_old + size
    row_of = np.repeat(np.arange(n_old, dtype=np.int64), np.diff(ss.ptrs))
    rows = [row_of + size]
    cols = [ss.inds + size]
    for i0 in range(0, size, 4096):
        i = np.arange(i0, min(size, i0 + 4096), dtype=np.uint64)
        j = np.arange(n_old, dtype=np.uint64)
        u = hash_unit(seed, (i[:, None] * np.uint64(n_old) + j[None, :]).ravel())
        ii, jj = np.nonzero(u.reshape(len(i), n_old) < fill)
        rows.append(jj.astype(np.int64) + size)
        cols.append(ii.astype(np.int64) + i0)
    return structure_from_pairs(n, np.concatenate(rows), np.concatenate(cols))


# ---- bundle-adjustment-at-large stand-in -------------------------------------------------------
def gen_bal_synthetic(num_cams=871, num_pts=527480, mean_track=5.8, band=48, far_prob=0.04,
                      seed=37):
    """Synthetic stand-in for a BAL problem (the datasets are not available offline).

    Same bipartite shape as benchmarking/BaAtLargeBench.cpp:50-65 builds from a BAL file:
    points first (size 3), cameras after (size 9), one off-diagonal block per observation.
    Cameras sit on a line; every point has a centre camera, a track length >= 2 with a heavy
    tail (shifted geometric mixture; the default mean_track=5.8 yields 2.79 M distinct
    observations for 871 x 527480, matching BAL problem-871-527480) and sees cameras drawn around its
    centre within +-band, except that each observation jumps to a uniformly random camera with
    probability far_prob (loop closures).  Points are ordered by centre camera, as
    reconstruction pipelines emit them.
    returns (param_sizes, SparseStructure, obs_cam, obs_pt)
    """
    rng = Rng(seed)
    centre = np.sort(np.floor(rng.unit(num_pts) * num_cams).astype(np.int64))
    # track length: 2 + geometric body, with a 6% heavy tail
    u = rng.unit(num_pts)
    tail = rng.unit(num_pts) < 0.06
    body_mean = max(mean_track - 2.0 - 0.06 * 18.0, 0.5)
    p_body = 1.0 / (1.0 + body_mean)
    extra = np.floor(np.log1p(-u) / np.log1p(-p_body)).astype(np.int64)
    extra_tail = np.floor(np.log1p(-u) / np.log1p(-1.0 / 19.0)).astype(np.int64)
    track = 2 + np.where(tail, extra_tail, extra)
    track = np.minimum(track, min(num_cams, 2 * band + 1))
    n_obs = int(track.sum())
    pt = np.repeat(np.arange(num_pts, dtype=np.int64), track)
    off = rng.unit(n_obs)
    far = rng.unit(n_obs) < far_prob
    farcam = np.floor(rng.unit(n_obs) * num_cams).astype(np.int64)
    near = centre[pt] + np.floor((off * 2.0 - 1.0) * band).astype(np.int64)
    cam = np.where(far, farcam, np.clip(near, 0, num_cams - 1))
    # one block per (point, camera): drop duplicate observations
    key = np.unique(pt * num_cams + cam)
    pt, cam = key // num_cams, key % num_cams
    # guarantee >= 2 distinct cameras per point
    cnt = np.bincount(pt, minlength=num_pts)
    lonely = np.nonzero(cnt < 2)[0]
    if len(lonely):
        first_cam = np.full(num_pts, -1, dtype=np.int64)
        first_cam[pt[::-1]] = cam[::-1]
        extra_cam = (first_cam[lonely] + 1) % num_cams
        pt = np.concatenate([pt, lonely])
        cam = np.concatenate([cam, extra_cam])
        key = np.unique(pt * num_cams + cam)
        pt, cam = key // num_cams, key % num_cams
    n = num_pts + num_cams
    sizes = np.concatenate([np.full(num_pts, 3, dtype=np.int64), np.full(num_cams, 9, dtype=np.int64)])
    ss = structure_from_pairs(n, num_pts + cam, pt)
    return sizes, ss, cam, pt
