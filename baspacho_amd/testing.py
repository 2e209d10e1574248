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


def connect_ranges_pairs(size, begin1, end1, begin2, end2, fill, max_offset, seed):
    """SparseMatGenerator::connectRanges (TestingMatGen.cpp:23-50) as a list of (row j, col i) pairs,
    i < j: for i in [begin1, end1), j = i + d with min(maxOffset, max(begin2 - i, 1)) <= d <
    min(maxOffset, end2 - i), each kept with probability `fill`; begin1 > begin2 swaps the ranges,
    end1 > end2 adds connectRanges(begin2, end2, end2, end1).  NOTE the reference's semantics: the
    offset limit applies to j - i whatever the ranges are, so two ranges further apart than
    maxOffset get NO connection.  Decisions come from the stateless hash of (i, j)."""
    if begin1 > begin2:
        return connect_ranges_pairs(size, begin2, end2, begin1, end1, fill, max_offset, seed)
    rows, cols = [], []
    if end1 > end2:
        r, c = connect_ranges_pairs(size, begin2, end2, end2, end1, fill, max_offset, seed)
        rows.append(r)
        cols.append(c)
    i = np.arange(begin1, end1, dtype=np.int64)
    d_begin = np.minimum(max_offset, np.maximum(begin2 - i, 1))
    d_end = np.minimum(max_offset, end2 - i)
    for d in range(1, int(max_offset)):
        ii = i[(d >= d_begin) & (d < d_end)]
        if len(ii) == 0:
            continue
        jj = ii + d
        if fill < 1.0:
            keep = hash_unit(seed, ii.astype(np.uint64) * np.uint64(size) + jj.astype(np.uint64)) < fill
            ii, jj = ii[keep], jj[keep]
        rows.append(jj)
        cols.append(ii)
    if not rows:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    return np.concatenate(rows), np.concatenate(cols)


def gen_meridians(num, line_len, fill, band, hair_len, n_pole_hairs, s_pole_hairs, seed=37):
    """SparseMatGenerator::genMeridians (TestingMatGen.cpp:87-168), call for call: `num` tracks of
    `line_len` parameters and n + s "hair" tracks of `hair_len`, each a band of width `band`
    (neighbours connected with probability `fill`), then the pole connections (meridian-meridian,
    meridian-hair, hair-hair), every one through connectRanges with maxOffset = band.  Because that
    limit is on the INDEX distance (see connect_ranges_pairs), pole connections between tracks whose
    index ranges are further apart than `band` are empty in the reference, and so they are here; the
    south-pole hair-hair loop's `kBegin` is computed from h, not k, in the reference (a track with
    itself): restated as written."""
    tot_hairs = n_pole_hairs + s_pole_hairs
    size = line_len * num + hair_len * tot_hairs
    end_meridians = line_len * num
    assert band <= line_len and band <= hair_len
    acc = []

    def conn(b1, e1, b2, e2):
        acc.append(connect_ranges_pairs(size, b1, e1, b2, e2, fill, band, seed))

    for i in range(num):
        b = line_len * i
        conn(b, b + line_len, b, b + line_len)
    for h in range(tot_hairs):
        b = end_meridians + hair_len * h
        conn(b, b + hair_len, b, b + hair_len)
    for i in range(num):
        ib = line_len * i
        for j in range(i):
            jb = line_len * j
            conn(ib, ib + band, jb, jb + band)
            conn(ib + line_len - band, ib + line_len, jb + line_len - band, jb + line_len)
    for i in range(num):
        ib = line_len * i
        for h in range(n_pole_hairs):
            hb = end_meridians + hair_len * h
            conn(ib, ib + band, hb, hb + band)
        for h in range(s_pole_hairs):
            hb = end_meridians + hair_len * (h + n_pole_hairs)
            conn(ib + line_len - band, ib + line_len, hb, hb + band)
    for h in range(n_pole_hairs):
        hb = end_meridians + hair_len * h
        for k in range(h):
            kb = end_meridians + hair_len * k
            conn(kb, kb + band, hb, hb + band)
    for h in range(s_pole_hairs):
        hb = end_meridians + hair_len * (h + n_pole_hairs)
        for k in range(h):
            kb = end_meridians + hair_len * (h + n_pole_hairs)   # (sic: h, TestingMatGen.cpp:160)
            conn(kb, kb + band, hb, hb + band)
    rows = np.concatenate([a[0] for a in acc])
    cols = np.concatenate([a[1] for a in acc])
    return structure_from_pairs(size, rows, cols)


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


def gen_bal_clustered(num_cams=871, num_pts=527480, mean_track=8.0, cluster_min=8, cluster_max=30,
                      link_prob=0.12, far_prob=0.01, degree_power=0.6, seed=37):
    """A second stand-in for a BAL problem, with CLUSTERED co-visibility (round 6: the headline's
    default stand-in draws a point's cameras uniformly in a +-48 band, so two cameras share few points;
    real structure-from-motion scenes are groups of cameras looking at the same thing).

    Cameras are cut into consecutive clusters of cluster_min .. cluster_max cameras; a point belongs to
    one cluster (bigger clusters get more points) and sees cameras of THAT cluster, drawn with a
    power-law popularity inside the cluster (camera of rank k with weight k^-degree_power: a few
    cameras see most of the scene); an observation leaves for the next / previous cluster with
    probability link_prob (what keeps the reduced camera system connected) and for a uniformly random
    camera with probability far_prob.  Track lengths and sizes as gen_bal_synthetic.
    returns (param_sizes, SparseStructure, obs_cam, obs_pt)
    """
    rng = Rng(seed)
    bounds = [0]
    k = 0
    while bounds[-1] < num_cams:
        width = cluster_min + int(hash_unit(seed + 11, np.array([k], dtype=np.uint64))[0] * (cluster_max - cluster_min + 1))
        bounds.append(min(num_cams, bounds[-1] + width))
        k += 1
    if bounds[-1] - bounds[-2] < 2 and len(bounds) > 2:
        bounds.pop(-2)
    bounds = np.array(bounds, dtype=np.int64)
    n_cl = len(bounds) - 1
    csize = np.diff(bounds)
    # cluster of every point: weight size^1.3, points ordered by cluster (as a pipeline emits them)
    wts = csize.astype(np.float64) ** 1.3
    cdf = np.cumsum(wts) / wts.sum()
    cl = np.sort(np.searchsorted(cdf, rng.unit(num_pts), side="right").clip(0, n_cl - 1))
    u = rng.unit(num_pts)
    tail = rng.unit(num_pts) < 0.06
    body_mean = max(mean_track - 2.0 - 0.06 * 18.0, 0.5)
    p_body = 1.0 / (1.0 + body_mean)
    extra = np.floor(np.log1p(-u) / np.log1p(-p_body)).astype(np.int64)
    extra_tail = np.floor(np.log1p(-u) / np.log1p(-1.0 / 19.0)).astype(np.int64)
    track = 2 + np.where(tail, extra_tail, extra)
    track = np.minimum(track, np.maximum(2, csize[cl] + 4))
    n_obs = int(track.sum())
    pt = np.repeat(np.arange(num_pts, dtype=np.int64), track)
    hop = rng.unit(n_obs)
    oc = cl[pt] + np.where(hop < link_prob / 2, -1, np.where(hop < link_prob, 1, 0))
    oc = np.clip(oc, 0, n_cl - 1)
    # power-law rank inside the cluster: rank = floor(size * v^(1 / (1 - p))) concentrates on rank 0
    v = rng.unit(n_obs)
    rank = np.floor(csize[oc] * v ** (1.0 / (1.0 - degree_power))).astype(np.int64)
    cam = bounds[oc] + np.minimum(rank, csize[oc] - 1)
    far = rng.unit(n_obs) < far_prob
    cam = np.where(far, np.floor(rng.unit(n_obs) * num_cams).astype(np.int64), cam)
    key = np.unique(pt * num_cams + cam)
    pt, cam = key // num_cams, key % num_cams
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

