"""ORACLE (test infrastructure only): build + ctypes bindings of the C restatement.

`build()` compiles oracle/ref_factor.c (and blas_factor.c) with gcc into oracle/*.so
(git-ignored; they travel to the GPU box with the snapshot)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_I64P = ctypes.POINTER(ctypes.c_int64)

_SKEL_FIELDS = ["spanStart", "spanToLump", "lumpStart", "lumpToSpan", "spanOffsetInLump",
                "chainColPtr", "chainRowSpan", "chainData", "chainRowsTillEnd",
                "boardColPtr", "boardRowLump", "boardChainColOrd",
                "boardRowPtr", "boardColLump", "boardColOrd"]


class OrcSkel(ctypes.Structure):
    _fields_ = [("numSpans", ctypes.c_int64), ("numLumps", ctypes.c_int64)] + \
               [(f, _I64P) for f in _SKEL_FIELDS]


def _needs_build(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = src_list + [os.path.join(_HERE, "ref_factor_impl.inc"), os.path.join(_HERE, "orc_skel.h")]
    return any(os.path.getmtime(s) > t for s in deps if os.path.exists(s))


def build(force=False):
    out = os.path.join(_HERE, "liboracle_ref.so")
    src = os.path.join(_HERE, "ref_factor.c")
    if force or _needs_build([src], out):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-shared", "-fPIC", src, "-o", out, "-lm"])
    out2 = os.path.join(_HERE, "liboracle_blas.so")
    src2 = os.path.join(_HERE, "blas_factor.c")
    if os.path.exists(src2) and (force or _needs_build([src2], out2)):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-fopenmp", "-shared", "-fPIC", src2, "-o",
                               out2, "-lm", "-ldl"])
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        _lib = ctypes.CDLL(path)
    return _lib


class SkelHandle:
    """keeps the numpy arrays alive next to the C struct"""

    def __init__(self, sk):
        self.arrays = {f: np.ascontiguousarray(sk[f], dtype=np.int64) for f in _SKEL_FIELDS}
        self.c = OrcSkel()
        self.c.numSpans = len(self.arrays["spanStart"]) - 1
        self.c.numLumps = len(self.arrays["lumpStart"]) - 1
        for f in _SKEL_FIELDS:
            setattr(self.c, f, self.arrays[f].ctypes.data_as(_I64P))


def _suffix(data):
    if data.dtype == np.float64:
        return "f64", ctypes.c_double
    if data.dtype == np.float32:
        return "f32", ctypes.c_float
    raise TypeError(data.dtype)


def factor(sk, data, elim_ranges=(), start_span=0, end_span=None):
    """Solver::factor / factorUpTo / factorFrom on the CPU restatement; in place on `data`."""
    h = sk if isinstance(sk, SkelHandle) else SkelHandle(sk)
    assert data.flags["C_CONTIGUOUS"]
    sfx, ct = _suffix(data)
    ranges = np.ascontiguousarray(elim_ranges, dtype=np.int64)
    if end_span is None:
        end_span = h.c.numSpans
    fn = getattr(lib(), "orc_factor_range_" + sfx)
    fn.restype = ctypes.c_int
    rc = fn(ctypes.byref(h.c), ranges.ctypes.data_as(_I64P), ctypes.c_int64(len(ranges)),
            data.ctypes.data_as(ctypes.POINTER(ct)), ctypes.c_int64(start_span),
            ctypes.c_int64(end_span))
    if rc != 0:
        raise RuntimeError("oracle factor_range failed: %d" % rc)
    return data


def do_elimination(sk, data, lump_begin, lump_end):
    h = sk if isinstance(sk, SkelHandle) else SkelHandle(sk)
    sfx, ct = _suffix(data)
    fn = getattr(lib(), "orc_do_elimination_" + sfx)
    fn(ctypes.byref(h.c), data.ctypes.data_as(ctypes.POINTER(ct)), ctypes.c_int64(lump_begin),
       ctypes.c_int64(lump_end))
    return data


def _solve(name, sk, data, vec, ldc, nrhs, start_lump, up_to_lump):
    h = sk if isinstance(sk, SkelHandle) else SkelHandle(sk)
    sfx, ct = _suffix(data)
    assert vec.dtype == data.dtype
    if up_to_lump is None:
        up_to_lump = h.c.numLumps
    fn = getattr(lib(), name + sfx)
    fn(ctypes.byref(h.c), data.ctypes.data_as(ctypes.POINTER(ct)),
       vec.ctypes.data_as(ctypes.POINTER(ct)), ctypes.c_int64(ldc), ctypes.c_int64(nrhs),
       ctypes.c_int64(start_lump), ctypes.c_int64(up_to_lump))
    return vec


def solve_l(sk, data, vec, ldc, nrhs, start_lump=0, up_to_lump=None):
    """vec: flat column-major (order x nrhs) buffer with leading dimension ldc; in place"""
    return _solve("orc_solve_l_", sk, data, vec, ldc, nrhs, start_lump, up_to_lump)


def solve_lt(sk, data, vec, ldc, nrhs, start_lump=0, up_to_lump=None):
    return _solve("orc_solve_lt_", sk, data, vec, ldc, nrhs, start_lump, up_to_lump)


def solve(sk, data, vec, ldc, nrhs):
    solve_l(sk, data, vec, ldc, nrhs)
    return solve_lt(sk, data, vec, ldc, nrhs)


# ---- BLAS restatement of BackendFast (cpu_baseline of bench.py) --------------------------------
_blas = None


def find_openblas():
    """system libopenblas if present, else the OpenBLAS bundled in the scipy wheel"""
    import glob
    cands = []
    for d in ("/usr/lib/x86_64-linux-gnu", "/usr/lib64", "/usr/lib", "/opt/OpenBLAS/lib"):
        cands += sorted(glob.glob(os.path.join(d, "libopenblas*.so*")))
    try:
        import scipy
        cands += sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs",
                                               "libscipy_openblas*.so")))
    except Exception:
        pass
    return cands[0] if cands else None


def blas_lib(num_threads=0):
    """load oracle/liboracle_blas.so and bind it to an OpenBLAS; returns (lib, description)"""
    global _blas
    if _blas is None:
        build()
        lib_ = ctypes.CDLL(os.path.join(_HERE, "liboracle_blas.so"))
        lib_.orc_blas_desc.restype = ctypes.c_char_p
        path = find_openblas()
        if path is None:
            raise RuntimeError("no OpenBLAS found for the CPU baseline")
        rc = lib_.orc_blas_init(path.encode(), ctypes.c_int(num_threads))
        if rc != 0:
            raise RuntimeError("orc_blas_init: " + lib_.orc_blas_desc().decode())
        _blas = lib_
    return _blas, _blas.orc_blas_desc().decode()


def blas_factor(sk, data, elim_ranges=(), num_threads=0):
    """Solver::factor on the BLAS restatement (fp64, in place); returns seconds spent in the
    sparse-elimination part"""
    lib_, _ = blas_lib(num_threads)
    h = sk if isinstance(sk, SkelHandle) else SkelHandle(sk)
    assert data.dtype == np.float64 and data.flags["C_CONTIGUOUS"]
    ranges = np.ascontiguousarray(elim_ranges, dtype=np.int64)
    elim_s = ctypes.c_double(0)
    rc = lib_.orc_blas_factor_f64(ctypes.byref(h.c), ranges.ctypes.data_as(_I64P),
                                  ctypes.c_int64(len(ranges)),
                                  data.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                  ctypes.byref(elim_s))
    if rc != 0:
        raise RuntimeError("oracle blas factor failed: %d" % rc)
    return elim_s.value


def probe_residual(sk, A, L, x):
    """|| L (L^T x) - A x || / || A x || through the skeleton (fp64 host arrays)"""
    h = sk if isinstance(sk, SkelHandle) else SkelHandle(sk)
    out = np.zeros(2, dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib().orc_probe_residual_f64(ctypes.byref(h.c), A.ctypes.data_as(dp), L.ctypes.data_as(dp),
                                      x.ctypes.data_as(dp), out.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError("probe failed")
    return out[0] / out[1]


def abs_row_sums(sk, A):
    """sum_{j != i} |A_ij| for every row of the symmetric matrix laid out by the skeleton (test-input
    helper: matrices that are diagonally dominant by a few percent only)"""
    h = sk if isinstance(sk, SkelHandle) else SkelHandle(sk)
    n = int(h.arrays["spanStart"][-1])
    out = np.zeros(n, dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib().orc_abs_row_sums_f64(ctypes.byref(h.c), A.ctypes.data_as(dp), out.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError("abs_row_sums failed")
    return out
