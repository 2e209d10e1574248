"""baspacho_amd -- MI355X-native supernodal sparse Cholesky (host-side Python mirror).

Thin mirror of the reference's public C++ surface (baspacho/baspacho/Solver.h) over the C ABI
of the HIP library (include/baspacho_amd.h): `create_solver`, `Solver.factor/solve`, block
accessors, skeleton arrays.  PyTorch is only plumbing here (device memory, streams,
torch.distributed); numeric work happens in the hand-written HIP kernels.
"""
import ctypes
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib

__all__ = ["Settings", "Solver", "create_solver", "SparseStructure", "BackendHip", "BackendCuda",
           "BackendRef", "BackendFast", "AddFillComplete", "AddFillForAutoElims",
           "AddFillForGivenElims", "AddFillNone"]

# Solver.h:189-209
BackendRef, BackendFast, BackendCuda, BackendHip = 0, 1, 2, 3
AddFillComplete, AddFillForAutoElims, AddFillForGivenElims, AddFillNone = 0, 1, 2, 3

_I64P = ctypes.POINTER(ctypes.c_int64)

_SKEL_IDS = {
    "spanStart": 0, "spanToLump": 1, "lumpStart": 2, "lumpToSpan": 3, "spanOffsetInLump": 4,
    "chainColPtr": 5, "chainRowSpan": 6, "chainData": 7, "chainRowsTillEnd": 8,
    "boardColPtr": 9, "boardRowLump": 10, "boardChainColOrd": 11, "boardRowPtr": 12,
    "boardColLump": 13, "boardColOrd": 14,
}

PROF_KINDS = ["elim_factor", "elim_update", "potrf", "trsm", "update", "chain_update"]


class _CSettings(ctypes.Structure):
    _fields_ = [("find_sparse_elimination_ranges", ctypes.c_int32),
                ("num_threads", ctypes.c_int32), ("backend", ctypes.c_int32),
                ("add_fill_policy", ctypes.c_int32),
                ("computation_model", ctypes.POINTER(ctypes.c_double))]


_HIP_OPTION_INTS = ["lookahead", "due_stream", "split_k", "gather_max_pairs", "gather_overlap",
                    "sub_batch_min", "sub_batches", "tail_blocks", "lazy_plan", "block_solve", "solve_inv",
                    "solve_sweep", "sweep_min_width", "solve_wide", "chain_contraction", "dense_merge", "expected_batch"]
_HIP_OPTION_REALS = ["lookahead_min_gf", "bulk_ahead", "level_cost_us"]


class _CHipOptions(ctypes.Structure):
    """bsp_hip_options (include/baspacho_amd.h): negative / NaN = the library's default"""
    _fields_ = [(n, ctypes.c_int32) for n in _HIP_OPTION_INTS] + [(n, ctypes.c_double) for n in _HIP_OPTION_REALS]


class _CPlanStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in
                ["flops", "upd_elems", "upd_flops", "elim_pair_elems", "elim_pair_flops",
                 "elim_col_elems", "upd_flops_direct", "elim_pair_operand_elems",
                 "elim_target_elems", "trsm_flops", "potrf_flops", "trsm_flops_merged",
                 "potrf_flops_fused"]] + \
               [(n, ctypes.c_int64) for n in
                ["num_launches", "num_levels", "num_panels", "num_segs", "num_upd_tasks",
                 "num_trsm_tasks", "chain_tab_entries", "max_panels_in_level",
                 "num_atomic_upd_tasks", "num_gather_groups", "num_fork_levels"]] + \
               [("deferred_flops", ctypes.c_double), ("tail_upd_flops", ctypes.c_double),
                ("num_tail_panels", ctypes.c_int64)]


class _CRunCounters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in
                ["sweep_launches", "sweep_timeouts", "split_lists_used", "sub_batches_enqueued",
                 "lookahead_forks", "sweeps_retired", "sweep_error_pending",
                 "gather_chunks_overlapped", "tail_launches", "sweep_mfma_launches",
                 "solve_wide_launches", "inv_reused", "potrf_folded_levels"]]


@dataclass
class Settings:
    """Solver.h:212-218"""
    findSparseEliminationRanges: bool = True
    numThreads: int = 16
    backend: int = BackendHip
    addFillPolicy: int = AddFillComplete
    computationModel: Optional[Sequence[float]] = None  # 20 coefficients, see C header
    # extension: schedule switches of the MI355X backend, {field of bsp_hip_options: value}; fields left
    # out keep the library's default (the BSP_* environment variables override both, for A/B scripts)
    hipOptions: Optional[dict] = None


@dataclass
class SparseStructure:
    """block CSR pattern (lower triangle incl. diagonal); SparseStructure.h:19-29"""
    ptrs: np.ndarray
    inds: np.ndarray

    def __post_init__(self):
        self.ptrs = np.ascontiguousarray(self.ptrs, dtype=np.int64)
        self.inds = np.ascontiguousarray(self.inds, dtype=np.int64)

    def order(self):
        return len(self.ptrs) - 1

    # SparseStructure.h:34-55 (executed by the C++ host library)
    def transpose(self):
        return _ss_op(self, SS_TRANSPOSE)

    def clear(self, clearLower=True):
        return _ss_op(self, SS_CLEAR, flag=int(clearLower))

    def symmetricPermutation(self, mapPerm, lowerHalf=True):
        return _ss_op(self, SS_SYM_PERMUTATION, mapPerm, int(lowerHalf))

    def addIndependentEliminationFill(self, start, end):
        return _ss_op(self, SS_INDEP_ELIM_FILL, [start, end])

    def addFullEliminationFill(self):
        return _ss_op(self, SS_FULL_ELIM_FILL)

    def fillReducingPermutation(self):
        return _ss_op(self, SS_FILL_REDUCING_PERM)

    def extractRightBottom(self, start):
        return _ss_op(self, SS_EXTRACT_RIGHT_BOTTOM, [start])


SS_TRANSPOSE, SS_CLEAR, SS_SYM_PERMUTATION, SS_INDEP_ELIM_FILL, SS_FULL_ELIM_FILL, \
    SS_FILL_REDUCING_PERM, SS_EXTRACT_RIGHT_BOTTOM = range(7)


def _ss_op(ss, op, arg=(), flag=0):
    lib = _lib.load()
    n = ss.order()
    a = _i64(arg)
    cap = max(16, 4 * len(ss.inds) + n)
    while True:
        out_ptrs = np.zeros(n + 2, dtype=np.int64)
        out_inds = np.zeros(cap, dtype=np.int64)
        on, onnz = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = lib.bsp_sparse_structure_op(
            op, ctypes.c_int64(n), ss.ptrs.ctypes.data_as(_I64P), ss.inds.ctypes.data_as(_I64P),
            a.ctypes.data_as(_I64P), ctypes.c_int64(len(a)), ctypes.c_int32(flag),
            out_ptrs.ctypes.data_as(_I64P), out_inds.ctypes.data_as(_I64P), ctypes.c_int64(cap),
            ctypes.byref(on), ctypes.byref(onnz))
        if rc != 0 and onnz.value > cap:
            cap = onnz.value
            continue
        _check(rc)
        if op == SS_FILL_REDUCING_PERM:
            return out_inds[:onnz.value].copy()
        return SparseStructure(out_ptrs[:on.value + 1].copy(), out_inds[:onnz.value].copy())


def _check(rc):
    if rc != 0:
        raise RuntimeError(_lib.load().bsp_last_error().decode("utf-8", "replace"))


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr_of(t):
    """device pointer of a torch tensor (or a raw int address)"""
    if isinstance(t, int):
        return t
    if not t.is_cuda:
        raise ValueError("numeric data must live in device memory (Solver.h:184-188)")
    if not t.is_contiguous():
        raise ValueError("numeric data must be contiguous")
    return t.data_ptr()


def _suffix(t):
    import torch
    if t.dtype == torch.float64:
        return "f64"
    if t.dtype == torch.float32:
        return "f32"
    raise TypeError("only float64 / float32 data is supported, got %s" % t.dtype)


class Solver:
    """Mirror of BaSpaCho::Solver (Solver.h:34-180).  Do not construct directly: use
    `create_solver`, `Solver.from_skeleton` or `Solver.from_plan`."""

    def __init__(self, handle):
        self._lib = _lib.load()
        self._h = handle
        self._skel = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.bsp_destroy_solver(h)

    # ---- construction -------------------------------------------------------------------
    @staticmethod
    def from_skeleton(span_start, lump_to_span, col_ptr, row_ind, sparse_elim_ranges=()):
        """Solver(CoalescedBlockMatrixSkel&&, sparseElimRanges, {}, ops)  (Solver.h:37-38)"""
        lib = _lib.load()
        ss, l2s, cp, ri = _i64(span_start), _i64(lump_to_span), _i64(col_ptr), _i64(row_ind)
        er = _i64(sparse_elim_ranges)
        h = ctypes.c_void_p()
        _check(lib.bsp_create_solver_from_skeleton(
            ctypes.c_int64(len(ss) - 1), ss.ctypes.data_as(_I64P), ctypes.c_int64(len(l2s) - 1),
            l2s.ctypes.data_as(_I64P), cp.ctypes.data_as(_I64P), ri.ctypes.data_as(_I64P),
            ctypes.c_int64(len(er)), er.ctypes.data_as(_I64P), ctypes.byref(h)))
        return Solver(h)

    @staticmethod
    def from_plan(buf):
        """rebuild a solver from `serialize_plan()` output (e.g. after an RCCL broadcast)"""
        lib = _lib.load()
        b = _i64(buf)
        h = ctypes.c_void_p()
        _check(lib.bsp_create_solver_from_plan(b.ctypes.data_as(_I64P), ctypes.c_int64(len(b)),
                                               ctypes.byref(h)))
        return Solver(h)

    def serialize_plan(self):
        need = ctypes.c_int64(0)
        _check(self._lib.bsp_plan_serialize(self._h, None, ctypes.c_int64(0), ctypes.byref(need)))
        buf = np.empty(need.value, dtype=np.int64)
        _check(self._lib.bsp_plan_serialize(self._h, buf.ctypes.data_as(_I64P), need,
                                            ctypes.byref(need)))
        return buf

    # ---- sizes / structure --------------------------------------------------------------
    def order(self):
        return int(self._lib.bsp_order(self._h))

    def dataSize(self):
        return int(self._lib.bsp_data_size(self._h))

    def canFactorUpToSpan(self):
        return int(self._lib.bsp_can_factor_up_to_span(self._h))

    def numSpans(self):
        return int(self._lib.bsp_num_spans(self._h))

    def numLumps(self):
        return int(self._lib.bsp_num_lumps(self._h))

    def spanVectorOffset(self, span):
        return int(self._lib.bsp_span_vector_offset(self._h, ctypes.c_int64(span)))

    def spanMatrixOffset(self, span):
        out = ctypes.c_int64(0)
        _check(self._lib.bsp_span_matrix_offset(self._h, ctypes.c_int64(span), ctypes.byref(out)))
        return out.value

    def _array(self, which):
        p, n = _I64P(), ctypes.c_int64(0)
        _check(self._lib.bsp_skeleton_array(self._h, which, ctypes.byref(p), ctypes.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.int64)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def skel(self):
        """dict of the CoalescedBlockMatrixSkel arrays (CoalescedBlockMatrix.h:88-110)"""
        if self._skel is None:
            self._skel = {k: self._array(v) for k, v in _SKEL_IDS.items()}
        return self._skel

    def paramToSpan(self):
        return self._array(15)

    def sparseEliminationRanges(self):
        return self._array(16)

    # ---- accessors ----------------------------------------------------------------------
    def blockOffset(self, row_param, col_param):
        """(offset, stride, flipped)  -- PermutedCoalescedAccessor::blockOffset"""
        off, st, fl = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0)
        _check(self._lib.bsp_block_offset(self._h, ctypes.c_int64(row_param),
                                          ctypes.c_int64(col_param), ctypes.byref(off),
                                          ctypes.byref(st), ctypes.byref(fl)))
        return off.value, st.value, bool(fl.value)

    def diagBlockOffset(self, param):
        off, st = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self._lib.bsp_diag_block_offset(self._h, ctypes.c_int64(param), ctypes.byref(off),
                                               ctypes.byref(st)))
        return off.value, st.value

    def diagBlockOffsetOfSpan(self, span):
        """(offset, stride) of the diagonal block of a SPAN (internal order), from the skeleton:
        CoalescedAccessor::diagBlockOffset (Accessor.h:75-85)"""
        sk = self.skel()
        lump = int(sk["spanToLump"][span])
        idx = span - int(sk["lumpToSpan"][lump])
        off = int(sk["chainData"][int(sk["chainColPtr"][lump]) + idx]) + int(sk["spanOffsetInLump"][span])
        return off, int(sk["lumpStart"][lump + 1] - sk["lumpStart"][lump])

    def deviceAccessor(self):
        """8 device addresses (spanStart, spanToLump, lumpStart, spanOffsetInLump, chainColPtr,
        chainRowSpan, chainData, permutation) usable from a caller's HIP kernel"""
        arr = (_I64P * 8)()
        _check(self._lib.bsp_device_accessor(self._h, arr))
        return [ctypes.cast(a, ctypes.c_void_p).value for a in arr]

    # ---- numeric ------------------------------------------------------------------------
    def setStream(self, stream):
        """stream: torch.cuda.Stream, raw hipStream_t address, or None (default stream)"""
        addr = 0 if stream is None else getattr(stream, "cuda_stream", stream)
        self._lib.bsp_set_stream(self._h, ctypes.c_void_p(addr))

    def _check_data(self, t):
        if t.numel() != self.dataSize():
            raise ValueError("data has %d elements, factor needs %d" % (t.numel(), self.dataSize()))

    def factor(self, data):
        """Solver::factor<T> / factor<std::vector<T*>> (list/tuple of tensors = batch)"""
        if isinstance(data, (list, tuple)):
            sfx = _suffix(data[0])
            for t in data:
                self._check_data(t)
            ptrs = (ctypes.c_void_p * len(data))(*[_ptr_of(t) for t in data])
            _check(getattr(self._lib, "bsp_factor_batched_" + sfx)(self._h, ptrs,
                                                                   ctypes.c_int32(len(data))))
        else:
            self._check_data(data)
            _check(getattr(self._lib, "bsp_factor_" + _suffix(data))(
                self._h, ctypes.c_void_p(_ptr_of(data))))

    def factorPerOp(self, data):
        """TESTING: factor() through the per-op NumericCtx boundary in the reference's call order"""
        self._check_data(data)
        _check(getattr(self._lib, "bsp_factor_per_op_" + _suffix(data))(
            self._h, ctypes.c_void_p(_ptr_of(data))))

    def forcePerOp(self, on=True):
        """TESTING: route factor / solve* / addMvFrom through the reference's per-op NumericCtx /
        SolveCtx boundary (bsp_force_per_op); usable as a context manager"""
        _check(self._lib.bsp_force_per_op(self._h, ctypes.c_int32(1 if on else 0)))
        solver = self

        class _Scope:
            def __enter__(self_inner):
                return solver

            def __exit__(self_inner, *exc):
                _check(solver._lib.bsp_force_per_op(solver._h, ctypes.c_int32(0)))
                return False
        return _Scope()

    def _testSetFault(self, kind):
        """TESTING, fault injection (bsp_test_set_fault): 1 = factor() skips the sparse-elimination
        update, 0 = off"""
        _check(self._lib.bsp_test_set_fault(self._h, ctypes.c_int32(kind)))

    def collectOpStats(self, on=True):
        """Solver::enableStats + per-op samples of the per-op boundary (bsp_collect_op_stats)"""
        _check(self._lib.bsp_collect_op_stats(self._h, ctypes.c_int32(1 if on else 0)))

    def opStats(self):
        """{"potrf": (n,2) [n, s], "trsm": (n,3) [n, k, s], "syge": (n,4) [m, n, k, s],
        "asmbl": (n,3) [blockRows, blockCols, s]} collected since collectOpStats(True)"""
        out = {}
        for which, (name, nsz) in enumerate((("potrf", 1), ("trsm", 2), ("syge", 3), ("asmbl", 2))):
            cnt = ctypes.c_int64(0)
            _check(self._lib.bsp_read_op_stats(self._h, ctypes.c_int32(which), None, ctypes.c_int64(0),
                                               ctypes.byref(cnt)))
            buf = np.zeros((cnt.value, 4))
            if cnt.value:
                _check(self._lib.bsp_read_op_stats(
                    self._h, ctypes.c_int32(which), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                    ctypes.c_int64(cnt.value), ctypes.byref(cnt)))
            out[name] = np.concatenate([buf[:, :nsz], buf[:, 3:4]], axis=1)
        return out

    def _factor_partial_batched(self, which, data, span_index):
        for t in data:
            self._check_data(t)
        ptrs = (ctypes.c_void_p * len(data))(*[_ptr_of(t) for t in data])
        _check(getattr(self._lib, "bsp_factor_partial_batched_" + _suffix(data[0]))(
            self._h, ptrs, ctypes.c_int32(len(data)), ctypes.c_int64(span_index), ctypes.c_int32(which)))

    def factorUpTo(self, data, span_index):
        """Solver::factorUpTo<T> / <std::vector<T*>> (list/tuple of tensors = batch)"""
        if isinstance(data, (list, tuple)):
            return self._factor_partial_batched(0, data, span_index)
        self._check_data(data)
        _check(getattr(self._lib, "bsp_factor_up_to_" + _suffix(data))(
            self._h, ctypes.c_void_p(_ptr_of(data)), ctypes.c_int64(span_index)))

    def factorFrom(self, data, span_index):
        if isinstance(data, (list, tuple)):
            return self._factor_partial_batched(1, data, span_index)
        self._check_data(data)
        _check(getattr(self._lib, "bsp_factor_from_" + _suffix(data))(
            self._h, ctypes.c_void_p(_ptr_of(data)), ctypes.c_int64(span_index)))

    def doElimination(self, data, elim_range_index):
        """TESTING hook: numCtx->doElimination(internalGetElimCtx(i), ...)"""
        self._check_data(data)
        _check(getattr(self._lib, "bsp_do_elimination_" + _suffix(data))(
            self._h, ctypes.c_void_p(_ptr_of(data)), ctypes.c_int64(elim_range_index)))

    def _solve(self, name, mat, vec, stride, nrhs):
        if stride is None:
            stride = self.order()
        if isinstance(mat, (list, tuple)):  # batch: one factored matrix and one vector block each
            assert isinstance(vec, (list, tuple)) and len(vec) == len(mat)
            for t in mat:
                self._check_data(t)
            which = {"bsp_solve_": 0, "bsp_solve_l_": 1, "bsp_solve_lt_": 2}[name]
            mats = (ctypes.c_void_p * len(mat))(*[_ptr_of(t) for t in mat])
            vecs = (ctypes.c_void_p * len(vec))(*[_ptr_of(t) for t in vec])
            _check(getattr(self._lib, "bsp_solve_batched_" + _suffix(mat[0]))(
                self._h, mats, vecs, ctypes.c_int32(len(mat)), ctypes.c_int64(stride),
                ctypes.c_int32(nrhs), ctypes.c_int32(which)))
            return
        self._check_data(mat)
        _check(getattr(self._lib, name + _suffix(mat))(
            self._h, ctypes.c_void_p(_ptr_of(mat)), ctypes.c_void_p(_ptr_of(vec)),
            ctypes.c_int64(stride), ctypes.c_int32(nrhs)))

    def solve(self, mat, vec, stride=None, nRHS=1):
        self._solve("bsp_solve_", mat, vec, stride, nRHS)

    def solveL(self, mat, vec, stride=None, nRHS=1):
        self._solve("bsp_solve_l_", mat, vec, stride, nRHS)

    def solveLt(self, mat, vec, stride=None, nRHS=1):
        self._solve("bsp_solve_lt_", mat, vec, stride, nRHS)

    def _solve_partial(self, which, mat, span_index, vec, stride, nrhs):
        if stride is None:
            stride = self.order()
        if isinstance(mat, (list, tuple)):  # batch
            assert isinstance(vec, (list, tuple)) and len(vec) == len(mat)
            for t in mat:
                self._check_data(t)
            mats = (ctypes.c_void_p * len(mat))(*[_ptr_of(t) for t in mat])
            vecs = (ctypes.c_void_p * len(vec))(*[_ptr_of(t) for t in vec])
            _check(getattr(self._lib, "bsp_solve_partial_batched_" + _suffix(mat[0]))(
                self._h, mats, vecs, ctypes.c_int32(len(mat)), ctypes.c_int64(stride),
                ctypes.c_int32(nrhs), ctypes.c_int32(which), ctypes.c_int64(span_index)))
            return
        self._check_data(mat)
        _check(getattr(self._lib, "bsp_solve_partial_" + _suffix(mat))(
            self._h, ctypes.c_void_p(_ptr_of(mat)), ctypes.c_void_p(_ptr_of(vec)),
            ctypes.c_int64(stride), ctypes.c_int32(nrhs), ctypes.c_int32(which),
            ctypes.c_int64(span_index)))

    def solveLUpTo(self, mat, span_index, vec, stride=None, nRHS=1):
        self._solve_partial(0, mat, span_index, vec, stride, nRHS)

    def solveLtUpTo(self, mat, span_index, vec, stride=None, nRHS=1):
        self._solve_partial(1, mat, span_index, vec, stride, nRHS)

    def solveLFrom(self, mat, span_index, vec, stride=None, nRHS=1):
        self._solve_partial(2, mat, span_index, vec, stride, nRHS)

    def solveLtFrom(self, mat, span_index, vec, stride=None, nRHS=1):
        self._solve_partial(3, mat, span_index, vec, stride, nRHS)

    def addMvFrom(self, mat, span_index, vec_in, in_stride, vec_out, out_stride, nRHS=1, alpha=1.0):
        """Solver::addMvFrom: out += alpha * A * in on the block from span_index on"""
        self._check_data(mat)
        sfx = _suffix(mat)
        a = ctypes.c_double(alpha) if sfx == "f64" else ctypes.c_float(alpha)
        _check(getattr(self._lib, "bsp_add_mv_from_" + sfx)(
            self._h, ctypes.c_void_p(_ptr_of(mat)), ctypes.c_int64(span_index),
            ctypes.c_void_p(_ptr_of(vec_in)), ctypes.c_int64(in_stride),
            ctypes.c_void_p(_ptr_of(vec_out)), ctypes.c_int64(out_stride), ctypes.c_int32(nRHS), a))

    def pseudoFactorFrom(self, data, span_index):
        """Solver::pseudoFactorFrom: per-span diagonal Cholesky + division of the rows below"""
        self._check_data(data)
        _check(getattr(self._lib, "bsp_pseudo_factor_from_" + _suffix(data))(
            self._h, ctypes.c_void_p(_ptr_of(data)), ctypes.c_int64(span_index)))

    # ---- measurement --------------------------------------------------------------------
    def factorFlops(self):
        return float(self._lib.bsp_factor_flops(self._h))

    def planLevels(self):
        """level table of the factor plan (bsp_plan_levels): int64 array [levels, 8]"""
        self._lib.bsp_plan_levels.restype = ctypes.c_int64
        n = self._lib.bsp_plan_levels(self._h, None, ctypes.c_int64(0))
        if n < 0:
            raise RuntimeError(_lib.load().bsp_last_error().decode("utf-8", "replace"))
        out = np.zeros(n, dtype=np.int64)
        self._lib.bsp_plan_levels(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int64(n))
        return out.reshape(-1, 8)

    def planStats(self):
        st = _CPlanStats()
        _check(self._lib.bsp_plan_stats_full(self._h, ctypes.byref(st)))
        return {n: getattr(st, n) for n, _ in _CPlanStats._fields_}

    def runCounters(self):
        """what the calls on this solver actually ran (bsp_run_counters_get): persistent sweeps
        launched / timed out, split-K lists used, sub-batches enqueued, lookahead forks"""
        st = _CRunCounters()
        _check(self._lib.bsp_run_counters_get(self._h, ctypes.byref(st)))
        return {n: getattr(st, n) for n, _ in _CRunCounters._fields_}

    def factorProfiled(self, data, in_situ=False, busy=False):
        """one factor() with every launch bracketed by HIP events; returns {kernel class: (total
        ms, launches)}.  in_situ=False: launches serialised on the execution stream (isolated
        kernel times); in_situ=True: the real two-stream schedule, each launch timed on the stream
        it runs on; busy=True adds, per class, the time during which at least one of its launches
        was running (launches on different streams overlap)"""
        self._check_data(data)
        ms = (ctypes.c_double * 6)()
        ln = (ctypes.c_int64 * 6)()
        if in_situ and busy:
            bs = (ctypes.c_double * 6)()
            _check(self._lib.bsp_factor_profiled_busy_f64(self._h, ctypes.c_void_p(_ptr_of(data)), ms, ln, bs))
            return {k: (ms[i], ln[i], bs[i]) for i, k in enumerate(PROF_KINDS)}
        fn = self._lib.bsp_factor_profiled_insitu_f64 if in_situ else self._lib.bsp_factor_profiled_f64
        _check(fn(self._h, ctypes.c_void_p(_ptr_of(data)), ms, ln))
        return {k: (ms[i], ln[i]) for i, k in enumerate(PROF_KINDS)}

    # ---- host helpers on the skeleton (CoalescedBlockMatrix.cpp:124-187) ------------------
    def densify(self, data, fill_upper_half=False, start_span_index=0):
        """dense numpy copy of host `data` laid out by this solver's skeleton"""
        sk = self.skel()
        data = np.asarray(data)
        off = int(sk["spanStart"][start_span_index])
        n = self.order() - off
        dense = np.zeros((n, n), dtype=data.dtype)
        ls, ss = sk["lumpStart"], sk["spanStart"]
        for l in range(int(sk["spanToLump"][start_span_index]), self.numLumps()):
            w = int(ls[l + 1] - ls[l])
            for c in range(int(sk["chainColPtr"][l]), int(sk["chainColPtr"][l + 1])):
                sp = int(sk["chainRowSpan"][c])
                r0, rows = int(ss[sp]) - off, int(ss[sp + 1] - ss[sp])
                d0 = int(sk["chainData"][c])
                dense[r0:r0 + rows, int(ls[l]) - off:int(ls[l]) - off + w] = \
                    data[d0:d0 + rows * w].reshape(rows, w)
        if fill_upper_half:
            dense = np.tril(dense) + np.tril(dense, -1).T
        return dense

    def lowerMask(self):
        """boolean mask over the numeric data: False on the strictly-upper entries of the square
        diagonal blocks (stored but meaningless, CoalescedBlockMatrix.h:23-37), True elsewhere"""
        sk = self.skel()
        ls = sk["lumpStart"]
        w = (ls[1:] - ls[:-1]).astype(np.int64)
        d0 = sk["chainData"][sk["chainColPtr"][:-1]]
        mask = np.ones(self.dataSize(), dtype=bool)
        for width in np.unique(w):
            if width < 2:
                continue
            i, j = np.triu_indices(int(width), 1)
            upper = (i * width + j).astype(np.int64)
            starts = d0[w == width]
            mask[(starts[:, None] + upper[None, :]).ravel()] = False
        return mask

    def damp(self, data, alpha, beta):
        """diagonal <- diagonal * (1 + alpha) + beta, in place on a host numpy array"""
        sk = self.skel()
        ls = sk["lumpStart"]
        w = (ls[1:] - ls[:-1]).astype(np.int64)
        d0 = sk["chainData"][sk["chainColPtr"][:-1]]
        idx = np.concatenate([d0[i] + np.arange(w[i]) * (w[i] + 1) for i in range(len(w))])
        data[idx] = data[idx] * (1 + alpha) + beta
        return data


def probe_mfma_f64_tflops():
    """sustained fp64 MFMA rate of the current GPU (register-only probe kernel), TFLOP/s"""
    out = ctypes.c_double(0)
    _check(_lib.load().bsp_probe_mfma_f64(ctypes.byref(out)))
    return out.value


def debug_read_trace(max_records=8192):
    """in-situ kernel clock records (library built with BSP_KTRACE=1), array [n, 8]"""
    out = np.zeros((max_records, 8), dtype=np.int64)
    n = ctypes.c_int(0)
    _check(_lib.load().bsp_debug_read_trace(out.ctypes.data_as(ctypes.c_void_p), max_records,
                                            ctypes.byref(n)))
    return out[:n.value]


def debug_read_extents(max_launches=2048):
    """per chain-step launch [first start, last start, last end, workgroup-0 end] (trace builds)"""
    out = np.zeros((max_launches, 4), dtype=np.uint64)
    n = ctypes.c_int(0)
    _check(_lib.load().bsp_debug_read_extents(out.ctypes.data_as(ctypes.c_void_p), max_launches,
                                              ctypes.byref(n)))
    return out[:n.value]


def create_solver(settings: Optional[Settings], param_sizes, ss: SparseStructure,
                  sparse_elim_ranges=(), elim_last_ids=()) -> Solver:
    """createSolver (Solver.h:235-237): symbolic analysis on the host; never touches the GPU."""
    lib = _lib.load()
    st = settings or Settings()
    cs = _CSettings(int(st.findSparseEliminationRanges), int(st.numThreads), int(st.backend),
                    int(st.addFillPolicy), None)
    model = None
    if st.computationModel is not None:
        model = (ctypes.c_double * 20)(*[float(x) for x in st.computationModel])
        cs.computation_model = ctypes.cast(model, ctypes.POINTER(ctypes.c_double))
    ps = _i64(param_sizes)
    if len(ps) != ss.order():
        raise ValueError("param_sizes and sparse structure disagree on the number of params")
    er, el = _i64(sparse_elim_ranges), _i64(sorted(elim_last_ids))
    h = ctypes.c_void_p()
    ho = None
    if st.hipOptions:
        ho = _CHipOptions()
        lib.bsp_hip_options_default(ctypes.byref(ho))
        for k, v in st.hipOptions.items():
            if k in _HIP_OPTION_INTS:
                setattr(ho, k, int(v))
            elif k in _HIP_OPTION_REALS:
                setattr(ho, k, float(v))
            else:
                raise ValueError("unknown backend option %r (bsp_hip_options has: %s)" %
                                 (k, ", ".join(_HIP_OPTION_INTS + _HIP_OPTION_REALS)))
    _check(lib.bsp_create_solver_opts(
        ctypes.byref(cs), ctypes.byref(ho) if ho is not None else None,
        ctypes.c_int64(len(ps)), ps.ctypes.data_as(_I64P),
        ss.ptrs.ctypes.data_as(_I64P), ss.inds.ctypes.data_as(_I64P), ctypes.c_int64(len(er)),
        er.ctypes.data_as(_I64P), ctypes.c_int64(len(el)), el.ctypes.data_as(_I64P),
        ctypes.byref(h)))
    return Solver(h)
