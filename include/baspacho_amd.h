/*
 * baspacho_amd.h -- C ABI of the MI355X-native supernodal sparse Cholesky.
 *
 * Plain C boundary (pointers + sizes, no C++/torch types) that a foreign-function binding
 * (ctypes, pybind, cgo, JNI ...) binds instead of the reference's C++ classes.  Each entry
 * point names the reference interface it stands for (paths relative to
 * /root/reference/baspacho/baspacho/).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; bsp_last_error() returns the
 *     message of the last failure on the calling thread (reference: BASPACHO_CHECK_* throw
 *     std::runtime_error, DebugMacros.h:17-50 / Utils.cpp:33-37);
 *   - index arrays are int64_t as in the reference (CoalescedBlockMatrix.h:88-110);
 *   - numeric data pointers are DEVICE pointers (Solver.h:184-188: GPU engines expect device
 *     memory); the caller owns them; factor works in place;
 *   - `stream` is a hipStream_t (NULL = default stream);
 *   - one factor()/solve() at a time per solver (Solver is not re-entrant).
 */
#ifndef BASPACHO_AMD_H_
#define BASPACHO_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsp_solver bsp_solver;

/* Solver.h:189-218  (BackendType, AddFillPolicy, Settings) */
enum { BSP_BACKEND_REF = 0, BSP_BACKEND_FAST = 1, BSP_BACKEND_CUDA = 2, BSP_BACKEND_HIP = 3 };
enum {
  BSP_FILL_COMPLETE = 0,
  BSP_FILL_FOR_AUTO_ELIMS = 1,
  BSP_FILL_FOR_GIVEN_ELIMS = 2,
  BSP_FILL_NONE = 3
};

typedef struct bsp_settings {
  int32_t find_sparse_elimination_ranges; /* default 1 */
  int32_t num_threads;                    /* ignored by the HIP engine */
  int32_t backend;                        /* BSP_BACKEND_HIP (CUDA accepted as alias) */
  int32_t add_fill_policy;                /* BSP_FILL_COMPLETE */
  /* optional cost model of the supernode-merge heuristic (ComputationModel.h:56-109):
     potrf[4], trsm[6], syge[6], asmbl[4]; NULL = built-in MI355X model */
  const double* computation_model;
} bsp_settings;

/* Extension (no counterpart in the reference): the schedule switches of the MI355X backend
   (csrc/backend_options.h).  Every field: a negative value (NaN for the doubles) = the library's
   default; a zeroed struct is NOT "all defaults" -- start from bsp_hip_options_default().  The
   environment variables of the same names (BSP_TAIL_BLOCKS, BSP_SOLVE_SWEEP, ...) still override what
   is passed here: they exist for A/B scripts. */
typedef struct bsp_hip_options {
  int32_t lookahead;         /* 0: lookahead units in line */
  int32_t due_stream;        /* 0: one auxiliary stream */
  int32_t split_k;           /* 0: no split-K tile lists */
  int32_t gather_max_pairs;  /* pairs per sparse-elimination gather item */
  int32_t gather_overlap;    /* 1: gather chunks beside the dense chain (measured: no gain) */
  int32_t sub_batch_min;     /* batches of at least this many as concurrent sub-batches; 0: never */
  int32_t sub_batches;       /* ... this many parts */
  int32_t tail_blocks;       /* outer blocks of a wide root lump for the persistent tail launch; 0: off */
  int32_t lazy_plan;         /* 1: no eager device plan when the solver is created */
  int32_t block_solve;       /* 0: wide lumps solved panel by panel */
  int32_t solve_inv;         /* 0: substitution instead of inverted diagonal blocks */
  int32_t solve_sweep;       /* 0: no persistent solve sweeps */
  int32_t sweep_min_width;   /* narrowest run of columns a sweep takes */
  int32_t solve_wide;        /* 0: no right-hand-sides-across-the-lanes backward elimination pass */
  int32_t chain_contraction; /* 0: no contraction of pivot chains before the ordering */
  int32_t dense_merge;       /* 0: no "rows >= 90 % of the parent's column" merge rule */
  int32_t expected_batch;    /* matrices per factor() call the supernode-merge model plans for (default 1) */
  double lookahead_min_gf;   /* GF per fork below which lookahead units stay in line */
  double bulk_ahead;         /* share of the next block's chain handed out as optional lookahead units */
  double level_cost_us;      /* supernode merges: cost of one level on the critical path (default 28 us) */
} bsp_hip_options;
void bsp_hip_options_default(bsp_hip_options* out);

const char* bsp_last_error(void);
const char* bsp_version(void);

/* createSolver(settings, paramSizes, SparseStructure{ptrs,inds}, sparseElimRanges, elimLastIds)
   Solver.h:235-237, Solver.cpp:611-752.  ptrs/inds: block CSR of the LOWER triangle incl.
   diagonal.  Symbolic analysis only: does not touch the GPU. */
int bsp_create_solver(const bsp_settings* settings, int64_t num_params, const int64_t* param_sizes,
                      const int64_t* ptrs, const int64_t* inds, int64_t num_elim_ranges,
                      const int64_t* elim_ranges, int64_t num_elim_last, const int64_t* elim_last,
                      bsp_solver** out);

/* ... the same with the backend's switches (NULL = bsp_create_solver) */
int bsp_create_solver_opts(const bsp_settings* settings, const bsp_hip_options* options,
                           int64_t num_params, const int64_t* param_sizes, const int64_t* ptrs,
                           const int64_t* inds, int64_t num_elim_ranges, const int64_t* elim_ranges,
                           int64_t num_elim_last, const int64_t* elim_last, bsp_solver** out);

/* Solver::Solver(CoalescedBlockMatrixSkel&&, sparseElimRanges, permutation, ops)
   Solver.h:37-38 -- solver from a RAW skeleton, as the reference's tests build it
   (tests/FactorTest.cpp:43-65).  col_ptr/row_ind: per-lump sorted row spans (csc). */
int bsp_create_solver_from_skeleton(int64_t num_spans, const int64_t* span_start,
                                    int64_t num_lumps, const int64_t* lump_to_span,
                                    const int64_t* col_ptr, const int64_t* row_ind,
                                    int64_t num_elim_ranges, const int64_t* elim_ranges,
                                    bsp_solver** out);

void bsp_destroy_solver(bsp_solver* s);

/* Solver::order / dataSize / canFactorUpToSpan, skel().numSpans()/numLumps()  Solver.h:111-131 */
int64_t bsp_order(const bsp_solver* s);
int64_t bsp_data_size(const bsp_solver* s);
int64_t bsp_num_spans(const bsp_solver* s);
int64_t bsp_num_lumps(const bsp_solver* s);
int64_t bsp_can_factor_up_to_span(const bsp_solver* s);
int64_t bsp_span_vector_offset(const bsp_solver* s, int64_t span);
int64_t bsp_span_matrix_offset(const bsp_solver* s, int64_t span, int64_t* out);

/* Skeleton arrays (CoalescedBlockMatrix.h:88-110); pointer stays valid while the solver lives */
enum {
  BSP_SKEL_SPAN_START = 0,
  BSP_SKEL_SPAN_TO_LUMP = 1,
  BSP_SKEL_LUMP_START = 2,
  BSP_SKEL_LUMP_TO_SPAN = 3,
  BSP_SKEL_SPAN_OFFSET_IN_LUMP = 4,
  BSP_SKEL_CHAIN_COL_PTR = 5,
  BSP_SKEL_CHAIN_ROW_SPAN = 6,
  BSP_SKEL_CHAIN_DATA = 7,
  BSP_SKEL_CHAIN_ROWS_TILL_END = 8,
  BSP_SKEL_BOARD_COL_PTR = 9,
  BSP_SKEL_BOARD_ROW_LUMP = 10,
  BSP_SKEL_BOARD_CHAIN_COL_ORD = 11,
  BSP_SKEL_BOARD_ROW_PTR = 12,
  BSP_SKEL_BOARD_COL_LUMP = 13,
  BSP_SKEL_BOARD_COL_ORD = 14,
  BSP_SKEL_PARAM_TO_SPAN = 15,     /* Solver::paramToSpan()            Solver.h:136-137 */
  BSP_SKEL_SPARSE_ELIM_RANGES = 16 /* Solver::sparseEliminationRanges  Solver.h:133-134 */
};
int bsp_skeleton_array(const bsp_solver* s, int which, const int64_t** data, int64_t* len);

/* PermutedCoalescedAccessor::blockOffset / diagBlockOffset  Accessor.h:145-166
   (param indices in USER order; flipped = stored block is the transpose).  Fails when the
   block is not part of the factor structure. */
int bsp_block_offset(const bsp_solver* s, int64_t row_param, int64_t col_param, int64_t* offset,
                     int64_t* stride, int32_t* flipped);
int bsp_diag_block_offset(const bsp_solver* s, int64_t param, int64_t* offset, int64_t* stride);

/* Solver::deviceAccessor()  Solver.h:47-48 / MatOpsCuda.cu:85-92: the 8 device arrays
   (spanStart, spanToLump, lumpStart, spanOffsetInLump, chainColPtr, chainRowSpan, chainData,
   permutation), in that order, for use inside a caller's HIP kernel. */
int bsp_device_accessor(bsp_solver* s, const int64_t* out_device_arrays[8]);

void bsp_set_stream(bsp_solver* s, void* stream);

/* Solver::factor<double|float>  Solver.h:60-61, Solver.cpp:149-151 */
int bsp_factor_f64(bsp_solver* s, double* dev_data);
int bsp_factor_f32(bsp_solver* s, float* dev_data);
/* Solver::factor<std::vector<T*>>  (batched, identical structure)  Solver.cpp:459-460;
   dev_ptrs is a HOST array of `batch` device pointers */
int bsp_factor_batched_f64(bsp_solver* s, double* const* dev_ptrs, int32_t batch);
int bsp_factor_batched_f32(bsp_solver* s, float* const* dev_ptrs, int32_t batch);
/* Solver::factorUpTo / factorFrom  Solver.h:75-76, 96-97 */
int bsp_factor_up_to_f64(bsp_solver* s, double* dev_data, int64_t span_index);
int bsp_factor_from_f64(bsp_solver* s, double* dev_data, int64_t span_index);
int bsp_factor_up_to_f32(bsp_solver* s, float* dev_data, int64_t span_index);
int bsp_factor_from_f32(bsp_solver* s, float* dev_data, int64_t span_index);
/* TESTING: Solver::factor driven op by op through the reference's NumericCtx boundary
   (potrf / trsm / saveSyrkGemm / prepareAssemble / assemble / doElimination, MatOps.h:113-136)
   in the reference's call order (Solver.cpp:164-219) instead of the fused path */
int bsp_factor_per_op_f64(bsp_solver* s, double* dev_data);
int bsp_factor_per_op_f32(bsp_solver* s, float* dev_data);
/* Solver::enableStats + the per-op stat callbacks of the reference (Utils.h:100-119) as bench -Z
   uses them (benchmarking/Bench.cpp:72-124): while on, every potrf / trsm / saveSyrkGemm / assemble
   call of the per-op boundary is timed (HIP events around its launches) and kept as one sample
   {size0, size1, size2, seconds}: which = 0 potrf {n}, 1 trsm {n, k}, 2 syrk/gemm {m, n, k},
   3 assemble {blockRows, blockCols}.  bsp_read_op_stats copies up to `capacity` samples (4 doubles
   each) and reports how many exist.  Input of the cost-model fit (tools/fit_computation_model.py,
   examples/OptimizeCompModel.cpp:64-275 in the reference). */
int bsp_collect_op_stats(bsp_solver* s, int32_t on);
int bsp_read_op_stats(bsp_solver* s, int32_t which, double* out, int64_t capacity, int64_t* count);
/* TESTING: while on, factor / solve* / addMvFrom of this solver are driven op by op through the
   reference's NumericCtx / SolveCtx boundary (MatOps.h:113-184) in the reference's call order
   (Solver.cpp:164-219, 270-397, 400-449) -- sparseElimSolveL/Lt, symm, solveL, gemv, assembleVec,
   solveLt, gemvT, assembleVecT, fragmentedMV/SolveL/SolveLt -- instead of the fused paths */
int bsp_force_per_op(bsp_solver* s, int32_t on);
/* Level table of the full-range factor plan (host only; the level schedule replaces the host-serial
 * per-lump loop of Solver.cpp:198-218): 8 values per level -- {sparse-elimination range or -1, panels,
 * widest panel, most rows below a panel, trsm row tiles, update tiles, lookahead tiles, rows below
 * summed over the panels}.  Returns the number of values (call with out = NULL to size), -1 on error. */
int64_t bsp_plan_levels(bsp_solver* s, int64_t* out, int64_t capacity);
/* TESTING hook of the reference: numCtx->doElimination(solver.internalGetElimCtx(i), ...)
   (Solver.h:139-145, tests/FactorTest.cpp:158-160) */
int bsp_do_elimination_f64(bsp_solver* s, double* dev_data, int64_t elim_range_index);
int bsp_do_elimination_f32(bsp_solver* s, float* dev_data, int64_t elim_range_index);

/* Solver::solve / solveL / solveLt  Solver.h:64-73 (vectors in internal order, column-major
   order x nRHS with leading dimension `stride`) */
int bsp_solve_f64(bsp_solver* s, const double* dev_mat, double* dev_vec, int64_t stride,
                  int32_t nrhs);
int bsp_solve_l_f64(bsp_solver* s, const double* dev_mat, double* dev_vec, int64_t stride,
                    int32_t nrhs);
int bsp_solve_lt_f64(bsp_solver* s, const double* dev_mat, double* dev_vec, int64_t stride,
                     int32_t nrhs);
int bsp_solve_f32(bsp_solver* s, const float* dev_mat, float* dev_vec, int64_t stride,
                  int32_t nrhs);
int bsp_solve_l_f32(bsp_solver* s, const float* dev_mat, float* dev_vec, int64_t stride,
                    int32_t nrhs);
int bsp_solve_lt_f32(bsp_solver* s, const float* dev_mat, float* dev_vec, int64_t stride,
                     int32_t nrhs);
/* Solver::solveLUpTo / solveLtUpTo / solveLFrom / solveLtFrom  Solver.h:79-108 (span_index must
   be a lump boundary).  which: 0 = LUpTo, 1 = LtUpTo, 2 = LFrom, 3 = LtFrom */
int bsp_solve_partial_f64(bsp_solver* s, const double* dev_mat, double* dev_vec, int64_t stride,
                          int32_t nrhs, int32_t which, int64_t span_index);
int bsp_solve_partial_f32(bsp_solver* s, const float* dev_mat, float* dev_vec, int64_t stride,
                          int32_t nrhs, int32_t which, int64_t span_index);
/* Solver::addMvFrom  Solver.h:89-91: out += alpha * A * in on the symmetric block from span_index
   (a lump boundary) to the end; vectors hold `order` rows, column-major, nRHS columns */
int bsp_add_mv_from_f64(bsp_solver* s, const double* dev_mat, int64_t span_index,
                        const double* dev_in, int64_t in_stride, double* dev_out,
                        int64_t out_stride, int32_t nrhs, double alpha);
int bsp_add_mv_from_f32(bsp_solver* s, const float* dev_mat, int64_t span_index, const float* dev_in,
                        int64_t in_stride, float* dev_out, int64_t out_stride, int32_t nrhs,
                        float alpha);
/* Solver::pseudoFactorFrom  Solver.h:92-94: Cholesky of the diagonal block of every span from
   span_index on, the rows below it divided by the factor (any span width, MatOpsCuda.cu:188-233) */
int bsp_pseudo_factor_from_f64(bsp_solver* s, double* dev_data, int64_t span_index);
int bsp_pseudo_factor_from_f32(bsp_solver* s, float* dev_data, int64_t span_index);
/* Solver::solve<std::vector<T*>> etc. (Solver.h:64-73 with the batch types of MatOps.h:38-42):
   `batch` factored matrices of the same structure and one block of right-hand sides each
   (host arrays of device pointers).  which: 0 = solve, 1 = solveL, 2 = solveLt */
int bsp_solve_batched_f64(bsp_solver* s, const double* const* dev_mats, double* const* dev_vecs,
                          int32_t batch, int64_t stride, int32_t nrhs, int32_t which);
int bsp_solve_batched_f32(bsp_solver* s, const float* const* dev_mats, float* const* dev_vecs,
                          int32_t batch, int64_t stride, int32_t nrhs, int32_t which);
/* the batch forms of the partial operations (Solver.cpp:491-519 instantiates factorUpTo / factorFrom
   / solveLUpTo / solveLtUpTo for std::vector<T*>; solveLFrom / solveLtFrom come with them here).
   factor: which 0 = factorUpTo, 1 = factorFrom; solve: which as bsp_solve_partial_* */
int bsp_factor_partial_batched_f64(bsp_solver* s, double* const* dev_ptrs, int32_t batch,
                                   int64_t span_index, int32_t which);
int bsp_factor_partial_batched_f32(bsp_solver* s, float* const* dev_ptrs, int32_t batch,
                                   int64_t span_index, int32_t which);
int bsp_solve_partial_batched_f64(bsp_solver* s, const double* const* dev_mats,
                                  double* const* dev_vecs, int32_t batch, int64_t stride,
                                  int32_t nrhs, int32_t which, int64_t span_index);
int bsp_solve_partial_batched_f32(bsp_solver* s, const float* const* dev_mats, float* const* dev_vecs,
                                  int32_t batch, int64_t stride, int32_t nrhs, int32_t which,
                                  int64_t span_index);

/* ---- BAL caller pipeline on the device (benchmarking/BaAtLargeOptimizer.cpp:100-131 computeStep,
   with the 9-parameter cameras of a BAL file: Rodrigues rotation, translation, f, k1, k2).  All
   pointers are device memory.
   bsp_bal_linearize: reprojection residual res[2 nObs] and Jacobians Jc[nObs][2][9] (camera),
   Jp[nObs][2][3] (point) of every observation (obs_xy[nObs][2], cams[nCams][9], pts[nPts][3]).
   bsp_bal_fill_hessian: data += J^T J through Solver::deviceAccessor() -- accessor.diagBlock(pt),
   accessor.diagBlock(cam), accessor.block(cam, pt) per observation -- grad += J^T r (may be NULL),
   then the LM damping d <- d (1 + lambda) + 1e-3 lambda of every diagonal entry.  The caller zeroes
   data / grad first; parameters are numbered points first, cameras after (BaAtLargeBench.cpp:50-57).
   dbg (may be NULL): 7 int64 per observation = what the accessor returned on the device (block
   offset, stride, flipped; camera diag offset, stride; point diag offset, stride). */
int bsp_bal_linearize_f64(int64_t num_obs, const int64_t* obs_cam, const int64_t* obs_pt,
                          const double* obs_xy, const double* cams, const double* pts, double* res,
                          double* Jc, double* Jp, void* stream);
int bsp_bal_fill_hessian_f64(bsp_solver* s, int64_t num_pts, int64_t num_cams, int64_t num_obs,
                             const int64_t* obs_cam, const int64_t* obs_pt, const double* Jc,
                             const double* Jp, const double* res, double lambda, double* dev_data,
                             double* dev_grad, int64_t* dev_dbg, void* stream);
int bsp_bal_fill_hessian_f32(bsp_solver* s, int64_t num_pts, int64_t num_cams, int64_t num_obs,
                             const int64_t* obs_cam, const int64_t* obs_pt, const double* Jc,
                             const double* Jp, const double* res, float lambda, float* dev_data,
                             float* dev_grad, int64_t* dev_dbg, void* stream);

/* The same pipeline in the REFERENCE's parameterisation (benchmarking/BaAtLarge.h:56-150
   Cost::compute_residual, BaAtLargeOptimizer.cpp:24-52,100-131,176-183): cameras are 6-wide blocks,
   the tangent of a left SE(3) perturbation T_W_C <- exp(delta) T_W_C (translation first), calibration
   (f, k1, k2) fixed; Jc[nObs][2][6]; a point with camPt.z > 0.01 gives the residual (25, 0) and zero
   Jacobians.  cams[nCams][9] keeps the BAL storage (Rodrigues rotation, translation, f, k1, k2).
   The solver must have been created with camera blocks of size 6 (checked). */
int bsp_bal_linearize_se3_f64(int64_t num_obs, const int64_t* obs_cam, const int64_t* obs_pt,
                              const double* obs_xy, const double* cams, const double* pts,
                              double* res, double* Jc, double* Jp, void* stream);
int bsp_bal_fill_hessian_se3_f64(bsp_solver* s, int64_t num_pts, int64_t num_cams, int64_t num_obs,
                                 const int64_t* obs_cam, const int64_t* obs_pt, const double* Jc,
                                 const double* Jp, const double* res, double lambda, double* dev_data,
                                 double* dev_grad, void* stream);
int bsp_bal_fill_hessian_se3_f32(bsp_solver* s, int64_t num_pts, int64_t num_cams, int64_t num_obs,
                                 const int64_t* obs_cam, const int64_t* obs_pt, const double* Jc,
                                 const double* Jp, const double* res, float lambda, float* dev_data,
                                 float* dev_grad, void* stream);

/* ---- measurement helpers (no reference counterpart; Solver::printStats is the analogue) */
/* algorithmic flops of a full factor: sum over lumps n^3/3 + r n^2 + r^2 n */
double bsp_factor_flops(const bsp_solver* s);

typedef struct bsp_plan_stats {
  double flops, upd_elems, upd_flops, elim_pair_elems, elim_pair_flops, elim_col_elems,
      upd_flops_direct,        /* part of upd_flops launched by the one-panel-level kernels */
      elim_pair_operand_elems, /* values of both source blocks, summed over the pairs */
      elim_target_elems,       /* distinct target elements of the sparse-elimination update */
      trsm_flops, potrf_flops, /* dense panels: rowsBelow*nb^2 and nb^3/3 */
      trsm_flops_merged,       /* ... of which inside chain-step launches (trsm + update + potrf) */
      potrf_flops_fused;       /* ... of which inside the previous level's update launch */
  int64_t num_launches, num_levels, num_panels, num_segs, num_upd_tasks, num_trsm_tasks,
      chain_tab_entries, max_panels_in_level, num_atomic_upd_tasks,
      num_gather_groups, /* always 0 (kept for ABI stability) */
      num_fork_levels;   /* levels that hand lookahead units to the auxiliary streams */
  double deferred_flops; /* flops of those units; the lookahead schedule is used when they are worth
                            the forks (HipPlanHost::lookaheadPays) */
  double tail_upd_flops; /* update flops done inside the persistent tail launch (csrc/hip_tail_kernel.h);
                            not part of upd_flops */
  int64_t num_tail_panels;
} bsp_plan_stats;
int bsp_plan_stats_full(bsp_solver* s, bsp_plan_stats* out);

/* What the calls on this solver actually RAN (cumulative counters; no reference counterpart): a
   test can see that the path it means to test was taken, and a caller can poll for a watchdog
   report of the persistent solve sweeps (hip_sweep_kernels.h) without waiting for the next solve,
   which would throw it. */
typedef struct bsp_run_counters {
  int64_t sweep_launches;       /* persistent solve sweeps launched */
  int64_t sweep_timeouts;       /* ... that ran into their watchdog (reported by a later call) */
  int64_t split_lists_used;     /* update launches that took a split-K tile list */
  int64_t sub_batches_enqueued; /* sub-batches enqueued on a stream of their own */
  int64_t lookahead_forks;      /* lookahead launches handed to the auxiliary streams */
  int64_t sweeps_retired;       /* 1: a time-out retired the sweeps of this solver */
  int64_t sweep_error_pending;  /* 1: a time-out has been raised and not been reported yet */
  int64_t gather_chunks_overlapped; /* sparse-elimination gather chunks launched beside the dense chain */
  int64_t tail_launches;        /* persistent tail launches (csrc/hip_tail_kernel.h) */
  int64_t sweep_mfma_launches;  /* ... of sweep_launches: the matrix-core form (several right-hand sides) */
  int64_t solve_wide_launches;  /* backward elimination passes with the right-hand sides across the lanes */
  int64_t inv_reused;           /* backward passes that found the inverted diagonal blocks of their forward pass */
  int64_t potrf_folded_levels;  /* tree levels whose potrf launch was folded into their trsm launch */
} bsp_run_counters;
int bsp_run_counters_get(bsp_solver* s, bsp_run_counters* out);

/* kernel classes timed by bsp_factor_profiled_* (HIP events on the execution stream) */
enum {
  BSP_PROF_ELIM_FACTOR = 0,
  BSP_PROF_ELIM_UPDATE = 1,
  BSP_PROF_POTRF = 2,
  BSP_PROF_TRSM = 3,
  BSP_PROF_UPDATE = 4,       /* updateTile: task-list launches, the bulk of the flops */
  BSP_PROF_CHAIN_UPDATE = 5, /* update launches of one-panel levels (next potrf fused in) */
  BSP_PROF_NUM_KINDS = 6
};
int bsp_factor_profiled_f64(bsp_solver* s, double* dev_data, double ms[6], int64_t launches[6]);
/* the same with the real two-stream schedule left on: every launch timed on the stream it runs
   on, i.e. what a kernel takes beside the others (= what a rocprofv3 kernel trace shows) */
int bsp_factor_profiled_insitu_f64(bsp_solver* s, double* dev_data, double ms[6],
                                   int64_t launches[6]);
/* ... and, per class, the time during which at least one of its launches was running (launches of
   one class that go to different streams overlap: their durations add up to more than that) */
int bsp_factor_profiled_busy_f64(bsp_solver* s, double* dev_data, double ms[6], int64_t launches[6],
                                 double busy_ms[6]);

/* sustained v_mfma_f64_16x16x4_f64 rate of the current GPU (TFLOP/s), register-only probe */
int bsp_probe_mfma_f64(double* tflops);

/* Developer aid: in-situ kernel clock records of a library built with BSP_KTRACE=1 (8 values per
 * stamped kernel launch); a normal build reports 0 records. */
int bsp_debug_read_trace(long long* out, int max_records, int* n_records);
/* ... and, for builds with -DBSP_TRACE_TILE as well: per chain-step launch {first workgroup start,
 * last workgroup start, last workgroup end, workgroup-0 end} in 10 ns units; reading resets. */
int bsp_debug_read_extents(unsigned long long* out, int max_launches, int* n);

/* Symbolic plan as one flat int64 buffer, so that rank 0 can analyse once and broadcast it
   (RCCL) to the ranks that factor the other matrices of a batch. */
int bsp_plan_serialize(const bsp_solver* s, int64_t* buf, int64_t capacity, int64_t* needed);
int bsp_create_solver_from_plan(const int64_t* buf, int64_t len, bsp_solver** out);

/* Host-side block-pattern operations of the symbolic phase (SparseStructure.h:34-55), exposed
   for bindings and tests.  `arg`/`flag` per op:
     TRANSPOSE            -                                  SparseStructure::transpose
     CLEAR                flag = clearLower                  SparseStructure::clear
     SYM_PERMUTATION      arg = mapPerm[n], flag = lowerHalf SparseStructure::symmetricPermutation
     INDEP_ELIM_FILL      arg = {start, end}                 addIndependentEliminationFill
     FULL_ELIM_FILL       -                                  addFullEliminationFill
     FILL_REDUCING_PERM   -  (out_inds = perm[n], out_ptrs untouched)  fillReducingPermutation
     EXTRACT_RIGHT_BOTTOM arg = {start}                      extractRightBottom
   out_ptrs needs n+1 entries; the call fails if out_inds (capacity entries) is too small,
   reporting the needed size in *out_nnz. */
enum {
  BSP_SS_TRANSPOSE = 0,
  BSP_SS_CLEAR = 1,
  BSP_SS_SYM_PERMUTATION = 2,
  BSP_SS_INDEP_ELIM_FILL = 3,
  BSP_SS_FULL_ELIM_FILL = 4,
  BSP_SS_FILL_REDUCING_PERM = 5,
  BSP_SS_EXTRACT_RIGHT_BOTTOM = 6
};
int bsp_sparse_structure_op(int op, int64_t n, const int64_t* ptrs, const int64_t* inds,
                            const int64_t* arg, int64_t arg_len, int32_t flag, int64_t* out_ptrs,
                            int64_t* out_inds, int64_t capacity, int64_t* out_n, int64_t* out_nnz);

#ifdef __cplusplus
}
#endif
#endif /* BASPACHO_AMD_H_ */
