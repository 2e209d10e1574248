// C ABI facade (include/baspacho_amd.h) over the C++ Solver.  Exceptions never cross the
// boundary: they become a non-zero return code + bsp_last_error().
#include "../../include/baspacho_amd.h"

#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_set>

#include "bal_pipeline.h"
#include "../../include/baspacho_amd_testing.h"
#include "computation_model.h"
#include "hip_backend.h"
#include "solver.h"

using namespace BaSpaCho;

struct bsp_solver {
  SolverPtr solver;
  ComputationModel model;  // storage for a user-supplied model
  // BAL layout already validated for (numPts, numCams, camSize): the parameter sizes of a solver
  // never change, so the O(numPts + numCams) walk of checkBalLayout runs once, not per LM iteration
  int64_t balOk[3] = {-1, -1, -1};
};

static thread_local std::string g_lastError;

const char* bsp_last_error(void) { return g_lastError.c_str(); }
const char* bsp_version(void) { return "baspacho_amd 0.1 (gfx950)"; }

#define BSP_TRY try {
#define BSP_CATCH                      \
  }                                    \
  catch (const std::exception& e) {    \
    g_lastError = e.what();            \
    return 1;                          \
  }                                    \
  catch (...) {                        \
    g_lastError = "unknown exception"; \
    return 1;                          \
  }                                    \
  return 0;

void bsp_hip_options_default(bsp_hip_options* o) {
  o->lookahead = o->due_stream = o->split_k = o->gather_max_pairs = o->gather_overlap = o->sub_batch_min =
      o->sub_batches = o->tail_blocks = o->lazy_plan = o->block_solve = o->solve_inv = o->solve_sweep =
          o->sweep_min_width = o->solve_wide = o->chain_contraction = o->dense_merge = o->expected_batch = -1;
  o->lookahead_min_gf = o->bulk_ahead = o->level_cost_us = NAN;
}

int bsp_create_solver_opts(const bsp_settings* st, const bsp_hip_options* ho, int64_t numParams,
                           const int64_t* paramSizes, const int64_t* ptrs, const int64_t* inds,
                           int64_t numElimRanges, const int64_t* elimRanges, int64_t numElimLast,
                           const int64_t* elimLast, bsp_solver** out) {
  BSP_TRY
  BASPACHO_CHECK_NOTNULL(out);
  std::unique_ptr<bsp_solver> h(new bsp_solver);
  Settings settings;
  HipBackendOptions options;
  if (ho) {
    options.lookahead = ho->lookahead;
    options.dueStream = ho->due_stream;
    options.splitK = ho->split_k;
    options.gatherMaxPairs = ho->gather_max_pairs;
    options.gatherOverlap = ho->gather_overlap;
    options.subBatchMin = ho->sub_batch_min;
    options.subBatches = ho->sub_batches;
    options.tailBlocks = ho->tail_blocks;
    options.lazyPlan = ho->lazy_plan;
    options.blockSolve = ho->block_solve;
    options.solveInv = ho->solve_inv;
    options.solveSweep = ho->solve_sweep;
    options.sweepMinWidth = ho->sweep_min_width;
    options.solveWide = ho->solve_wide;
    options.chainContraction = ho->chain_contraction;
    options.denseMerge = ho->dense_merge;
    options.expectedBatch = ho->expected_batch;
    options.lookaheadMinGF = ho->lookahead_min_gf;
    options.bulkAhead = ho->bulk_ahead;
    options.levelCostUs = ho->level_cost_us;
    settings.hipOptions = &options;
  }
  if (st) {
    settings.findSparseEliminationRanges = st->find_sparse_elimination_ranges != 0;
    settings.numThreads = st->num_threads;
    settings.backend = (BackendType)st->backend;
    settings.addFillPolicy = (AddFillPolicy)st->add_fill_policy;
    if (st->computation_model) {
      const double* p = st->computation_model;
      std::copy(p, p + 4, h->model.potrfParams.begin());
      std::copy(p + 4, p + 10, h->model.trsmParams.begin());
      std::copy(p + 10, p + 16, h->model.sygeParams.begin());
      std::copy(p + 16, p + 20, h->model.asmblParams.begin());
      settings.computationModel = &h->model;
    }
  }
  std::vector<int64_t> sizes(paramSizes, paramSizes + numParams);
  SparseStructure ss(std::vector<int64_t>(ptrs, ptrs + numParams + 1),
                     std::vector<int64_t>(inds, inds + ptrs[numParams]));
  std::vector<int64_t> ranges(elimRanges, elimRanges + (elimRanges ? numElimRanges : 0));
  std::unordered_set<int64_t> last(elimLast, elimLast + (elimLast ? numElimLast : 0));
  h->solver = createSolver(settings, sizes, ss, ranges, last);
  *out = h.release();
  BSP_CATCH
}

int bsp_create_solver(const bsp_settings* st, int64_t numParams, const int64_t* paramSizes,
                      const int64_t* ptrs, const int64_t* inds, int64_t numElimRanges,
                      const int64_t* elimRanges, int64_t numElimLast, const int64_t* elimLast,
                      bsp_solver** out) {
  return bsp_create_solver_opts(st, nullptr, numParams, paramSizes, ptrs, inds, numElimRanges, elimRanges,
                                numElimLast, elimLast, out);
}

int bsp_create_solver_from_skeleton(int64_t numSpans, const int64_t* spanStart, int64_t numLumps,
                                    const int64_t* lumpToSpan, const int64_t* colPtr,
                                    const int64_t* rowInd, int64_t numElimRanges,
                                    const int64_t* elimRanges, bsp_solver** out) {
  BSP_TRY
  BASPACHO_CHECK_NOTNULL(out);
  std::vector<int64_t> ss(spanStart, spanStart + numSpans + 1);
  std::vector<int64_t> l2s(lumpToSpan, lumpToSpan + numLumps + 1);
  std::vector<int64_t> cp(colPtr, colPtr + numLumps + 1);
  std::vector<int64_t> ri(rowInd, rowInd + colPtr[numLumps]);
  std::vector<int64_t> ranges(elimRanges, elimRanges + (elimRanges ? numElimRanges : 0));
  CoalescedBlockMatrixSkel skel(ss, l2s, cp, ri);
  std::unique_ptr<bsp_solver> h(new bsp_solver);
  h->solver.reset(new Solver(std::move(skel), std::move(ranges), {}, hipOps()));
  *out = h.release();
  BSP_CATCH
}

void bsp_destroy_solver(bsp_solver* s) { delete s; }

int64_t bsp_order(const bsp_solver* s) { return s->solver->order(); }
int64_t bsp_data_size(const bsp_solver* s) { return s->solver->dataSize(); }
int64_t bsp_num_spans(const bsp_solver* s) { return s->solver->skel().numSpans(); }
int64_t bsp_num_lumps(const bsp_solver* s) { return s->solver->skel().numLumps(); }
int64_t bsp_can_factor_up_to_span(const bsp_solver* s) { return s->solver->canFactorUpToSpan(); }
int64_t bsp_span_vector_offset(const bsp_solver* s, int64_t span) {
  return s->solver->spanVectorOffset(span);
}
int64_t bsp_span_matrix_offset(const bsp_solver* s, int64_t span, int64_t* out) {
  BSP_TRY
  *out = s->solver->spanMatrixOffset(span);
  BSP_CATCH
}

int bsp_skeleton_array(const bsp_solver* s, int which, const int64_t** data, int64_t* len) {
  BSP_TRY
  const CoalescedBlockMatrixSkel& k = s->solver->skel();
  const std::vector<int64_t>* v = nullptr;
  switch (which) {
    case BSP_SKEL_SPAN_START: v = &k.spanStart; break;
    case BSP_SKEL_SPAN_TO_LUMP: v = &k.spanToLump; break;
    case BSP_SKEL_LUMP_START: v = &k.lumpStart; break;
    case BSP_SKEL_LUMP_TO_SPAN: v = &k.lumpToSpan; break;
    case BSP_SKEL_SPAN_OFFSET_IN_LUMP: v = &k.spanOffsetInLump; break;
    case BSP_SKEL_CHAIN_COL_PTR: v = &k.chainColPtr; break;
    case BSP_SKEL_CHAIN_ROW_SPAN: v = &k.chainRowSpan; break;
    case BSP_SKEL_CHAIN_DATA: v = &k.chainData; break;
    case BSP_SKEL_CHAIN_ROWS_TILL_END: v = &k.chainRowsTillEnd; break;
    case BSP_SKEL_BOARD_COL_PTR: v = &k.boardColPtr; break;
    case BSP_SKEL_BOARD_ROW_LUMP: v = &k.boardRowLump; break;
    case BSP_SKEL_BOARD_CHAIN_COL_ORD: v = &k.boardChainColOrd; break;
    case BSP_SKEL_BOARD_ROW_PTR: v = &k.boardRowPtr; break;
    case BSP_SKEL_BOARD_COL_LUMP: v = &k.boardColLump; break;
    case BSP_SKEL_BOARD_COL_ORD: v = &k.boardColOrd; break;
    case BSP_SKEL_PARAM_TO_SPAN: v = &s->solver->paramToSpan(); break;
    case BSP_SKEL_SPARSE_ELIM_RANGES: v = &s->solver->sparseEliminationRanges(); break;
    default: throw std::runtime_error("bsp_skeleton_array: unknown array id");
  }
  *data = v->data();
  *len = (int64_t)v->size();
  BSP_CATCH
}

int bsp_block_offset(const bsp_solver* s, int64_t rowParam, int64_t colParam, int64_t* offset,
                     int64_t* stride, int32_t* flipped) {
  BSP_TRY
  const int64_t n = s->solver->skel().numSpans();
  BASPACHO_CHECK(rowParam >= 0 && rowParam < n && colParam >= 0 && colParam < n);
  auto acc = s->solver->accessor();
  const int64_t pr = acc.permutation[rowParam], pc = acc.permutation[colParam];
  if (!acc.plainAcc.hasBlock(std::max(pr, pc), std::min(pr, pc))) {
    throw std::runtime_error("bsp_block_offset: block is not in the factor structure");
  }
  auto t = acc.blockOffset(rowParam, colParam);
  *offset = std::get<0>(t);
  *stride = std::get<1>(t);
  *flipped = std::get<2>(t) ? 1 : 0;
  BSP_CATCH
}

int bsp_diag_block_offset(const bsp_solver* s, int64_t param, int64_t* offset, int64_t* stride) {
  BSP_TRY
  BASPACHO_CHECK(param >= 0 && param < s->solver->skel().numSpans());
  auto p = s->solver->accessor().diagBlockOffset(param);
  *offset = p.first;
  *stride = p.second;
  BSP_CATCH
}

int bsp_device_accessor(bsp_solver* s, const int64_t* out[8]) {
  BSP_TRY
  PermutedCoalescedAccessor a = s->solver->deviceAccessor();
  out[0] = a.plainAcc.spanStart;
  out[1] = a.plainAcc.spanToLump;
  out[2] = a.plainAcc.lumpStart;
  out[3] = a.plainAcc.spanOffsetInLump;
  out[4] = a.plainAcc.chainColPtr;
  out[5] = a.plainAcc.chainRowSpan;
  out[6] = a.plainAcc.chainData;
  out[7] = a.permutation;
  BSP_CATCH
}

void bsp_set_stream(bsp_solver* s, void* stream) { s->solver->setStream(stream); }

int bsp_factor_f64(bsp_solver* s, double* d) {
  BSP_TRY
  s->solver->factor(d);
  BSP_CATCH
}
int bsp_factor_f32(bsp_solver* s, float* d) {
  BSP_TRY
  s->solver->factor(d);
  BSP_CATCH
}
int bsp_factor_batched_f64(bsp_solver* s, double* const* ptrs, int32_t batch) {
  BSP_TRY
  std::vector<double*> v(ptrs, ptrs + batch);
  s->solver->factor(&v);
  BSP_CATCH
}
int bsp_factor_batched_f32(bsp_solver* s, float* const* ptrs, int32_t batch) {
  BSP_TRY
  std::vector<float*> v(ptrs, ptrs + batch);
  s->solver->factor(&v);
  BSP_CATCH
}
int bsp_factor_up_to_f64(bsp_solver* s, double* d, int64_t span) {
  BSP_TRY
  s->solver->factorUpTo(d, span);
  BSP_CATCH
}
int bsp_factor_from_f64(bsp_solver* s, double* d, int64_t span) {
  BSP_TRY
  s->solver->factorFrom(d, span);
  BSP_CATCH
}
int bsp_factor_up_to_f32(bsp_solver* s, float* d, int64_t span) {
  BSP_TRY
  s->solver->factorUpTo(d, span);
  BSP_CATCH
}
int bsp_factor_from_f32(bsp_solver* s, float* d, int64_t span) {
  BSP_TRY
  s->solver->factorFrom(d, span);
  BSP_CATCH
}

template <typename T>
static void factorPerOp(bsp_solver* s, T* d) {
  SymbolicCtx& sym = s->solver->internalSymbolicContext();
  hipBackendForcePerOp(sym, true);
  try {
    s->solver->factor(d);
  } catch (...) {
    hipBackendForcePerOp(sym, false);
    throw;
  }
  hipBackendForcePerOp(sym, false);
}
int bsp_factor_per_op_f64(bsp_solver* s, double* d) {
  BSP_TRY
  factorPerOp(s, d);
  BSP_CATCH
}
int bsp_factor_per_op_f32(bsp_solver* s, float* d) {
  BSP_TRY
  factorPerOp(s, d);
  BSP_CATCH
}

int bsp_collect_op_stats(bsp_solver* s, int32_t on) {
  BSP_TRY
  SymbolicCtx& sym = s->solver->internalSymbolicContext();
  s->solver->enableStats(on != 0);
  s->solver->resetStats();
  for (OpStat* st : {&sym.potrfStat, &sym.trsmStat, &sym.sygeStat, &sym.asmblStat}) st->keepSamples = on != 0;
  BSP_CATCH
}
int bsp_read_op_stats(bsp_solver* s, int32_t which, double* out, int64_t capacity, int64_t* count) {
  BSP_TRY
  SymbolicCtx& sym = s->solver->internalSymbolicContext();
  const OpStat* sts[4] = {&sym.potrfStat, &sym.trsmStat, &sym.sygeStat, &sym.asmblStat};
  BASPACHO_CHECK(which >= 0 && which < 4);
  const auto& v = sts[which]->samples;
  *count = (int64_t)v.size();
  for (int64_t i = 0; i < std::min<int64_t>(capacity, (int64_t)v.size()); i++) {
    for (int j = 0; j < 4; j++) out[4 * i + j] = v[i][j];
  }
  BSP_CATCH
}

int bsp_force_per_op(bsp_solver* s, int32_t on) {
  BSP_TRY
  hipBackendForcePerOp(s->solver->internalSymbolicContext(), on != 0);
  BSP_CATCH
}

int bsp_test_read_sweep_trace(bsp_solver* s, long long* out, int32_t max_blocks, int32_t* n_blocks) {
  BSP_TRY
  *n_blocks = hipBackendReadSweepTrace(s->solver->internalSymbolicContext(), out, max_blocks);
  BSP_CATCH
}

int bsp_test_set_fault(bsp_solver* s, int32_t kind) {
  BSP_TRY
  hipBackendSetFault(s->solver->internalSymbolicContext(), kind);
  BSP_CATCH
}

template <typename T>
static void doElim(bsp_solver* s, T* d, int64_t idx) {
  const auto& ranges = s->solver->sparseEliminationRanges();
  BASPACHO_CHECK(idx >= 0 && idx + 1 < (int64_t)ranges.size());
  NumericCtxPtr<T> ctx = s->solver->internalSymbolicContext().createNumericCtx<T>(0, d);
  ctx->doElimination(s->solver->internalGetElimCtx((size_t)idx), d, ranges[idx], ranges[idx + 1]);
}
int bsp_do_elimination_f64(bsp_solver* s, double* d, int64_t idx) {
  BSP_TRY
  doElim(s, d, idx);
  BSP_CATCH
}
int bsp_do_elimination_f32(bsp_solver* s, float* d, int64_t idx) {
  BSP_TRY
  doElim(s, d, idx);
  BSP_CATCH
}

int bsp_solve_f64(bsp_solver* s, const double* m, double* v, int64_t stride, int32_t nrhs) {
  BSP_TRY
  s->solver->solve(m, v, stride, nrhs);
  BSP_CATCH
}
int bsp_solve_l_f64(bsp_solver* s, const double* m, double* v, int64_t stride, int32_t nrhs) {
  BSP_TRY
  s->solver->solveL(m, v, stride, nrhs);
  BSP_CATCH
}
int bsp_solve_lt_f64(bsp_solver* s, const double* m, double* v, int64_t stride, int32_t nrhs) {
  BSP_TRY
  s->solver->solveLt(m, v, stride, nrhs);
  BSP_CATCH
}
int bsp_solve_f32(bsp_solver* s, const float* m, float* v, int64_t stride, int32_t nrhs) {
  BSP_TRY
  s->solver->solve(m, v, stride, nrhs);
  BSP_CATCH
}
int bsp_solve_l_f32(bsp_solver* s, const float* m, float* v, int64_t stride, int32_t nrhs) {
  BSP_TRY
  s->solver->solveL(m, v, stride, nrhs);
  BSP_CATCH
}
int bsp_solve_lt_f32(bsp_solver* s, const float* m, float* v, int64_t stride, int32_t nrhs) {
  BSP_TRY
  s->solver->solveLt(m, v, stride, nrhs);
  BSP_CATCH
}

// partial solves: which = 0 solveLUpTo, 1 solveLtUpTo, 2 solveLFrom, 3 solveLtFrom
template <typename T>
static void solvePartial(bsp_solver* s, const T* m, T* v, int64_t stride, int32_t nrhs,
                         int32_t which, int64_t span) {
  if (which == 0) {
    s->solver->solveLUpTo(m, span, v, stride, nrhs);
  } else if (which == 1) {
    s->solver->solveLtUpTo(m, span, v, stride, nrhs);
  } else if (which == 2) {
    s->solver->solveLFrom(m, span, v, stride, nrhs);
  } else if (which == 3) {
    s->solver->solveLtFrom(m, span, v, stride, nrhs);
  } else {
    throw std::runtime_error("bsp_solve_partial: which must be 0..3");
  }
}
int bsp_solve_partial_f64(bsp_solver* s, const double* m, double* v, int64_t stride, int32_t nrhs,
                          int32_t which, int64_t span) {
  BSP_TRY
  solvePartial<double>(s, m, v, stride, nrhs, which, span);
  BSP_CATCH
}
int bsp_solve_partial_f32(bsp_solver* s, const float* m, float* v, int64_t stride, int32_t nrhs,
                          int32_t which, int64_t span) {
  BSP_TRY
  solvePartial<float>(s, m, v, stride, nrhs, which, span);
  BSP_CATCH
}

int bsp_add_mv_from_f64(bsp_solver* s, const double* m, int64_t span, const double* in,
                        int64_t in_stride, double* out, int64_t out_stride, int32_t nrhs,
                        double alpha) {
  BSP_TRY
  s->solver->addMvFrom(m, span, in, in_stride, out, out_stride, nrhs, alpha);
  BSP_CATCH
}
int bsp_add_mv_from_f32(bsp_solver* s, const float* m, int64_t span, const float* in,
                        int64_t in_stride, float* out, int64_t out_stride, int32_t nrhs,
                        float alpha) {
  BSP_TRY
  s->solver->addMvFrom(m, span, in, in_stride, out, out_stride, nrhs, alpha);
  BSP_CATCH
}
int bsp_pseudo_factor_from_f64(bsp_solver* s, double* d, int64_t span) {
  BSP_TRY
  s->solver->pseudoFactorFrom(d, span);
  BSP_CATCH
}
int bsp_pseudo_factor_from_f32(bsp_solver* s, float* d, int64_t span) {
  BSP_TRY
  s->solver->pseudoFactorFrom(d, span);
  BSP_CATCH
}

// batched solve: which = 0 solve, 1 solveL, 2 solveLt
template <typename T>
static void solveBatched(bsp_solver* s, const T* const* mats, T* const* vecs, int32_t batch,
                         int64_t stride, int32_t nrhs, int32_t which) {
  std::vector<T*> m(batch), v(vecs, vecs + batch);
  for (int32_t q = 0; q < batch; q++) m[q] = const_cast<T*>(mats[q]);
  if (which == 0) {
    s->solver->solve(&m, &v, stride, nrhs);
  } else if (which == 1) {
    s->solver->solveL(&m, &v, stride, nrhs);
  } else if (which == 2) {
    s->solver->solveLt(&m, &v, stride, nrhs);
  } else {
    throw std::runtime_error("bsp_solve_batched: which must be 0 (solve), 1 (L) or 2 (Lt)");
  }
}
// batched partial factor: which = 0 factorUpTo, 1 factorFrom
template <typename T>
static void factorPartialBatched(bsp_solver* s, T* const* ptrs, int32_t batch, int64_t span,
                                 int32_t which) {
  std::vector<T*> v(ptrs, ptrs + batch);
  if (which == 0) {
    s->solver->factorUpTo(&v, span);
  } else if (which == 1) {
    s->solver->factorFrom(&v, span);
  } else {
    throw std::runtime_error("bsp_factor_partial_batched: which must be 0 (UpTo) or 1 (From)");
  }
}
int bsp_factor_partial_batched_f64(bsp_solver* s, double* const* ptrs, int32_t batch, int64_t span,
                                   int32_t which) {
  BSP_TRY
  factorPartialBatched<double>(s, ptrs, batch, span, which);
  BSP_CATCH
}
int bsp_factor_partial_batched_f32(bsp_solver* s, float* const* ptrs, int32_t batch, int64_t span,
                                   int32_t which) {
  BSP_TRY
  factorPartialBatched<float>(s, ptrs, batch, span, which);
  BSP_CATCH
}
// batched partial solves: which as bsp_solve_partial (0 LUpTo, 1 LtUpTo, 2 LFrom, 3 LtFrom)
template <typename T>
static void solvePartialBatched(bsp_solver* s, const T* const* mats, T* const* vecs, int32_t batch,
                                int64_t stride, int32_t nrhs, int32_t which, int64_t span) {
  std::vector<T*> m(batch), v(vecs, vecs + batch);
  for (int32_t q = 0; q < batch; q++) m[q] = const_cast<T*>(mats[q]);
  if (which == 0) {
    s->solver->solveLUpTo(&m, span, &v, stride, nrhs);
  } else if (which == 1) {
    s->solver->solveLtUpTo(&m, span, &v, stride, nrhs);
  } else if (which == 2) {
    s->solver->solveLFrom(&m, span, &v, stride, nrhs);
  } else if (which == 3) {
    s->solver->solveLtFrom(&m, span, &v, stride, nrhs);
  } else {
    throw std::runtime_error("bsp_solve_partial_batched: which must be 0..3");
  }
}
int bsp_solve_partial_batched_f64(bsp_solver* s, const double* const* mats, double* const* vecs,
                                  int32_t batch, int64_t stride, int32_t nrhs, int32_t which,
                                  int64_t span) {
  BSP_TRY
  solvePartialBatched<double>(s, mats, vecs, batch, stride, nrhs, which, span);
  BSP_CATCH
}
int bsp_solve_partial_batched_f32(bsp_solver* s, const float* const* mats, float* const* vecs,
                                  int32_t batch, int64_t stride, int32_t nrhs, int32_t which,
                                  int64_t span) {
  BSP_TRY
  solvePartialBatched<float>(s, mats, vecs, batch, stride, nrhs, which, span);
  BSP_CATCH
}
int bsp_solve_batched_f64(bsp_solver* s, const double* const* mats, double* const* vecs,
                          int32_t batch, int64_t stride, int32_t nrhs, int32_t which) {
  BSP_TRY
  solveBatched<double>(s, mats, vecs, batch, stride, nrhs, which);
  BSP_CATCH
}
int bsp_solve_batched_f32(bsp_solver* s, const float* const* mats, float* const* vecs,
                          int32_t batch, int64_t stride, int32_t nrhs, int32_t which) {
  BSP_TRY
  solveBatched<float>(s, mats, vecs, batch, stride, nrhs, which);
  BSP_CATCH
}

int bsp_bal_linearize_f64(int64_t numObs, const int64_t* obsCam, const int64_t* obsPt,
                          const double* obsXy, const double* cams, const double* pts, double* res,
                          double* Jc, double* Jp, void* stream) {
  BSP_TRY
  balLinearize(numObs, obsCam, obsPt, obsXy, cams, pts, res, Jc, Jp, stream);
  BSP_CATCH
}
// bsp_bal_fill_hessian_*: the kernels hard-code 3 x 3 point blocks, 9 x 9 camera blocks and 9 x 3
// camera-point blocks, points first -- a solver with any other parameter sizes would send their
// atomics out of bounds, so the layout is checked on the host before the launch
static void checkBalLayout(bsp_solver* s, int64_t numPts, int64_t numCams, int64_t camSize = 9) {
  if (s->balOk[0] == numPts && s->balOk[1] == numCams && s->balOk[2] == camSize) return;
  const BaSpaCho::Solver& solver = *s->solver;
  if (camSize != 9 && camSize != 6) throw std::runtime_error("bsp_bal_fill_hessian: camera size must be 9 or 6");
  BASPACHO_CHECK_GE(numPts, 0);
  BASPACHO_CHECK_GE(numCams, 0);
  BASPACHO_CHECK_EQ(numPts + numCams, solver.skel().numSpans());
  const auto acc = solver.accessor();
  for (int64_t p = 0; p < numPts + numCams; p++) {
    const int64_t want = p < numPts ? 3 : camSize;
    if (acc.paramSize(p) != want) {
      throw std::runtime_error("bsp_bal_fill_hessian: parameter " + std::to_string(p) + " has size " +
                               std::to_string(acc.paramSize(p)) + ", the BAL layout needs " +
                               std::to_string(want) + " (points of size 3 first, then cameras of size " +
                               std::to_string(camSize) + ")");
    }
  }
  s->balOk[0] = numPts;
  s->balOk[1] = numCams;
  s->balOk[2] = camSize;
}

int bsp_bal_fill_hessian_f64(bsp_solver* s, int64_t numPts, int64_t numCams, int64_t numObs,
                             const int64_t* obsCam, const int64_t* obsPt, const double* Jc,
                             const double* Jp, const double* res, double lambda, double* data,
                             double* grad, int64_t* dbg, void* stream) {
  BSP_TRY
  checkBalLayout(s, numPts, numCams);
  balFillHessian<double>(s->solver->deviceAccessor(), 9, numPts, numCams, numObs, obsCam, obsPt, Jc, Jp,
                         res, lambda, data, grad, dbg, stream);
  BSP_CATCH
}
int bsp_bal_fill_hessian_f32(bsp_solver* s, int64_t numPts, int64_t numCams, int64_t numObs,
                             const int64_t* obsCam, const int64_t* obsPt, const double* Jc,
                             const double* Jp, const double* res, float lambda, float* data,
                             float* grad, int64_t* dbg, void* stream) {
  BSP_TRY
  checkBalLayout(s, numPts, numCams);
  balFillHessian<float>(s->solver->deviceAccessor(), 9, numPts, numCams, numObs, obsCam, obsPt, Jc, Jp,
                        res, lambda, data, grad, dbg, stream);
  BSP_CATCH
}

int bsp_bal_linearize_se3_f64(int64_t numObs, const int64_t* obsCam, const int64_t* obsPt,
                              const double* obsXy, const double* cams, const double* pts, double* res,
                              double* Jc, double* Jp, void* stream) {
  BSP_TRY
  balLinearizeSe3(numObs, obsCam, obsPt, obsXy, cams, pts, res, Jc, Jp, stream);
  BSP_CATCH
}
int bsp_bal_fill_hessian_se3_f64(bsp_solver* s, int64_t numPts, int64_t numCams, int64_t numObs,
                                 const int64_t* obsCam, const int64_t* obsPt, const double* Jc,
                                 const double* Jp, const double* res, double lambda, double* data,
                                 double* grad, void* stream) {
  BSP_TRY
  checkBalLayout(s, numPts, numCams, 6);
  balFillHessian<double>(s->solver->deviceAccessor(), 6, numPts, numCams, numObs, obsCam, obsPt, Jc, Jp,
                         res, lambda, data, grad, nullptr, stream);
  BSP_CATCH
}
int bsp_bal_fill_hessian_se3_f32(bsp_solver* s, int64_t numPts, int64_t numCams, int64_t numObs,
                                 const int64_t* obsCam, const int64_t* obsPt, const double* Jc,
                                 const double* Jp, const double* res, float lambda, float* data,
                                 float* grad, void* stream) {
  BSP_TRY
  checkBalLayout(s, numPts, numCams, 6);
  balFillHessian<float>(s->solver->deviceAccessor(), 6, numPts, numCams, numObs, obsCam, obsPt, Jc, Jp,
                        res, lambda, data, grad, nullptr, stream);
  BSP_CATCH
}

double bsp_factor_flops(const bsp_solver* s) { return s->solver->factorFlops(); }

int64_t bsp_plan_levels(bsp_solver* s, int64_t* out, int64_t capacity) {
  try {
    std::vector<int64_t> v = hipBackendPlanLevels(s->solver->internalSymbolicContext(), 0,
                                                  s->solver->skel().numLumps());
    const int64_t n = (int64_t)v.size();
    if (out) std::copy(v.begin(), v.begin() + std::min(n, capacity), out);
    return n;
  } catch (const std::exception& e) {
    g_lastError = e.what();
    return -1;
  }
}

int bsp_plan_stats_full(bsp_solver* s, bsp_plan_stats* out) {
  BSP_TRY
  HipPlanStats p = hipBackendPlanStats(s->solver->internalSymbolicContext(), 0,
                                       s->solver->skel().numLumps());
  out->flops = p.flops;
  out->upd_elems = p.updElems;
  out->upd_flops = p.updFlops;
  out->elim_pair_elems = p.elimPairElems;
  out->elim_pair_flops = p.elimPairFlops;
  out->elim_col_elems = p.elimColElems;
  out->upd_flops_direct = p.updFlopsDirect;
  out->elim_pair_operand_elems = p.elimPairOperandElems;
  out->elim_target_elems = p.elimTargetElems;
  out->trsm_flops = p.trsmFlops;
  out->potrf_flops = p.potrfFlops;
  out->trsm_flops_merged = p.trsmFlopsMerged;
  out->potrf_flops_fused = p.potrfFlopsFused;
  out->num_launches = p.numLaunches;
  out->num_levels = p.numLevels;
  out->num_panels = p.numPanels;
  out->num_segs = p.numSegs;
  out->num_upd_tasks = p.numUpdTasks;
  out->num_trsm_tasks = p.numTrsmTasks;
  out->chain_tab_entries = p.chainTabEntries;
  out->max_panels_in_level = p.maxPanelsInLevel;
  out->num_atomic_upd_tasks = p.numAtomicUpdTasks;
  out->num_gather_groups = p.numGatherGroups;
  out->num_fork_levels = p.numForkLevels;
  out->deferred_flops = p.deferredFlops;
  out->tail_upd_flops = p.tailUpdFlops;
  out->num_tail_panels = p.numTailPanels;
  BSP_CATCH
}

int bsp_run_counters_get(bsp_solver* s, bsp_run_counters* out) {
  BSP_TRY
  const HipRunCounters c = hipBackendRunCounters(s->solver->internalSymbolicContext());
  out->sweep_launches = c.sweepLaunches;
  out->sweep_timeouts = c.sweepTimeouts;
  out->split_lists_used = c.splitListsUsed;
  out->sub_batches_enqueued = c.subBatchesEnqueued;
  out->lookahead_forks = c.lookaheadForks;
  out->sweeps_retired = c.sweepsRetired;
  out->sweep_error_pending = c.sweepErrorPending;
  out->gather_chunks_overlapped = c.gatherChunksOverlapped;
  out->tail_launches = c.tailLaunches;
  out->sweep_mfma_launches = c.sweepMfmaLaunches;
  out->solve_wide_launches = c.solveWideLaunches;
  out->inv_reused = c.invReused;
  out->potrf_folded_levels = c.potrfFoldedLevels;
  BSP_CATCH
}

int bsp_probe_mfma_f64(double* tflops) {
  BSP_TRY
  *tflops = hipBackendMfmaF64ProbeTflops();
  BSP_CATCH
}

int bsp_debug_read_extents(unsigned long long* out, int max_launches, int* n) {
  BSP_TRY
  *n = hipBackendReadExtents(out, max_launches);
  BSP_CATCH
}
int bsp_debug_read_trace(long long* out, int max_records, int* n_records) {
  BSP_TRY
  *n_records = hipBackendReadTrace(out, max_records);
  BSP_CATCH
}

static void factorProfiled(bsp_solver* s, double* d, bool inSitu, double ms[6], int64_t launches[6],
                           double* busy = nullptr) {
  HipKernelProfile prof;
  SymbolicCtx& sym = s->solver->internalSymbolicContext();
  hipBackendSetProfile(sym, &prof, inSitu);
  try {
    s->solver->factor(d);
  } catch (...) {
    hipBackendSetProfile(sym, nullptr);
    throw;
  }
  hipBackendSetProfile(sym, nullptr);
  for (int i = 0; i < kProfNumKinds; i++) {
    ms[i] = prof.ms[i];
    launches[i] = prof.launches[i];
    if (busy) busy[i] = prof.busyMs[i];
  }
}
int bsp_factor_profiled_f64(bsp_solver* s, double* d, double ms[6], int64_t launches[6]) {
  BSP_TRY
  factorProfiled(s, d, false, ms, launches);
  BSP_CATCH
}
int bsp_factor_profiled_insitu_f64(bsp_solver* s, double* d, double ms[6], int64_t launches[6]) {
  BSP_TRY
  factorProfiled(s, d, true, ms, launches);
  BSP_CATCH
}
int bsp_factor_profiled_busy_f64(bsp_solver* s, double* d, double ms[6], int64_t launches[6],
                                 double busy_ms[6]) {
  BSP_TRY
  factorProfiled(s, d, true, ms, launches, busy_ms);
  BSP_CATCH
}

// ---- plan (de)serialisation: [magic, canFactorUpTo, nSpans, nLumps, nRowInd, nRanges,
//      spanStart.., lumpToSpan.., colPtr.., rowInd.., ranges.., permutation..]
static const int64_t kPlanMagic = 0x42535041434d4431LL;  // "BSPACMD1"

int bsp_plan_serialize(const bsp_solver* s, int64_t* buf, int64_t capacity, int64_t* needed) {
  BSP_TRY
  const CoalescedBlockMatrixSkel& k = s->solver->skel();
  const auto& ranges = s->solver->sparseEliminationRanges();
  const auto& perm = s->solver->paramToSpan();
  // column pointers / row indices of the lump columns are exactly chainColPtr / chainRowSpan
  const int64_t nSpans = k.numSpans(), nLumps = k.numLumps();
  const int64_t nRow = (int64_t)k.chainRowSpan.size(), nRanges = (int64_t)ranges.size();
  const int64_t total = 6 + (nSpans + 1) + (nLumps + 1) + (nLumps + 1) + nRow + nRanges + nSpans;
  *needed = total;
  if (!buf || capacity < total) return 0;
  int64_t* p = buf;
  *p++ = kPlanMagic;
  *p++ = s->solver->canFactorUpToSpan();
  *p++ = nSpans;
  *p++ = nLumps;
  *p++ = nRow;
  *p++ = nRanges;
  p = std::copy(k.spanStart.begin(), k.spanStart.end(), p);
  p = std::copy(k.lumpToSpan.begin(), k.lumpToSpan.end(), p);
  p = std::copy(k.chainColPtr.begin(), k.chainColPtr.end(), p);
  p = std::copy(k.chainRowSpan.begin(), k.chainRowSpan.end(), p);
  p = std::copy(ranges.begin(), ranges.end(), p);
  p = std::copy(perm.begin(), perm.end(), p);
  BSP_CATCH
}

int bsp_create_solver_from_plan(const int64_t* buf, int64_t len, bsp_solver** out) {
  BSP_TRY
  BASPACHO_CHECK_NOTNULL(out);
  BASPACHO_CHECK_GE(len, 6);
  BASPACHO_CHECK_EQ(buf[0], kPlanMagic);
  const int64_t canUpTo = buf[1], nSpans = buf[2], nLumps = buf[3], nRow = buf[4],
                nRanges = buf[5];
  const int64_t total = 6 + (nSpans + 1) + (nLumps + 1) + (nLumps + 1) + nRow + nRanges + nSpans;
  BASPACHO_CHECK_EQ(len, total);
  const int64_t* p = buf + 6;
  std::vector<int64_t> spanStart(p, p + nSpans + 1);
  p += nSpans + 1;
  std::vector<int64_t> lumpToSpan(p, p + nLumps + 1);
  p += nLumps + 1;
  std::vector<int64_t> colPtr(p, p + nLumps + 1);
  p += nLumps + 1;
  std::vector<int64_t> rowInd(p, p + nRow);
  p += nRow;
  std::vector<int64_t> ranges(p, p + nRanges);
  p += nRanges;
  std::vector<int64_t> perm(p, p + nSpans);
  CoalescedBlockMatrixSkel skel(spanStart, lumpToSpan, colPtr, rowInd);
  std::unique_ptr<bsp_solver> h(new bsp_solver);
  h->solver.reset(
      new Solver(std::move(skel), std::move(ranges), std::move(perm), hipOps(), canUpTo));
  *out = h.release();
  BSP_CATCH
}

int bsp_sparse_structure_op(int op, int64_t n, const int64_t* ptrs, const int64_t* inds,
                            const int64_t* arg, int64_t argLen, int32_t flag, int64_t* outPtrs,
                            int64_t* outInds, int64_t capacity, int64_t* outN, int64_t* outNnz) {
  BSP_TRY
  SparseStructure ss(std::vector<int64_t>(ptrs, ptrs + n + 1),
                     std::vector<int64_t>(inds, inds + ptrs[n]));
  SparseStructure res;
  switch (op) {
    case BSP_SS_TRANSPOSE: res = ss.transpose(); break;
    case BSP_SS_CLEAR: res = ss.clear(flag != 0); break;
    case BSP_SS_SYM_PERMUTATION: {
      BASPACHO_CHECK_EQ(argLen, n);
      res = ss.symmetricPermutation(std::vector<int64_t>(arg, arg + n), flag != 0);
      break;
    }
    case BSP_SS_INDEP_ELIM_FILL:
      BASPACHO_CHECK_EQ(argLen, 2);
      res = ss.addIndependentEliminationFill(arg[0], arg[1]);
      break;
    case BSP_SS_FULL_ELIM_FILL: res = ss.addFullEliminationFill(); break;
    case BSP_SS_FILL_REDUCING_PERM: {
      std::vector<int64_t> perm = ss.fillReducingPermutation();
      *outN = n;
      *outNnz = (int64_t)perm.size();
      if (capacity < (int64_t)perm.size()) throw std::runtime_error("output capacity too small");
      std::copy(perm.begin(), perm.end(), outInds);
      return 0;
    }
    case BSP_SS_EXTRACT_RIGHT_BOTTOM:
      BASPACHO_CHECK_EQ(argLen, 1);
      res = ss.extractRightBottom(arg[0]);
      break;
    default: throw std::runtime_error("bsp_sparse_structure_op: unknown op");
  }
  *outN = res.order();
  *outNnz = (int64_t)res.inds.size();
  if (capacity < (int64_t)res.inds.size()) throw std::runtime_error("output capacity too small");
  std::copy(res.ptrs.begin(), res.ptrs.end(), outPtrs);
  std::copy(res.inds.begin(), res.inds.end(), outInds);
  BSP_CATCH
}
