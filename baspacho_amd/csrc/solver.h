// Solver: a symbolic decomposition plus the operations that run on externally allocated
// numeric data.  Public surface mirrors baspacho/baspacho/Solver.h:34-237 (createSolver,
// factor/solve families, accessors, Settings, BackendType, AddFillPolicy) so that callers of
// the reference (bench, BAL_bench, Theseus' baspacho_solver.cpp) compile against it unchanged.
// Numeric data is DEVICE memory: the only engine in this build is the MI355X HIP backend.
#pragma once

#include <memory>
#include <unordered_set>
#include <vector>

#include "backend_options.h"
#include "mat_ops.h"
#include "skeleton.h"
#include "sparse_structure.h"

namespace BaSpaCho {

class Solver {
 public:
  // from a RAW factor skeleton (normally use createSolver)
  Solver(CoalescedBlockMatrixSkel&& factorSkel, std::vector<int64_t>&& sparseElimRanges,
         std::vector<int64_t>&& permutation, OpsPtr&& ops, int64_t canFactorUpTo = -1);

  PermutedCoalescedAccessor accessor() const {
    PermutedCoalescedAccessor acc;
    acc.init(factorSkel.accessor(), permutation.data());
    return acc;
  }

  // accessor whose arrays live in device memory (usable inside a HIP kernel)
  PermutedCoalescedAccessor deviceAccessor() const { return symCtx->deviceAccessor(); }

  void enableStats(bool enabled = true);
  void printStats() const;
  void resetStats();

  // execution stream (hipStream_t) used by every subsequent factor/solve call
  void setStream(void* stream) { symCtx->setStream(stream); }

  template <typename T>
  void factor(T* data, bool verbose = false) const;

  template <typename T>
  void solve(const T* matData, T* vecData, int64_t stride, int nRHS) const;
  template <typename T>
  void solveL(const T* matData, T* vecData, int64_t stride, int nRHS) const;
  template <typename T>
  void solveLt(const T* matData, T* vecData, int64_t stride, int nRHS) const;

  template <typename T>
  void factorUpTo(T* data, int64_t spanIndex, bool verbose = false) const;
  template <typename T>
  void factorFrom(T* data, int64_t spanIndex, bool verbose = false) const;

  template <typename T>
  void solveLUpTo(const T* data, int64_t spanIndex, T* vecData, int64_t stride, int nRHS) const;
  template <typename T>
  void solveLtUpTo(const T* data, int64_t spanIndex, T* vecData, int64_t stride, int nRHS) const;
  template <typename T>
  void solveLFrom(const T* data, int64_t spanIndex, T* vecData, int64_t stride, int nRHS) const;
  template <typename T>
  void solveLtFrom(const T* data, int64_t spanIndex, T* vecData, int64_t stride, int nRHS) const;

  // out += alpha * A * in on the (un-factored) block from spanIndex on   (Solver.h:89-91)
  template <typename T>
  void addMvFrom(const T* matData, int64_t spanIndex, const T* inVecData, int64_t inStride,
                 T* outVecData, int64_t outStride, int nRHS, BaseType<T> alpha = 1.0) const;

  // pseudo-factor: Cholesky of the diagonal block of every span from spanIndex on, rows below
  // divided by it   (Solver.h:92-94; preconditioners of the PCG example)
  template <typename T>
  void pseudoFactorFrom(T* data, int64_t spanIndex, bool verbose = false) const;

  int64_t order() const { return factorSkel.order(); }
  int64_t dataSize() const { return factorSkel.dataSize(); }
  int64_t canFactorUpToSpan() const { return canFactorUpTo; }
  int64_t spanVectorOffset(int64_t spanIndex) const {
    return factorSkel.spanVectorOffset(spanIndex);
  }
  int64_t spanMatrixOffset(int64_t spanIndex) const {
    return factorSkel.spanMatrixOffset(spanIndex);
  }
  const CoalescedBlockMatrixSkel& skel() const { return factorSkel; }
  const std::vector<int64_t>& sparseEliminationRanges() const { return sparseElimRanges; }
  const std::vector<int64_t>& paramToSpan() const { return permutation; }

  // algorithmic flops of a full factor: sum over lumps of n^3/3 + r n^2 + r^2 n
  // (n = lump width, r = rows below the diagonal block); SURVEY.md section 8(d)
  double factorFlops() const;

  // TESTING
  SymbolicCtx& internalSymbolicContext() { return *symCtx; }
  SymElimCtx& internalGetElimCtx(size_t i) {
    BASPACHO_CHECK_LT(i, elimCtxs.size());
    return *elimCtxs[i];
  }

 private:
  void initElimination();
  int64_t boardElimTempSize(int64_t lump, int64_t boardIndexInCol) const;

  template <typename T>
  void factorLump(NumericCtx<T>& numCtx, T* data, int64_t lump) const;
  template <typename T>
  void eliminateBoard(NumericCtx<T>& numCtx, T* data, int64_t ptr) const;
  template <typename T>
  void internalFactorRange(T* data, int64_t startSpanIndex, int64_t endSpanIndex,
                           bool verbose = false) const;
  template <typename T>
  void internalSolveLRange(SolveCtx<T>& slvCtx, const T* data, int64_t startSpanIndex,
                           int64_t endSpanIndex, T* vecData, int64_t stride, int nRHS) const;
  template <typename T>
  void internalSolveLtRange(SolveCtx<T>& slvCtx, const T* data, int64_t startSpanIndex,
                            int64_t endSpanIndex, T* vecData, int64_t stride, int nRHS) const;

  CoalescedBlockMatrixSkel factorSkel;
  std::vector<int64_t> sparseElimRanges;
  std::vector<int64_t> permutation;  // on indices: v'[p[i]] = v[i]
  int64_t canFactorUpTo;

  OpsPtr ops;
  SymbolicCtxPtr symCtx;
  std::vector<SymElimCtxPtr> elimCtxs;
  std::vector<int64_t> startElimRowPtr;
  int64_t maxElimTempSize = 0;
};

using SolverPtr = std::unique_ptr<Solver>;

// Engine selection.  BackendHip is the MI355X engine; BackendCuda is accepted as an alias so
// that GPU callers of the reference keep working.  BackendRef / BackendFast (CPU) are not part
// of this build and make createSolver throw.
enum BackendType {
  BackendRef,
  BackendFast,
  BackendCuda,
  BackendHip,
};

enum AddFillPolicy {
  AddFillComplete,       // fill for complete factoring, reorder
  AddFillForAutoElims,   // fill for given+auto elimination ranges, reorder
  AddFillForGivenElims,  // fill for the given elimination ranges, no reorder
  AddFillNone,           // no fill, no reorder
};

struct ComputationModel;

struct Settings {
  bool findSparseEliminationRanges = true;
  int numThreads = 16;  // unused by the HIP engine; kept for source compatibility
  BackendType backend = BackendHip;
  AddFillPolicy addFillPolicy = AddFillComplete;
  const ComputationModel* computationModel = nullptr;
  // extension: schedule switches of the MI355X backend (backend_options.h); null = defaults
  const HipBackendOptions* hipOptions = nullptr;
};

SolverPtr createSolver(const Settings& settings, const std::vector<int64_t>& paramSizes,
                       const SparseStructure& ss, const std::vector<int64_t>& sparseElimRanges = {},
                       const std::unordered_set<int64_t>& elimLastIds = {});

}  // namespace BaSpaCho
