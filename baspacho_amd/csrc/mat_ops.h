// Backend (operator) boundary between the Solver driver and a numeric engine.
// Same abstract classes, method names, argument meaning and typed-context convention as the
// reference's baspacho/baspacho/MatOps.h:48-221 so that a backend written against that header
// plugs in here and vice versa.  T is double, float, or std::vector<double*> / std::vector<float*>
// for a batch of identical-structure matrices (device pointers).
//
// Extension (not in the reference): a backend may advertise a *fused* factor path
// (NumericCtx::hasFusedFactor / factorRange).  The HIP backend uses it to replace the
// reference's host-serial per-op loop (Solver.cpp:198-218) by level-scheduled launches driven
// from a device-resident plan; the per-op virtuals remain the contract for other backends.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeindex>
#include <vector>

#include "bsp_utils.h"
#include "skeleton.h"

namespace BaSpaCho {

struct Ops;
struct SymbolicCtx;
struct SymElimCtx;
template <typename T>
struct NumericCtx;
template <typename T>
struct SolveCtx;
using OpsPtr = std::unique_ptr<Ops>;
using SymbolicCtxPtr = std::unique_ptr<SymbolicCtx>;
using SymElimCtxPtr = std::unique_ptr<SymElimCtx>;
template <typename T>
using NumericCtxPtr = std::unique_ptr<NumericCtx<T>>;
template <typename T>
using SolveCtxPtr = std::unique_ptr<SolveCtx<T>>;

template <typename T>
struct Batch {
  using BaseType = T;
  static int getSize(const T*) { return 1; }
};

template <typename T>
struct Batch<std::vector<T*>> {
  using BaseType = T;
  static int getSize(const std::vector<T*>* data) { return data ? (int)data->size() : 0; }
};

template <typename T>
using BaseType = typename Batch<T>::BaseType;

struct Ops {
  virtual ~Ops() {}
  virtual SymbolicCtxPtr createSymbolicCtx(const CoalescedBlockMatrixSkel& skel,
                                           const std::vector<int64_t>& permutation) = 0;
};

struct NumericCtxBase {
  virtual ~NumericCtxBase() {}
};

struct SolveCtxBase {
  virtual ~SolveCtxBase() {}
};

struct SymbolicCtx {
  virtual ~SymbolicCtx() {}

  virtual SymElimCtxPtr prepareElimination(int64_t lumpsBegin, int64_t lumpsEnd) = 0;

  virtual NumericCtxBase* createNumericCtxForType(std::type_index tIdx, int64_t tempBufSize,
                                                  int batchSize) = 0;

  virtual SolveCtxBase* createSolveCtxForType(std::type_index tIdx, int nRHS, int batchSize) = 0;

  virtual PermutedCoalescedAccessor deviceAccessor() = 0;

  // extension: the solver tells the backend which lump ranges get sparse elimination, so a
  // backend with a fused path can lay out its whole plan once (default: ignore)
  virtual void setSparseElimRanges(const std::vector<int64_t>& /*ranges*/) {}

  // extension: called once by the Solver constructor, after setSparseElimRanges: lumps [0, upToLump)
  // are what factor() will be asked for (Solver::canFactorUpTo) -- a backend may build its numeric
  // plan now, as the reference builds its contexts in the constructor (Solver.cpp:24-40)
  virtual void prepareFactor(int64_t /*upToLump*/) {}

  // extension: execution stream for backends that have one (HIP: hipStream_t)
  virtual void setStream(void* /*stream*/) {}

  template <typename T>
  NumericCtxPtr<T> createNumericCtx(int64_t tempBufSize, const T* data);

  template <typename T>
  SolveCtxPtr<T> createSolveCtx(int nRHS, const T* data);

  mutable OpStat potrfStat;
  mutable int64_t potrfBiggestN = 0;
  mutable OpStat trsmStat;
  mutable OpStat sygeStat;
  mutable int64_t gemmCalls = 0;
  mutable int64_t syrkCalls = 0;
  mutable OpStat asmblStat;

  mutable OpStat solveSparseLStat;
  mutable OpStat solveSparseLtStat;
  mutable OpStat pseudoFactorStat;
  mutable OpStat symmStat;
  mutable OpStat solveLStat;
  mutable OpStat solveLtStat;
  mutable OpStat solveGemvStat;
  mutable OpStat solveGemvTStat;
  mutable OpStat solveAssVStat;
  mutable OpStat solveAssVTStat;
};

struct SymElimCtx {
  virtual ~SymElimCtx() {}
  mutable OpStat elimStat;
};

template <typename T>
struct NumericCtx : NumericCtxBase {
  virtual ~NumericCtx() {}

  // per span: factor the diagonal block, solve the column below it
  virtual void pseudoFactorSpans(T* data, int64_t spanBegin, int64_t spanEnd) = 0;

  // (parallel) sparse elimination of the independent lumps [lumpsBegin, lumpsEnd)
  virtual void doElimination(const SymElimCtx& elimData, T* data, int64_t lumpsBegin,
                             int64_t lumpsEnd) = 0;

  // in-place dense Cholesky of the row-major n x n block at offA (lower triangle)
  virtual void potrf(int64_t n, T* data, int64_t offA) = 0;

  // X * A.lower().transpose() = B, in place on the k x n row-major block at offB
  virtual void trsm(int64_t n, int64_t k, T* data, int64_t offA, int64_t offB) = 0;

  // temp(n x m, row-major) = P(n x k) * P(0:m,:)^T, P row-major at `offset`
  virtual void saveSyrkGemm(int64_t m, int64_t n, int64_t k, const T* data, int64_t offset) = 0;

  virtual void prepareAssemble(int64_t targetLump) = 0;

  // target column -= temp, scattered by block rows / block columns
  virtual void assemble(T* data, int64_t rectRowBegin, int64_t dstStride, int64_t srcColDataOffset,
                        int64_t srcRectWidth, int64_t numBlockRows, int64_t numBlockCols) = 0;

  // ---- extension: fused factor path
  virtual bool hasFusedFactor() const { return false; }

  // factor everything between lump boundaries [startLump, upToLump): all sparse-elimination
  // ranges inside, then the dense lumps; target columns >= upToLump still receive the updates
  // of the factored sources (Schur complement), exactly like the per-op loop.
  virtual void factorRange(T* /*data*/, int64_t /*startLump*/, int64_t /*upToLump*/) {
    throw std::runtime_error("factorRange: not supported by this backend");
  }
};

template <typename T>
struct SolveCtx : SolveCtxBase {
  virtual ~SolveCtx() {}

  virtual void sparseElimSolveL(const SymElimCtx& elimData, const T* data, int64_t lumpsBegin,
                                int64_t lumpsEnd, T* C, int64_t ldc) = 0;

  virtual void sparseElimSolveLt(const SymElimCtx& elimData, const T* data, int64_t lumpsBegin,
                                 int64_t lumpsEnd, T* C, int64_t ldc) = 0;

  virtual void symm(const T* data, int64_t offset, int64_t n, const T* C, int64_t offC, int64_t ldc,
                    T* D, int64_t ldd, BaseType<T> alpha) = 0;

  virtual void solveL(const T* data, int64_t offset, int64_t n, T* C, int64_t offC,
                      int64_t ldc) = 0;

  virtual void gemv(const T* data, int64_t offset, int64_t nRows, int64_t nCols, const T* A,
                    int64_t offA, int64_t lda, BaseType<T> alpha) = 0;

  virtual void assembleVec(int64_t chainColPtr, int64_t numColItems, T* C, int64_t ldc) = 0;

  virtual void solveLt(const T* data, int64_t offset, int64_t n, T* C, int64_t offC,
                       int64_t ldc) = 0;

  virtual void gemvT(const T* data, int64_t offset, int64_t nRows, int64_t nCols, T* A,
                     int64_t offA, int64_t lda, BaseType<T> alpha) = 0;

  virtual void assembleVecT(const T* C, int64_t ldc, int64_t chainColPtr, int64_t numColItems) = 0;

  virtual bool hasFragmentedOps() { return false; }

  virtual void fragmentedMV(const T*, const T*, int64_t, int64_t, T*, BaseType<T>) {
    throw std::runtime_error("fragmentedMV: not supported");
  }
  virtual void fragmentedSolveL(const T*, int64_t, int64_t, T*) {
    throw std::runtime_error("fragmentedSolveL: not supported");
  }
  virtual void fragmentedSolveLt(const T*, int64_t, int64_t, T*) {
    throw std::runtime_error("fragmentedSolveLt: not supported");
  }

  // ---- extension: fused solve path (whole L / L^T sweeps between lump boundaries)
  virtual bool hasFusedSolve() const { return false; }
  virtual void solveLRange(const T*, int64_t /*startLump*/, int64_t /*upToLump*/, T* /*C*/,
                           int64_t /*ldc*/) {
    throw std::runtime_error("solveLRange: not supported by this backend");
  }
  virtual void solveLtRange(const T*, int64_t /*startLump*/, int64_t /*upToLump*/, T* /*C*/,
                            int64_t /*ldc*/) {
    throw std::runtime_error("solveLtRange: not supported by this backend");
  }
  // extension (fused Solver::addMvFrom): out += alpha * A * in for the symmetric trailing block
  // made of the lump columns [startLump, upToLump)
  virtual void addMvRange(const T*, int64_t /*startLump*/, int64_t /*upToLump*/, const T* /*in*/,
                          int64_t /*inStride*/, T* /*out*/, int64_t /*outStride*/,
                          BaseType<T> /*alpha*/) {
    throw std::runtime_error("addMvRange: not supported by this backend");
  }
};

template <typename T>
NumericCtxPtr<T> SymbolicCtx::createNumericCtx(int64_t tempBufSize, const T* data) {
  int batchSize = Batch<T>::getSize(data);
  NumericCtxBase* ctx = createNumericCtxForType(std::type_index(typeid(T)), tempBufSize, batchSize);
  NumericCtx<T>* typed = dynamic_cast<NumericCtx<T>*>(ctx);
  if (!typed) delete ctx;
  BASPACHO_CHECK_NOTNULL(typed);
  return NumericCtxPtr<T>(typed);
}

template <typename T>
SolveCtxPtr<T> SymbolicCtx::createSolveCtx(int nRHS, const T* data) {
  int batchSize = Batch<T>::getSize(data);
  SolveCtxBase* ctx = createSolveCtxForType(std::type_index(typeid(T)), nRHS, batchSize);
  SolveCtx<T>* typed = dynamic_cast<SolveCtx<T>*>(ctx);
  if (!typed) delete ctx;
  BASPACHO_CHECK_NOTNULL(typed);
  return SolveCtxPtr<T>(typed);
}

// MI355X backend (hand-written HIP for gfx950).  The CPU backends of the reference
// (simpleOps/fastOps) are deliberately NOT part of the product: a CPU restatement lives under
// oracle/ as test infrastructure only.
struct HipBackendOptions;
// (options: already resolved against the environment by the caller, or null = defaults + environment)
OpsPtr hipOps(const HipBackendOptions* options = nullptr);

}  // namespace BaSpaCho
