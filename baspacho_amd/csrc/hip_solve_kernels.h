// HIP kernels of the triangular solves (Solver::solve / solveL / solveLt), correctness-first.
// They reuse the factor plan: panels, levels and the 64-row task tiles.  Replaces the solve part of
// MatOpsCuda.cu (sparseElim_* kernels :883-1012, assembleVec(T) :836-880, cublas trsm/gemm calls
// :1093-1181).  One right-hand side per blockIdx.y; vec is column-major (order x nRHS, ld = ldc).
#pragma once

#include <hip/hip_runtime.h>

#include "hip_kernels.h"

namespace BaSpaCho {
namespace hipk {

// matrix data and right-hand sides of one launch: a single pair, or one pair per batch entry
// (blockIdx.z); nRHS columns of a vector are ldc apart (blockIdx.y)
template <typename T>
struct SolveRef {
  const T* mat;
  T* vec;
  const T* const* mats;
  T* const* vecs;
  int64_t ldc;
};
template <typename T>
__device__ __forceinline__ GP<const T> solveMat(const SolveRef<T>& r) {
  return (GP<const T>)(r.mats ? r.mats[blockIdx.z] : r.mat);
}
template <typename T>
__device__ __forceinline__ GP<T> solveVec(const SolveRef<T>& r) {
  return (GP<T>)(r.vecs ? r.vecs[blockIdx.z] : r.vec) + (int64_t)blockIdx.y * r.ldc;
}

__device__ __forceinline__ double waveSum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float waveSum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- sparse-elimination ranges: one THREAD per (small) lump ---------------------------------
// forward:  x_l <- D^-1 x_l ;  x[rows of every chain below] -= B * x_l   (atomics: several lumps
//           hit the same rows)          (reference: sparseElim_diagSolveL + subDiagMult, :883-946)
// backward: x_l -= sum B^T x[rows] ;  x_l <- D^-T x_l                 (:949-1012)
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) void solveElimSmall(SkelDev sk, SolveRef<T> ref,
                                                      int64_t lumpBegin, int64_t lumpEnd) {
  const int64_t l = lumpBegin + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (l >= lumpEnd) return;
  const int n = (int)(sk.lumpStart[l + 1] - sk.lumpStart[l]);
  if (n > kElimSmallMax) return;  // wide lumps go through the panel kernels
  GP<const T> data = solveMat(ref);
  GP<T> vec = solveVec(ref);
  const int64_t c0 = sk.chainColPtr[l], cEnd = sk.chainColPtr[l + 1];
  const int64_t diagCh = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
  GP<const T> D = data + sk.chainData[c0];
  GP<T> xl = vec + sk.lumpStart[l];
  T x[kElimSmallMax];
  for (int i = 0; i < n; i++) x[i] = xl[i];
  if (!BACKWARD) {
    for (int i = 0; i < n; i++) {
      T s = x[i];
      for (int j = 0; j < i; j++) s -= D[i * n + j] * x[j];
      x[i] = s / D[i * n + i];
    }
    for (int i = 0; i < n; i++) xl[i] = x[i];
    for (int64_t c = c0 + diagCh; c < cEnd; c++) {
      const int64_t span = sk.chainRowSpan[c];
      const int rows = (int)(sk.spanStart[span + 1] - sk.spanStart[span]);
      GP<const T> B = data + sk.chainData[c];
      GP<T> y = vec + sk.spanStart[span];
      for (int r = 0; r < rows; r++) {
        T s = T(0);
        for (int k = 0; k < n; k++) s += B[r * n + k] * x[k];
        atomicSub(y + r, s);
      }
    }
  } else {
    for (int64_t c = c0 + diagCh; c < cEnd; c++) {
      const int64_t span = sk.chainRowSpan[c];
      const int rows = (int)(sk.spanStart[span + 1] - sk.spanStart[span]);
      GP<const T> B = data + sk.chainData[c];
      GP<const T> y = vec + sk.spanStart[span];
      for (int r = 0; r < rows; r++) {
        const T yr = y[r];
        for (int k = 0; k < n; k++) x[k] -= B[r * n + k] * yr;
      }
    }
    for (int i = n - 1; i >= 0; i--) {
      T s = x[i];
      for (int j = i + 1; j < n; j++) s -= D[j * n + i] * x[j];
      x[i] = s / D[i * n + i];
    }
    for (int i = 0; i < n; i++) xl[i] = x[i];
  }
}

// ---- sparse-elimination ranges, forward pass in gather form --------------------------------
// K-S1: x_l <- D^-1 x_l for the small lumps of a range (thread per lump)
template <typename T>
__global__ __launch_bounds__(256) void solveElimDiagL(SkelDev sk, SolveRef<T> ref,
                                                      int64_t lumpBegin, int64_t lumpEnd) {
  const int64_t l = lumpBegin + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (l >= lumpEnd) return;
  const int n = (int)(sk.lumpStart[l + 1] - sk.lumpStart[l]);
  if (n > kElimSmallMax) return;
  GP<const T> D = solveMat(ref) + sk.chainData[sk.chainColPtr[l]];
  GP<T> xl = solveVec(ref) + sk.lumpStart[l];
  T x[kElimSmallMax];
  for (int i = 0; i < n; i++) x[i] = xl[i];
  for (int i = 0; i < n; i++) {
    T s = x[i];
    for (int j = 0; j < i; j++) s -= D[i * n + j] * x[j];
    x[i] = s / D[i * n + i];
  }
  for (int i = 0; i < n; i++) xl[i] = x[i];
}

// K-S2: y[span] -= sum over the blocks B that sit in that row span, B * x_lump.  The blocks are
// listed per target span (SolveGatherEntry, sorted by span), an item is <= 256 entries of one
// span: thread per block, workgroup reduction, ONE atomic per row and item (instead of one per
// row and block, all on the few camera rows).
template <typename T>
__global__ __launch_bounds__(256) void solveElimGatherL(const SolveGatherItem* items,
                                                        const SolveGatherEntry* entries,
                                                        SolveRef<T> ref) {
  __shared__ T part[4];
  const SolveGatherItem it = items[blockIdx.x];
  GP<const T> data = solveMat(ref);
  GP<T> vec = solveVec(ref);
  const int e = it.entryBegin + (int)threadIdx.x;
  const bool live = e < it.entryEnd;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int n = 0;
  GP<const T> B = data;
  T x[kElimSmallMax];
  if (live) {
    const SolveGatherEntry en = entries[e];
    n = en.n;
    B = data + en.dataOff;
    GP<const T> xl = vec + en.xOff;
    for (int k = 0; k < n; k++) x[k] = xl[k];
  }
  for (int r = 0; r < it.rows; r++) {
    T s = T(0);
    for (int k = 0; k < n; k++) s += B[r * n + k] * x[k];
    s = waveSum(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(vec + it.rowStart + r, part[0] + part[1] + part[2] + part[3]);
    __syncthreads();
  }
}

// ---- dense panels ----------------------------------------------------------------------------
// triangular solve with the nb x nb diagonal block of a panel, one workgroup per panel: 256
// threads stage L (batched, coalesced loads), then wave 0 solves with x_i in lane i and the
// pivot value broadcast by readlane; the block is padded to 64 x 64 with the identity so the
// 64-step loop unrolls completely and the LDS reads do not sit on the dependency chain.
__device__ __forceinline__ double laneBcast(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float laneBcast(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) void solveTriPanel(const PanelDesc* panels,
                                                     const int32_t* levelPanels,
                                                     SolveRef<T> ref) {
  constexpr int NB = kPanelWidth, LD = NB + 1;
  __shared__ T Ls[NB * LD];
  const PanelDesc pd = panels[levelPanels[blockIdx.x]];
  GP<const T> A = solveMat(ref) + pd.diagOff;
  GP<T> x = solveVec(ref) + pd.vecOff;
  const int nb = pd.nb, lda = pd.lda, tid = threadIdx.x;
  {
    T v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      v[i] = (r < nb && c <= r) ? A[(int64_t)r * lda + c] : (r == c ? T(1) : T(0));
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      Ls[r * LD + c] = v[i];
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  const int lane = tid;
  T xi = lane < nb ? x[lane] : T(0);
  const T inv = T(1) / Ls[lane * LD + lane];
  if (!BACKWARD) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const T lij = Ls[lane * LD + j];  // column j of L, lane = row
      const T xj = laneBcast(xi * inv, j);
      xi = lane == j ? xj : (lane > j ? xi - lij * xj : xi);
    }
  } else {
#pragma unroll
    for (int j = NB - 1; j >= 0; j--) {
      const T lji = Ls[j * LD + lane];  // row j of L, lane = column
      const T xj = laneBcast(xi * inv, j);
      xi = lane == j ? xj : (lane < j ? xi - lji * xj : xi);
    }
  }
  if (lane < nb) x[lane] = xi;
}

__device__ __forceinline__ int solveTargetRow(const PanelDesc& pd, const int32_t* rowGlobal, int q) {
  // below-row q of a panel -> row index in the full vector
  return q < pd.nRest ? pd.vecOff + pd.nb + q : rowGlobal[pd.lumpRowBase + (q - pd.nRest)];
}

// forward: x[target(q)] -= P[q][:] . x_p for a 64-row tile.  16 lanes share a row (4 columns
// each), a wave covers 4 rows per step and 16 rows in all; every load is issued before the
// reductions start.
template <typename T>
__global__ __launch_bounds__(256) void solveGemvL(const PanelDesc* panels, const TrsmTask* tasks,
                                                  const int32_t* rowGlobal, SolveRef<T> ref) {
  const TrsmTask task = tasks[blockIdx.x];
  const PanelDesc pd = panels[task.panel];
  GP<const T> data = solveMat(ref);
  GP<T> vec = solveVec(ref);
  const int nb = pd.nb, lda = pd.lda, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = (lane & 15) * 4, sub = lane >> 4;
  GP<const T> P = data + pd.diagOff + (int64_t)nb * lda;
  T xk[4];
#pragma unroll
  for (int i = 0; i < 4; i++) xk[i] = k0 + i < nb ? vec[pd.vecOff + k0 + i] : T(0);
  const int rows = min(kTile, pd.rowsBelow - task.rowTile);
  T p[4][4];
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int r = wave * 16 + it * 4 + sub;
    GP<const T> row = P + (int64_t)(task.rowTile + r) * lda + k0;
#pragma unroll
    for (int i = 0; i < 4; i++) p[it][i] = (r < rows && k0 + i < nb) ? row[i] : T(0);
  }
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int r = wave * 16 + it * 4 + sub;
    T s = p[it][0] * xk[0] + p[it][1] * xk[1] + p[it][2] * xk[2] + p[it][3] * xk[3];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((lane & 15) == 0 && r < rows) {
      atomicSub(vec + solveTargetRow(pd, rowGlobal, task.rowTile + r), s);
    }
  }
}

// backward: x_p[k] -= sum_q P[q][k] * x[target(q)] over a 64-row tile (lane = column k, a wave
// takes 16 rows, loads issued up front)
template <typename T>
__global__ __launch_bounds__(256) void solveGemvLt(const PanelDesc* panels, const TrsmTask* tasks,
                                                   const int32_t* rowGlobal, SolveRef<T> ref) {
  __shared__ T part[4][kPanelWidth];
  const TrsmTask task = tasks[blockIdx.x];
  const PanelDesc pd = panels[task.panel];
  GP<const T> data = solveMat(ref);
  GP<T> vec = solveVec(ref);
  const int nb = pd.nb, lda = pd.lda, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GP<const T> P = data + pd.diagOff + (int64_t)nb * lda;
  const int rows = min(kTile, pd.rowsBelow - task.rowTile);
  T p[16], xq[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int r = wave * 16 + i, q = task.rowTile + r;
    const bool ok = r < rows;
    xq[i] = ok ? vec[solveTargetRow(pd, rowGlobal, q)] : T(0);
    p[i] = (ok && lane < nb) ? P[(int64_t)q * lda + lane] : T(0);
  }
  T acc = T(0);
#pragma unroll
  for (int i = 0; i < 16; i++) acc += p[i] * xq[i];
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < nb) {
    atomicSub(vec + pd.vecOff + lane, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
  }
}

}  // namespace hipk
}  // namespace BaSpaCho
