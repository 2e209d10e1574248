// HIP kernels of the triangular solves (Solver::solve / solveL / solveLt), correctness-first.
// They reuse the factor plan: panels, levels and the 64-row task tiles.  Replaces the solve part of
// MatOpsCuda.cu (sparseElim_* kernels :883-1012, assembleVec(T) :836-880, cublas trsm/gemm calls
// :1093-1181).  One right-hand side per blockIdx.y; vec is column-major (order x nRHS, ld = ldc).
#pragma once

#include <hip/hip_runtime.h>

#include "hip_kernels.h"

namespace BaSpaCho {
namespace hipk {

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
__global__ __launch_bounds__(256) void solveElimSmall(SkelDev sk, const T* data, T* vecAll,
                                                      int64_t ldc, int64_t lumpBegin,
                                                      int64_t lumpEnd) {
  const int64_t l = lumpBegin + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (l >= lumpEnd) return;
  const int n = (int)(sk.lumpStart[l + 1] - sk.lumpStart[l]);
  if (n > kElimSmallMax) return;  // wide lumps go through the panel kernels
  T* vec = vecAll + (int64_t)blockIdx.y * ldc;
  const int64_t c0 = sk.chainColPtr[l], cEnd = sk.chainColPtr[l + 1];
  const int64_t diagCh = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
  const T* D = data + sk.chainData[c0];
  T* xl = vec + sk.lumpStart[l];
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
      const T* B = data + sk.chainData[c];
      T* y = vec + sk.spanStart[span];
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
      const T* B = data + sk.chainData[c];
      const T* y = vec + sk.spanStart[span];
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

// ---- dense panels ----------------------------------------------------------------------------
// triangular solve with the nb x nb diagonal block of a panel, one workgroup per panel
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(64) void solveTriPanel(const PanelDesc* panels,
                                                    const int32_t* levelPanels, const T* data,
                                                    T* vecAll, int64_t ldc) {
  constexpr int LD = kPanelWidth + 1;
  __shared__ T Ls[kPanelWidth * LD];
  __shared__ T xs[kPanelWidth];
  const PanelDesc pd = panels[levelPanels[blockIdx.x]];
  const T* A = data + pd.diagOff;
  T* x = vecAll + (int64_t)blockIdx.y * ldc + pd.vecOff;
  const int nb = pd.nb, lda = pd.lda, lane = threadIdx.x;
  for (int i = 0; i < nb; i++) {
    if (lane <= i) Ls[i * LD + lane] = A[(int64_t)i * lda + lane];
  }
  if (lane < nb) xs[lane] = x[lane];
  __syncthreads();
  if (!BACKWARD) {
    for (int j = 0; j < nb; j++) {
      const T xj = xs[j] / Ls[j * LD + j];
      __syncthreads();
      if (lane == j) xs[j] = xj;
      if (lane > j && lane < nb) xs[lane] -= Ls[lane * LD + j] * xj;
      __syncthreads();
    }
  } else {
    for (int j = nb - 1; j >= 0; j--) {
      const T xj = xs[j] / Ls[j * LD + j];
      __syncthreads();
      if (lane == j) xs[j] = xj;
      if (lane < j) xs[lane] -= Ls[j * LD + lane] * xj;
      __syncthreads();
    }
  }
  if (lane < nb) x[lane] = xs[lane];
}

__device__ __forceinline__ int solveTargetRow(const PanelDesc& pd, const int32_t* rowGlobal, int q) {
  // below-row q of a panel -> row index in the full vector
  return q < pd.nRest ? pd.vecOff + pd.nb + q : rowGlobal[pd.lumpRowBase + (q - pd.nRest)];
}

// forward: x[target(q)] -= P[q][:] . x_p for a 64-row tile (wave per row, lanes over k)
template <typename T>
__global__ __launch_bounds__(256) void solveGemvL(const PanelDesc* panels, const TrsmTask* tasks,
                                                  const int32_t* rowGlobal, const T* data,
                                                  T* vecAll, int64_t ldc) {
  const TrsmTask task = tasks[blockIdx.x];
  const PanelDesc pd = panels[task.panel];
  T* vec = vecAll + (int64_t)blockIdx.y * ldc;
  const int nb = pd.nb, lda = pd.lda, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* P = data + pd.diagOff + (int64_t)nb * lda;
  const T xk = lane < nb ? vec[pd.vecOff + lane] : T(0);
  const int rows = min(kTile, pd.rowsBelow - task.rowTile);
  for (int r = wave; r < rows; r += 4) {
    const int q = task.rowTile + r;
    const T p = lane < nb ? P[(int64_t)q * lda + lane] : T(0);
    const T s = waveSum(p * xk);
    if (lane == 0) atomicSub(vec + solveTargetRow(pd, rowGlobal, q), s);
  }
}

// backward: x_p[k] -= sum_q P[q][k] * x[target(q)] over a 64-row tile
template <typename T>
__global__ __launch_bounds__(256) void solveGemvLt(const PanelDesc* panels, const TrsmTask* tasks,
                                                   const int32_t* rowGlobal, const T* data,
                                                   T* vecAll, int64_t ldc) {
  __shared__ T part[4][kPanelWidth];
  const TrsmTask task = tasks[blockIdx.x];
  const PanelDesc pd = panels[task.panel];
  T* vec = vecAll + (int64_t)blockIdx.y * ldc;
  const int nb = pd.nb, lda = pd.lda, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* P = data + pd.diagOff + (int64_t)nb * lda;
  const int rows = min(kTile, pd.rowsBelow - task.rowTile);
  T acc = T(0);
  for (int r = wave; r < rows; r += 4) {
    const int q = task.rowTile + r;
    const T xq = vec[solveTargetRow(pd, rowGlobal, q)];
    if (lane < nb) acc += P[(int64_t)q * lda + lane] * xq;
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < nb) {
    atomicSub(vec + pd.vecOff + lane, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
  }
}

}  // namespace hipk
}  // namespace BaSpaCho
