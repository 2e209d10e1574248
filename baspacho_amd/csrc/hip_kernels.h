// Hand-written HIP kernels of the MI355X (gfx950, CDNA4) numeric factor path.
// wave = 64 lanes; fp64 MFMA v_mfma_f64_16x16x4_f64 for the rank-nb update; LDS-staged
// panels; native fp64 atomics (global_atomic_add_f64) for concurrent scatter targets.
// Each kernel names the reference op it replaces (file:line in /root/reference/baspacho).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "hip_plan.h"

namespace BaSpaCho {
namespace hipk {

// One matrix (single) or a batch of identical-structure matrices (many, indexed by blockIdx.y);
// replaces the Plain/Batched policy structs of MatOpsCuda.cu:345-368.
template <typename T>
struct DataRef {
  T* single;
  T* const* many;
};

template <typename T>
__device__ __forceinline__ T* pickData(const DataRef<T>& d) {
  return d.many ? d.many[blockIdx.y] : d.single;
}

__device__ __forceinline__ void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void atomicSub(double* p, double v) { unsafeAtomicAdd(p, -v); }
__device__ __forceinline__ void atomicSub(float* p, float v) { unsafeAtomicAdd(p, -v); }

// device view of the int64 skeleton arrays (also what deviceAccessor() hands out)
struct SkelDev {
  const int64_t* spanStart;
  const int64_t* spanToLump;
  const int64_t* lumpStart;
  const int64_t* spanOffsetInLump;
  const int64_t* chainColPtr;
  const int64_t* chainRowSpan;
  const int64_t* chainData;
  const int64_t* chainRowsTillEnd;
  const int64_t* boardColPtr;
  const int64_t* boardChainColOrd;
};

// ------------------------------------------------------------------------------------------
// K1  sparse-elimination lump factor: potrf of the small diagonal block + solve of every row
// below it.  Replaces factor_lumps_kernel (MatOpsCuda.cu:148-186: one THREAD per lump, scalar
// loops) by one WAVE per lump: the n x n Cholesky runs lane-parallel in LDS, then each lane
// owns rows lane, lane+64, ... of the panel (coalesced: consecutive lanes read consecutive
// rows of n contiguous values).  NMAX = compile-time bound on n (4, 8 or 16).
// ------------------------------------------------------------------------------------------
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void elimFactorSmall(SkelDev sk, DataRef<T> dref,
                                                       int64_t lumpBegin, int64_t lumpEnd) {
  constexpr int LD = NMAX + 1;
  __shared__ T diagS[4][NMAX * LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t l = lumpBegin + (int64_t)blockIdx.x * 4 + wave;
  if (l >= lumpEnd) return;
  const int n = (int)(sk.lumpStart[l + 1] - sk.lumpStart[l]);
  if (n > NMAX) return;  // wide lumps of the range go through the panel kernels
  T* data = pickData(dref);
  const int64_t c0 = sk.chainColPtr[l];
  const int64_t nCh = sk.chainColPtr[l + 1] - c0;
  const int64_t diagCh = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
  T* D = data + sk.chainData[c0];
  T* B = data + sk.chainData[c0 + diagCh];
  const int rowsBelow =
      (int)(sk.chainRowsTillEnd[c0 + nCh - 1] - sk.chainRowsTillEnd[c0 + diagCh - 1]);
  T* S = diagS[wave];

  for (int e = lane; e < n * n; e += 64) {
    int i = e / n, j = e - i * n;
    S[i * LD + j] = D[e];
  }
  waveSync();
  // right-looking Cholesky, lanes over rows / trailing pairs
  for (int j = 0; j < n; j++) {
    const T d = sqrt(S[j * LD + j]);
    waveSync();
    if (lane == 0) S[j * LD + j] = d;
    if (lane > j && lane < n) S[lane * LD + j] /= d;
    waveSync();
    const int rem = n - j - 1;
    for (int e = lane; e < rem * rem; e += 64) {
      int a = e / rem, b = e - a * rem;
      if (b <= a) S[(j + 1 + a) * LD + (j + 1 + b)] -= S[(j + 1 + a) * LD + j] * S[(j + 1 + b) * LD + j];
    }
    waveSync();
  }
  for (int e = lane; e < n * n; e += 64) {
    int i = e / n, j = e - i * n;
    if (j <= i) D[e] = S[i * LD + j];
  }
  // rows below: x * L^T = b  (forward substitution per row)
  for (int r = lane; r < rowsBelow; r += 64) {
    T* row = B + (int64_t)r * n;
    T x[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; j++) x[j] = j < n ? row[j] : T(0);
#pragma unroll
    for (int j = 0; j < NMAX; j++) {
      if (j < n) {
        T s = x[j];
#pragma unroll
        for (int i = 0; i < j; i++) s -= x[i] * S[j * LD + i];
        x[j] = s / S[j * LD + j];
      }
    }
#pragma unroll
    for (int j = 0; j < NMAX; j++) {
      if (j < n) row[j] = x[j];
    }
  }
}

// ------------------------------------------------------------------------------------------
// K2  sparse-elimination update: for column l and every pair of below-diagonal chains i<=j:
//     target(sj,si) -= L(sj,l) * L(si,l)^T.
// Replaces sparse_elim_straight_kernel + do_sparse_elim (MatOpsCuda.cu:235-331: one thread per
// pair, scalar 9x3*3x9 product, CAS-style atomics).  Here one wave owns (column, chain i):
// it walks j = i..end, lanes cover the |sj| x |si| output block, the target chain inside the
// target column is found by a wave-uniform binary search, and the subtraction is one native
// fp64 atomic per element.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void elimUpdate(SkelDev sk, const int32_t* chainLump,
                                                  DataRef<T> dref, int64_t chainBegin,
                                                  int64_t chainEnd) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t c = chainBegin + (int64_t)blockIdx.x * 4 + wave;
  if (c >= chainEnd) return;
  const int64_t l = chainLump[c - chainBegin];
  const int64_t c0 = sk.chainColPtr[l], cEnd = sk.chainColPtr[l + 1];
  const int64_t diagCh = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
  if (c - c0 < diagCh) return;  // diagonal chain: nothing to push
  T* data = pickData(dref);
  const int n = (int)(sk.lumpStart[l + 1] - sk.lumpStart[l]);
  const int64_t si = sk.chainRowSpan[c];
  const int siSize = (int)(sk.spanStart[si + 1] - sk.spanStart[si]);
  const T* Bi = data + sk.chainData[c];
  const int64_t t = sk.spanToLump[si];
  const int64_t tStride = sk.lumpStart[t + 1] - sk.lumpStart[t];
  const int64_t colOff = sk.spanOffsetInLump[si];
  const int64_t t0 = sk.chainColPtr[t], tCount = sk.chainColPtr[t + 1] - t0;

  int64_t lo = 0;  // target chains are visited in increasing order: resume the search
  for (int64_t j = c; j < cEnd; j++) {
    const int64_t sj = sk.chainRowSpan[j];
    const int sjSize = (int)(sk.spanStart[sj + 1] - sk.spanStart[sj]);
    const T* Bj = data + sk.chainData[j];
    int64_t hi = tCount;
    while (hi - lo > 1) {
      int64_t mid = lo + (hi - lo) / 2;
      if (sk.chainRowSpan[t0 + mid] <= sj) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    T* tgt = data + sk.chainData[t0 + lo] + colOff;
    const int total = sjSize * siSize;
    for (int e = lane; e < total; e += 64) {
      const int r = e / siSize, q = e - r * siSize;
      if (j == c && q > r) continue;  // diagonal target block: lower triangle only
      T acc = T(0);
      for (int k = 0; k < n; k++) acc += Bj[r * n + k] * Bi[q * n + k];
      atomicSub(tgt + (int64_t)r * tStride + q, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K3  panel potrf: in-place Cholesky of the nb x nb (nb <= 64) diagonal block of a panel, one
// workgroup per panel, whole block in LDS.  Replaces cusolverDn?potrf / potrfBatched
// (MatOpsCuda.cu:508-548, 727-755) on the panel granularity.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void potrfPanel(const PanelDesc* panels,
                                                  const int32_t* levelPanels, DataRef<T> dref) {
  constexpr int LD = kPanelWidth + 1;
  __shared__ T S[kPanelWidth * LD];
  const PanelDesc pd = panels[levelPanels[blockIdx.x]];
  T* A = pickData(dref) + pd.diagOff;
  const int nb = pd.nb, lda = pd.lda, tid = threadIdx.x;
  for (int e = tid; e < nb * nb; e += 256) {
    int i = e / nb, j = e - i * nb;
    if (j <= i) S[i * LD + j] = A[(int64_t)i * lda + j];
  }
  __syncthreads();
  for (int j = 0; j < nb; j++) {
    const T d = sqrt(S[j * LD + j]);
    __syncthreads();
    if (tid == 0) S[j * LD + j] = d;
    if (tid > j && tid < nb) S[tid * LD + j] /= d;
    __syncthreads();
    const int rem = nb - j - 1;
    for (int e = tid; e < rem * rem; e += 256) {
      int a = e / rem, b = e - a * rem;
      if (b <= a) S[(j + 1 + a) * LD + (j + 1 + b)] -= S[(j + 1 + a) * LD + j] * S[(j + 1 + b) * LD + j];
    }
    __syncthreads();
  }
  for (int e = tid; e < nb * nb; e += 256) {
    int i = e / nb, j = e - i * nb;
    if (j <= i) A[(int64_t)i * lda + j] = S[i * LD + j];
  }
}

// ------------------------------------------------------------------------------------------
// K4  panel trsm: X * L^T = B for a tile of 64 rows below the panel's diagonal block.
// One wave per task; L and the row tile live in LDS (tile transposed, xs[j][row], padded so
// that both the coalesced fill and the per-lane column walk are bank-conflict free).
// Replaces cublas?trsm LEFT/UPPER/OP_C (MatOpsCuda.cu:550-566, 757-781).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void trsmPanel(const PanelDesc* panels, const TrsmTask* tasks,
                                                DataRef<T> dref) {
  constexpr int LDL = kPanelWidth + 1, LDX = kTile + 1;
  __shared__ T Ls[kPanelWidth * LDL];
  __shared__ T xs[kPanelWidth * LDX];
  const TrsmTask task = tasks[blockIdx.x];
  const PanelDesc pd = panels[task.panel];
  T* data = pickData(dref);
  const T* A = data + pd.diagOff;
  const int nb = pd.nb, lda = pd.lda, lane = threadIdx.x;
  T* P = data + pd.diagOff + (int64_t)(nb + task.rowTile) * lda;
  const int rows = min(kTile, pd.rowsBelow - task.rowTile);

  for (int e = lane; e < nb * nb; e += 64) {
    int i = e / nb, j = e - i * nb;
    if (j <= i) Ls[i * LDL + j] = A[(int64_t)i * lda + j];
  }
  for (int e = lane; e < rows * nb; e += 64) {
    int r = e / nb, j = e - r * nb;
    xs[j * LDX + r] = P[(int64_t)r * lda + j];
  }
  __syncthreads();
  if (lane < rows) {
    for (int j = 0; j < nb; j++) {
      T s0 = xs[j * LDX + lane], s1 = T(0);
      int i = 0;
      for (; i + 1 < j; i += 2) {
        s0 -= xs[i * LDX + lane] * Ls[j * LDL + i];
        s1 -= xs[(i + 1) * LDX + lane] * Ls[j * LDL + i + 1];
      }
      if (i < j) s0 -= xs[i * LDX + lane] * Ls[j * LDL + i];
      xs[j * LDX + lane] = (s0 + s1) / Ls[j * LDL + j];
    }
  }
  __syncthreads();
  for (int e = lane; e < rows * nb; e += 64) {
    int r = e / nb, j = e - r * nb;
    P[(int64_t)r * lda + j] = xs[j * LDX + r];
  }
}

// ------------------------------------------------------------------------------------------
// K5  rank-nb update with fused scatter:  target -= B_rows * B_cols^T  on one 64x64 tile of
// the lower trapezoid of one segment (panel -> target lump).  Replaces the cublas?gemm into a
// temp buffer (MatOpsCuda.cu:568-590), prepareAssemble's per-target table upload (:471-481)
// and assemble_kernel (:370-406): the product never touches HBM, the accumulator is scattered
// straight from the MFMA registers.
//   * 256 threads = 4 waves, each wave a 32x32 sub-tile = 2x2 MFMA 16x16 tiles
//   * K = nb <= 64 staged in one shot; LDS row stride 66 (== 2 mod 4 in doubles) makes the
//     MFMA operand fetch (lane l reads [l&15][k0 + (l>>4)]) bank-conflict free
//   * fp64: v_mfma_f64_16x16x4_f64, C layout col = lane&15, row = (lane>>4) + 4*reg
//   * fp32: v_mfma_f32_16x16x4_f32, C layout col = lane&15, row = 4*(lane>>4) + reg
// ------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
  using Acc = double4_t;
  static __device__ __forceinline__ Acc run(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Mfma<float> {
  using Acc = float4_t;
  static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <typename T>
__global__ __launch_bounds__(256) void updateTile(const PanelDesc* panels, const SegDesc* segs,
                                                  const UpdTask* tasks, const int64_t* chainOffTab,
                                                  const int32_t* rowChain, const int32_t* rowLocal,
                                                  const int32_t* rowColOff, DataRef<T> dref) {
  constexpr int LD = kPanelWidth + 2;
  __shared__ T As[kTile * LD];
  __shared__ T Bs[kTile * LD];
  __shared__ int64_t rowBase[kTile];
  __shared__ int32_t colOff[kTile];

  const UpdTask task = tasks[blockIdx.x];
  const SegDesc sd = segs[task.seg];
  const PanelDesc pd = panels[sd.panel];
  T* data = pickData(dref);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = pd.nb, lda = pd.lda;
  const int kPad = (nb + 3) & ~3;
  const T* P = data + pd.diagOff + (int64_t)nb * lda;  // first row below the diagonal block
  const bool diagTile = task.rowTile == task.colTile;
  const int segEnd = sd.q0 + sd.m;

  // stage the two row tiles (rows beyond the panel / segment and the K padding are zero)
  for (int e = tid; e < kTile * kPad; e += 256) {
    const int r = e / kPad, k = e - r * kPad;
    const int qa = task.rowTile + r;
    T va = T(0);
    if (k < nb && qa < pd.rowsBelow) va = P[(int64_t)qa * lda + k];
    As[r * LD + k] = va;
    if (!diagTile) {
      const int qb = task.colTile + r;
      T vb = T(0);
      if (k < nb && qb < segEnd) vb = P[(int64_t)qb * lda + k];
      Bs[r * LD + k] = vb;
    }
  }
  // per-row / per-column target addressing of this tile
  if (tid < kTile) {
    const int q = task.rowTile + tid;
    int64_t base = 0;
    if (q < pd.rowsBelow) {
      if (sd.kind == kSegIntra) {
        base = sd.tgtBase + (int64_t)q * sd.tgtStride;
      } else {
        const int rr = pd.lumpRowBase + (q - pd.nRest);
        base = chainOffTab[sd.chainTabPtr + (rowChain[rr] - sd.firstChainOrd)] +
               (int64_t)rowLocal[rr] * sd.tgtStride;
      }
    }
    rowBase[tid] = base;
  } else if (tid < 2 * kTile) {
    const int cidx = tid - kTile;
    const int q = task.colTile + cidx;
    int32_t off = 0;
    if (q < segEnd) off = sd.kind == kSegIntra ? q : rowColOff[pd.lumpRowBase + (q - pd.nRest)];
    colOff[cidx] = off;
  }
  __syncthreads();

  const T* Bt = diagTile ? As : Bs;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  using Acc = typename Mfma<T>::Acc;
  Acc acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  // a diagonal tile only needs sub-tiles on or below the diagonal
  const bool skipUpper = diagTile && wr < wc;
  if (!skipUpper) {
    for (int k0 = 0; k0 < kPad; k0 += 4) {
      const T a0 = As[(wr + li) * LD + k0 + lk];
      const T a1 = As[(wr + 16 + li) * LD + k0 + lk];
      const T b0 = Bt[(wc + li) * LD + k0 + lk];
      const T b1 = Bt[(wc + 16 + li) * LD + k0 + lk];
      acc00 = Mfma<T>::run(a0, b0, acc00);
      acc01 = Mfma<T>::run(a0, b1, acc01);
      acc10 = Mfma<T>::run(a1, b0, acc10);
      acc11 = Mfma<T>::run(a1, b1, acc11);
    }
    auto scatter = [&](const Acc& acc, int r0, int c0) {
      const int cIn = c0 + li;
      const int qc = task.colTile + cIn;
      if (qc >= segEnd) return;
      const int32_t co = colOff[cIn];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int rIn = r0 + Mfma<T>::row(lane, reg);
        const int qr = task.rowTile + rIn;
        if (qr < pd.rowsBelow && qr >= qc) {
          T* p = data + rowBase[rIn] + co;
          if (task.atomic) {
            atomicSub(p, acc[reg]);
          } else {
            *p -= acc[reg];
          }
        }
      }
    };
    scatter(acc00, wr, wc);
    scatter(acc01, wr, wc + 16);
    scatter(acc10, wr + 16, wc);
    scatter(acc11, wr + 16, wc + 16);
  }
}

}  // namespace hipk
}  // namespace BaSpaCho
