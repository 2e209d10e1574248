// Backward pass over a sparse-elimination range with the RIGHT-HAND SIDES across the lanes (round 6).
// Replaces, for 2 .. 16 right-hand sides per launch row, K-S3m of hip_solve_kernels.h (reference:
// sparseElim_subDiagMultT + sparseElim_diagSolveLt, MatOpsCuda.cu:949-1012, one thread per lump and
// right-hand side).
//
// K-S3m gives a lump 16 lanes = 16 ROWS of a block and keeps 8 right-hand sides in registers: the 9 x 3
// blocks of bundle adjustment use 9 lanes of 16, ten right-hand sides read the point columns twice, and
// every y value is a load of its own from a column-major vector (1.80 ms of a 3.95 ms solve with ten
// right-hand sides on BAL-871: 68 clocks per wave load, latency of dependent round trips).  Here:
//   - the rows below the range (the camera part: final when the backward pass reaches the range) are
//     first copied span by span into blocks [16][rows of the span] (K-S3t), so that lane c of a lump's 16
//     lanes finds the rows of a block for ITS right-hand side as consecutive words -- two per load, the 16
//     lanes together one contiguous piece;
//   - the lane group holds a block of L as its raw, contiguous words (lane s: words 2 s and 2 s + 1: one
//     load per block), and the product  acc[k] += B[r][k] * y[r][c]  is ONE instruction:
//     v_fmac_f64 with the DPP modifier row_newbcast, which feeds word r * n + k of the group to all of
//     its lanes -- no shuffle, no LDS, no select;
//   - 16 right-hand sides per pass over L (the blocks are read once for up to 16).
#pragma once

#include <hip/hip_runtime.h>

#include "hip_solve_kernels.h"

namespace BaSpaCho {
namespace hipk {

// acc += (word of lane E of the 16-lane row holding b) * y
// (inline assembly is invisible to the compiler's hazard recogniser: a DPP operand written by a vector
//  instruction needs two wait states before it is read -- FRESH = this is the first use of b since it was
//  produced, e.g. by the select that masks a load)
template <int E, bool FRESH = false>
__device__ __forceinline__ void fmaRowBcast(double& acc, double b, double y) {
  static_assert(E >= 0 && E < 16, "lane of a DPP row");
  if constexpr (FRESH) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(b), "v"(y), "n"(E));
  } else {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(b), "v"(y), "n"(E));
  }
}
template <int E, bool FRESH = false>
__device__ __forceinline__ void fmaRowBcast(float& acc, float b, float y) {
  static_assert(E >= 0 && E < 16, "lane of a DPP row");
  if constexpr (FRESH) {
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(b), "v"(y), "n"(E));
  } else {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(b), "v"(y), "n"(E));
  }
}
// word E of the row, in every lane of the row
template <int E, typename T>
__device__ __forceinline__ T rowBcast(T b) {
  T r = T(0);
  fmaRowBcast<E, true>(r, b, T(1));
  return r;
}

constexpr int kWideRhs = 16;  // right-hand sides per lane group = lanes of a DPP row

// two consecutive values with ONE load (16 bytes from an 8-byte-aligned address: global memory takes it)
template <typename T>
struct WideTwo {
  typedef T type __attribute__((ext_vector_type(2), aligned(sizeof(T))));
};

// elements of the wide copy per group of 16 right-hand sides and batch entry (one pair of slack: the last
// lane's last pair may reach past its rows)
__host__ __device__ inline int64_t wideGroupStride(int64_t nRows) { return nRows * kWideRhs + 16; }

// K-S3t: the rows from row0 on (spans span0 .. span0 + numSpans) of up to 16 right-hand sides ->
// out[group]: per SPAN a block [16][rows of the span], so that the lane of right-hand side c finds the rows
// of a block of L as consecutive words (pairs per load).  One wave per span; blockIdx.y = group of 16,
// blockIdx.z = batch entry; zero-filled past the last right-hand side.
template <typename T>
__global__ __launch_bounds__(256) void solveRowsToWide(SolveRef<T> ref, int nRhs, const int64_t* spanStart,
                                                       int64_t span0, int64_t numSpans, int64_t row0,
                                                       int64_t nRows, T* out) {
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= numSpans) return;
  const int lane = threadIdx.x & 63;
  const int rhs0 = kWideRhs * blockIdx.y, nR = min(kWideRhs, nRhs - rhs0);
  const int64_t s0 = spanStart[span0 + w];
  const int rows = (int)(spanStart[span0 + w + 1] - s0);
  GP<const T> vec = solveVecBase(ref) + (int64_t)rhs0 * ref.ldc + s0;
  T* dst = out + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * wideGroupStride(nRows) +
           (s0 - row0) * kWideRhs;
  for (int t = lane; t < rows * kWideRhs; t += 64) {
    const int c = t / rows, r = t - c * rows;
    dst[t] = c < nR ? vec[(int64_t)c * ref.ldc + r] : T(0);
  }
}

// rows of a block handled per pass: rows * N words <= 32 = one pair of words per lane of the group
template <int N>
struct WideChunk {
  static constexpr int kRows = N == 1 ? 16 : N == 2 ? 16 : N == 3 ? 10 : 8;
};

// products of one chunk: R <= kRows rows of a block that is `rows` tall, starting at word `off` of the
// data; yRows = the lane's words for the block's span in the wide copy, at the chunk's first row
template <typename T, int N>
struct WidePiece {
  static constexpr int RM = WideChunk<N>::kRows;
  using Two = typename WideTwo<T>::type;
  Two v, y[RM / 2];
  __device__ __forceinline__ void load(GP<const T> data, int64_t off, const T* yRows, int R, int c) {
    const int words = R * N;
    v = Two{T(0), T(0)};
    // (words past the chunk are never fetched: the last block of a matrix ends where its allocation ends)
    if (2 * c + 1 < words) {
      v = *(GP<const Two>)(data + off + 2 * c);
    } else if (2 * c < words) {
      v.x = data[off + 2 * c];
    }
#pragma unroll
    for (int j = 0; j < RM / 2; j++) {
      // (pairs past the chunk: the next lane's words or the slack of the copy -- masked below)
      const Two t = *(const Two*)(yRows + 2 * j);
      y[j].x = 2 * j < R ? t.x : T(0);
      y[j].y = 2 * j + 1 < R ? t.y : T(0);
    }
  }
  template <int E>
  __device__ __forceinline__ void one(T& acc, T yv) const {
    if constexpr ((E & 1) == 0) {
      T b = v.x;
      fmaRowBcast<E / 2, E == 0>(acc, b, yv);
    } else {
      T b = v.y;
      fmaRowBcast<E / 2, E == 1>(acc, b, yv);
    }
  }
  template <int R, int K>
  __device__ __forceinline__ void step(T (&acc)[N]) const {
    if constexpr (R < RM) {
      one<R * N + K>(acc[K], (R & 1) ? y[R / 2].y : y[R / 2].x);
      if constexpr (K + 1 < N) {
        step<R, K + 1>(acc);
      } else {
        step<R + 1, 0>(acc);
      }
    }
  }
  __device__ __forceinline__ void multiply(T (&acc)[N]) const { step<0, 0>(acc); }
};

// K-S3w: x_l <- L_ll^-T (x_l - sum over the blocks B of the lump, B^T y[rows of B]) for the lumps of a
// range that are all N wide.  16 lanes per lump (lane = right-hand side), four lumps per wave, 16 per
// workgroup; blockIdx.y = group of 16 right-hand sides, blockIdx.z = batch entry.  yW = K-S3t's copy of
// the rows from yRow0 on.
template <typename T, int N>
__global__ __launch_bounds__(256) void solveElimLumpsLtWide(const SolveLumpDesc* descs,
                                                            const SolveLumpBlock* blocks,
                                                            SolveRef<T> ref, int numLumps, int nRhs,
                                                            const T* yWide, int64_t yRow0,
                                                            int64_t yRows) {
  using Piece = WidePiece<T, N>;
  constexpr int RM = Piece::RM;
  const int lane = threadIdx.x & 63, c = lane & 15;
  const int idx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool live = idx < numLumps;
  const SolveLumpDesc ld = descs[live ? idx : numLumps - 1];
  GP<const T> data = solveMat(ref);
  const int rhs0 = kWideRhs * blockIdx.y, nR = min(kWideRhs, nRhs - rhs0);
  // (yW[row * 16 + c * rows of the span + r]: word r of the lane's right-hand side in the span at `row`)
  const T* yW = yWide + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * wideGroupStride(yRows) -
                yRow0 * kWideRhs;
  GP<T> xcol = solveVecBase(ref) + (int64_t)(rhs0 + min(c, nR - 1)) * ref.ldc + ld.xOff;
  T acc[N];
#pragma unroll
  for (int k = 0; k < N; k++) acc[k] = T(0);
  const int nBlocks = live ? ld.blockEnd - ld.blockBegin : 0;
  // two blocks per trip, all their loads issued before the first product; the descriptors of the next
  // trip are requested a trip ahead (the same address in the 16 lanes of a group: one fetch)
  const SolveLumpBlock* bl = blocks + ld.blockBegin;
  const int last = max(nBlocks - 1, 0);
  SolveLumpBlock n0 = {0, 0, 0}, n1 = {0, 0, 0};
  if (nBlocks > 0) {
    n0 = bl[0];
    n1 = bl[min(1, last)];
  }
  // x_l and the diagonal block: requested before the loop, used after it
  T xin[N];
#pragma unroll
  for (int k = 0; k < N; k++) xin[k] = xcol[k];
  const T dd = c < N * N ? data[ld.diagOff + c] : T(0);
  bool tall = false;
  for (int e0 = 0; e0 < nBlocks; e0 += 2) {
    const SolveLumpBlock b0 = n0, b1 = n1;
    const bool two = e0 + 1 < nBlocks;
    n0 = bl[min(e0 + 2, last)];
    n1 = bl[min(e0 + 3, last)];
    Piece p0, p1;
    p0.load(data, b0.dataOff, yW + (int64_t)b0.yOff * kWideRhs + c * b0.rows, min(b0.rows, RM), c);
    p1.load(data, b1.dataOff, yW + (int64_t)b1.yOff * kWideRhs + c * b1.rows, two ? min(b1.rows, RM) : 0, c);
    tall = tall || b0.rows > RM || (two && b1.rows > RM);
    p0.multiply(acc);
    p1.multiply(acc);
  }
  if (__any(tall)) {  // rows kRows .. of taller blocks (not the 9 x 3 blocks of bundle adjustment)
    for (int e = 0; e < nBlocks; e++) {
      const SolveLumpBlock b = bl[e];
      for (int r0 = RM; r0 < b.rows; r0 += RM) {
        Piece p;
        p.load(data, b.dataOff + (int64_t)r0 * N, yW + (int64_t)b.yOff * kWideRhs + c * b.rows + r0,
               min(b.rows - r0, RM), c);
        p.multiply(acc);
      }
    }
  }
  // back substitution with the upper triangle L_ll^T, every lane for its right-hand side; d[i][j] =
  // word i * N + j of the diagonal block, broadcast from the lane that fetched it
  T x[N];
#pragma unroll
  for (int k = 0; k < N; k++) x[k] = xin[k] - acc[k];
  T d[N][N];
  {
    T* dp = &d[0][0];
    // (static lane indices: N * N <= 16)
    if constexpr (N >= 1) dp[0] = rowBcast<0>(dd);
    if constexpr (N >= 2) {
      dp[1] = rowBcast<1>(dd);
      dp[2] = rowBcast<2>(dd);
      dp[3] = rowBcast<3>(dd);
    }
    if constexpr (N >= 3) {
      dp[4] = rowBcast<4>(dd);
      dp[5] = rowBcast<5>(dd);
      dp[6] = rowBcast<6>(dd);
      dp[7] = rowBcast<7>(dd);
      dp[8] = rowBcast<8>(dd);
    }
    if constexpr (N >= 4) {
      dp[9] = rowBcast<9>(dd);
      dp[10] = rowBcast<10>(dd);
      dp[11] = rowBcast<11>(dd);
      dp[12] = rowBcast<12>(dd);
      dp[13] = rowBcast<13>(dd);
      dp[14] = rowBcast<14>(dd);
      dp[15] = rowBcast<15>(dd);
    }
  }
#pragma unroll
  for (int j = N - 1; j >= 0; j--) {
    T s = x[j];
#pragma unroll
    for (int i = j + 1; i < N; i++) s -= d[i][j] * x[i];
    x[j] = s / d[j][j];
  }
  if (live && c < nR) {
#pragma unroll
    for (int k = 0; k < N; k++) xcol[k] = x[k];
  }
}

}  // namespace hipk
}  // namespace BaSpaCho
