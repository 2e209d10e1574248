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
constexpr int kSolveBlock = 256;  // columns per step of the wide-lump solve (= the factor's outer block)

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

// right-hand side 0 of this batch entry (kernels that take several right-hand sides per workgroup)
template <typename T>
__device__ __forceinline__ GP<T> solveVecBase(const SolveRef<T>& r) {
  return (GP<T>)(r.vecs ? r.vecs[blockIdx.z] : r.vec);
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

// Sum N per-lane values across the 64 lanes, N sums at once: a butterfly in which every exchange
// also halves the number of values a lane carries (N/2 + N/4 + ... + 1 + log2(64/N) shuffles
// instead of 6 N).  The sum of v[u] ends up in the lanes with ((lane >> SHIFT) & (N-1)) == u,
// SHIFT = 6 - log2(N); returned value = that lane's sum.
template <typename T>
__device__ __forceinline__ T waveSum16(T (&v)[16], int lane) {
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
  T a[8], b[4], c[2];
#pragma unroll
  for (int j = 0; j < 8; j++) a[j] = (b5 ? v[j + 8] : v[j]) + __shfl_xor(b5 ? v[j] : v[j + 8], 32, 64);
#pragma unroll
  for (int j = 0; j < 4; j++) b[j] = (b4 ? a[j + 4] : a[j]) + __shfl_xor(b4 ? a[j] : a[j + 4], 16, 64);
#pragma unroll
  for (int j = 0; j < 2; j++) c[j] = (b3 ? b[j + 2] : b[j]) + __shfl_xor(b3 ? b[j] : b[j + 2], 8, 64);
  T d = (b2 ? c[1] : c[0]) + __shfl_xor(b2 ? c[0] : c[1], 4, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  return d;  // sum of v[(lane >> 2) & 15]
}
template <typename T>
__device__ __forceinline__ T waveSum8(T (&v)[8], int lane) {
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
  T a[4], b[2];
#pragma unroll
  for (int j = 0; j < 4; j++) a[j] = (b5 ? v[j + 4] : v[j]) + __shfl_xor(b5 ? v[j] : v[j + 4], 32, 64);
#pragma unroll
  for (int j = 0; j < 2; j++) b[j] = (b4 ? a[j + 2] : a[j]) + __shfl_xor(b4 ? a[j] : a[j + 2], 16, 64);
  T d = (b3 ? b[1] : b[0]) + __shfl_xor(b3 ? b[0] : b[1], 8, 64);
  d += __shfl_xor(d, 4, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  return d;  // sum of v[(lane >> 3) & 7]
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
// K-S1: x_l <- D^-1 x_l for the small lumps of a range (thread per lump, every right-hand side: the
// diagonal block is fetched from memory once, not once per right-hand side)
template <typename T>
__global__ __launch_bounds__(256) void solveElimDiagL(SkelDev sk, SolveRef<T> ref,
                                                      int64_t lumpBegin, int64_t lumpEnd, int nRhs) {
  const int64_t l = lumpBegin + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (l >= lumpEnd) return;
  const int n = (int)(sk.lumpStart[l + 1] - sk.lumpStart[l]);
  if (n > kElimSmallMax) return;
  GP<const T> D = solveMat(ref) + sk.chainData[sk.chainColPtr[l]];
  if (n <= 4) {  // (the lower triangle stays in registers over the right-hand sides)
    T d[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) d[i][j] = i < n ? D[i * n + j] : (i == j ? T(1) : T(0));
    }
    const int64_t x0 = sk.lumpStart[l];
    for (int rhs = 0; rhs < nRhs; rhs++) {
      GP<T> xl = solveVecBase(ref) + (int64_t)rhs * ref.ldc + x0;
      T x[4];
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = i < n ? xl[i] : T(0);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        T s = x[i];
#pragma unroll
        for (int j = 0; j < i; j++) s -= d[i][j] * x[j];
        x[i] = s / d[i][i];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (i < n) xl[i] = x[i];
      }
    }
    return;
  }
  for (int rhs = 0; rhs < nRhs; rhs++) {
    GP<T> xl = solveVecBase(ref) + (int64_t)rhs * ref.ldc + sk.lumpStart[l];
    T x[kElimSmallMax];
    for (int i = 0; i < n; i++) x[i] = xl[i];
    for (int i = 0; i < n; i++) {
      T s = x[i];
      for (int j = 0; j < i; j++) s -= D[i * n + j] * x[j];
      x[i] = s / D[i * n + i];
    }
    for (int i = 0; i < n; i++) xl[i] = x[i];
  }
}

// K-S2: y[span] -= sum over the blocks B that sit in that row span, B * x_lump.  The blocks are
// listed per target span (SolveGatherEntry, sorted by span), an item is <= 256 entries of one
// span: thread per block, workgroup reduction, ONE atomic per row and item (instead of one per
// row and block, all on the few camera rows).
// Round 5: up to 16 right-hand sides per workgroup (blockIdx.y = group of 16).  The v_mfma's B operand
// has 16 columns and used one: column c now carries x_lump of right-hand side c, so the blocks of L
// are read once for all of them (ten right-hand sides re-read the 0.64 GB of point columns of BAL-871
// ten times: 2.40 ms of a 7.3 ms solve).
template <typename T>
__global__ __launch_bounds__(256) void solveElimGatherL(const SolveGatherItem* items,
                                                        const SolveGatherEntry* entries,
                                                        SolveRef<T> ref, int nRhs) {
  __shared__ T part[4][16][17];
  const SolveGatherItem it = items[blockIdx.x];
  GP<const T> data = solveMat(ref);
  const int rhs0 = 16 * blockIdx.y, nR = min(16, nRhs - rhs0);
  GP<T> vec0 = solveVecBase(ref) + (int64_t)rhs0 * ref.ldc;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (it.rows <= 16 && it.maxN <= 4) {
    // matrix-core path: per block ONE 8-byte load per lane -- lane (i, k) = (lane & 15, lane >> 4)
    // fetches B[i][k], the lanes together read the contiguous block -- and one v_mfma 16x16x4
    // whose B operand carries x_lump of right-hand side c in column c; the sum over the wave's
    // blocks stays in the accumulator (the thread-per-block loop below re-reads every cache line
    // n x rows times)
    using Acc = typename Mfma<T>::Acc;
    constexpr int U = 8;
    const int li = lane & 15, lk = lane >> 4;
    GP<const T> xcol = vec0 + (int64_t)min(li, nR - 1) * ref.ldc;
    Acc acc = {0, 0, 0, 0};
    const int e0 = it.entryBegin + 64 * wave;
    const int cnt = min(64, it.entryEnd - e0);
    SolveGatherEntry en = {0, 0, 0};
    if (lane < cnt) en = entries[e0 + lane];
    const int offLo = (int)(uint32_t)en.dataOff, offHi = (int)(en.dataOff >> 32);
    for (int t0 = 0; t0 < cnt; t0 += U) {
      T a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = min(t0 + u, cnt - 1);
        const int64_t off = ((int64_t)__builtin_amdgcn_readlane(offHi, t) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane(offLo, t);
        const int xo = __builtin_amdgcn_readlane(en.xOff, t);
        const int n = __builtin_amdgcn_readlane(en.n, t);
        // (round 3, tried: x_lump fetched by lanes (15, k) of the SAME wave load as B and shuffled to
        //  lanes (0, k) -- one load per block instead of two in a kernel whose texture addresser is 90 %
        //  busy -- measured 297 against 247 us: the two-region load costs more than it saves)
        const bool okA = li < it.rows && lk < n, okB = li < nR && lk < n;
        a[u] = okA ? data[off + li * n + lk] : T(0);
        b[u] = okB ? xcol[xo + lk] : T(0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (t0 + u < cnt) acc = Mfma<T>::run(a[u], b[u], acc);  // wave-uniform
      }
    }
    // column c of the product = right-hand side c: rows Mfma::row(lane, g)
#pragma unroll
    for (int g = 0; g < 4; g++) part[wave][li][Mfma<T>::row(lane, g)] = acc[g];
    __syncthreads();
    const int r = threadIdx.x & 15, c = threadIdx.x >> 4;
    if (r < it.rows && c < nR) {
      atomicSub(vec0 + (int64_t)c * ref.ldc + it.rowStart + r,
                part[0][c][r] + part[1][c][r] + part[2][c][r] + part[3][c][r]);
    }
    return;
  }
  const int e = it.entryBegin + (int)threadIdx.x;
  const bool live = e < it.entryEnd;
  for (int rhs = 0; rhs < nR; rhs++) {
    GP<T> vec = vec0 + (int64_t)rhs * ref.ldc;
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
      if (lane == 0) part[wave][0][0] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        atomicSub(vec + it.rowStart + r, part[0][0][0] + part[1][0][0] + part[2][0][0] + part[3][0][0]);
      }
      __syncthreads();
    }
  }
}

// sums over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15) by row rotations: vector-ALU
// instructions, no LDS crossbar (a __shfl is a ds_bpermute: ~14 clocks of a CU's crossbar each)
template <int N>
__device__ __forceinline__ double rowRorT(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ float rowRorT(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), in every lane of the row
template <typename T>
__device__ __forceinline__ T rowSum16(T v) {
  v += rowRorT<8>(v);
  v += rowRorT<4>(v);
  v += rowRorT<2>(v);
  v += rowRorT<1>(v);
  return v;
}

// K-S3: backward pass over a range of <= 4-wide lumps: x_l <- L_ll^-T (x_l - sum_blocks B^T y).
// 16 lanes per lump (four lumps per wave): lane (k, ip) = (sub & 3, sub >> 2) walks rows
// ip, ip+4, ... of every block and column k, so that the 16 lanes read 12 consecutive values per
// step; the blocks come from a lump-major descriptor list (no skeleton lookups: thread-per-lump
// K-S0 chases five dependent index loads per block and touches every cache line rows x n times).
template <typename T>
__global__ __launch_bounds__(256) void solveElimLumpsLt(const SolveLumpDesc* descs,
                                                        const SolveLumpBlock* blocks,
                                                        SolveRef<T> ref, int numLumps) {
  const int lane = threadIdx.x & 63, sub = lane & 15, k = sub & 3, ip = sub >> 2;
  const int idx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  if (idx >= numLumps) return;
  const SolveLumpDesc ld = descs[idx];
  const int n = ld.n;
  if (n > 4) return;  // (wider lumps of the range: panel kernels)
  GP<const T> data = solveMat(ref);
  GP<T> vec = solveVec(ref);
  const bool kOk = k < n;
  T acc = T(0);
  const int grp = lane & 48;
  // Round 3: PMC says this kernel is bound by the texture addresser (12.1 M wave loads per call on
  // BAL-871, TA busy 76 %, ~22 cycles per wave load whatever it fetches), so it issues fewer of
  // them.  The group's block descriptors come with ONE load (lane e takes block e, 16 per pass) and
  // travel by shuffle; a block of at most 32 values and 16 rows (the 9 x 3 blocks of bundle
  // adjustment) is two loads -- lane s takes elements s and s + 16 -- plus one for its y values
  // (lane s = row s), instead of four + four behind a dependent descriptor load.  Element e = (row e
  // / n, column e % n): its y comes from lane e / n of the group, its product goes to column e % n.
  const int nBlocks = ld.blockEnd - ld.blockBegin;
  T c0 = T(0), c1 = T(0);  // products of the lane's elements s and s + 16
  bool allFast = true;
  for (int b0 = 0; b0 < nBlocks; b0 += 16) {
    const bool hasB = b0 + sub < nBlocks;
    const SolveLumpBlock myB = blocks[ld.blockBegin + (hasB ? b0 + sub : 0)];
    const bool fastB = !hasB || (myB.rows <= 16 && myB.rows * n <= 32);
    // (a vote among the 16 lanes of the group: the groups of a wave run different trip counts, and
    //  the lanes of a group past the last lump have left the kernel)
    const unsigned long long slow = __ballot(!fastB);
    allFast = allFast && ((slow >> grp) & 0xffffull) == 0;
    if (!allFast) break;
    const int passB = min(16, nBlocks - b0);
    // (round 5, measured: y by two direct loads per block instead of one load + two lane gathers 299
    //  against 270 us; the descriptor by a load per block instead of four lane broadcasts too: 300)
    for (int e = 0; e < passB; e++) {
      const int64_t off = (int64_t)__shfl((int)(myB.dataOff >> 32), grp + e, 64) << 32 |
                          (uint32_t)__shfl((int)(uint32_t)myB.dataOff, grp + e, 64);
      const int yOff = __shfl(myB.yOff, grp + e, 64), rows = __shfl(myB.rows, grp + e, 64);
      const int ne = rows * n;
      const T v0 = sub < ne ? data[off + sub] : T(0);
      const T v1 = sub + 16 < ne ? data[off + sub + 16] : T(0);
      const T yv = sub < rows ? vec[yOff + sub] : T(0);
      c0 += v0 * __shfl(yv, grp + sub / n, 64);
      c1 += v1 * __shfl(yv, grp + min(15, (sub + 16) / n), 64);
    }
  }
  T colSum[4];  // column sums of the lump, in every lane of its group
  if (allFast) {
    // column j of the sum: the lanes whose element index is j modulo n.  Round 5: DPP row sums instead
    // of 32 broadcasts per lump -- the kernel was bound by ds_bpermute (64 of its 87 per wave sat here)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const T p = (sub % n == j ? c0 : T(0)) + ((sub + 16) % n == j ? c1 : T(0));
      colSum[j] = rowSum16(p);
    }
  } else {
    acc = T(0);
    for (int e = ld.blockBegin; e < ld.blockEnd; e++) {
      const SolveLumpBlock b = blocks[e];
      for (int i0 = 0; i0 < b.rows; i0 += 8) {
        const int ia = i0 + ip, ib = i0 + 4 + ip;
        const bool oa = kOk && ia < b.rows, ob = kOk && ib < b.rows;
        const T va = oa ? data[b.dataOff + ia * n + k] : T(0);
        const T vb = ob ? data[b.dataOff + ib * n + k] : T(0);
        const T ya = oa ? vec[b.yOff + ia] : T(0);
        const T yb = ob ? vec[b.yOff + ib] : T(0);
        acc += va * ya + vb * yb;
      }
    }
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);  // every lane: the sum for its column k
#pragma unroll
    for (int j = 0; j < 4; j++) colSum[j] = __shfl(acc, grp + j, 64);
  }
  // back substitution with the upper triangle L^T, redundantly in every lane of the group
  T x[4], d[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const T xj = j < n ? vec[ld.xOff + j] : T(0);
    x[j] = xj - colSum[j];
#pragma unroll
    for (int i = 0; i < 4; i++) d[i][j] = (i < n && j <= i) ? data[ld.diagOff + i * n + j] : (i == j ? T(1) : T(0));
  }
#pragma unroll
  for (int j = 3; j >= 0; j--) {
    T sres = x[j];
#pragma unroll
    for (int i = j + 1; i < 4; i++) sres -= d[i][j] * x[i];
    x[j] = sres / d[j][j];
  }
  if (ip == 0 && kOk) vec[ld.xOff + k] = k == 0 ? x[0] : k == 1 ? x[1] : k == 2 ? x[2] : x[3];
}

// K-S3m  the same pass for SEVERAL right-hand sides (round 5; blockIdx.y = group of RB).  K-S3 per
// right-hand side re-read the point columns of BAL-871 ten times for ten right-hand sides (2.76 ms of
// a 7.3 ms solve), and holding RB accumulator sets in ITS layout moved nothing (2.71 ms): with one
// element per lane every product needs its y through a lane gather, and the final column sums cost 64
// LDS-crossbar permutes per right-hand side -- the kernel is bound by ds_bpermute, not by loads.
// Here lane s of a lump's 16 lanes owns ROW s of every block: its n values come with n strided loads
// (the 16 lanes together read the contiguous block), its y with one load per right-hand side, no
// shuffle inside the loop; the column sums over the rows are DPP row rotations (vector ALU, no LDS),
// and blocks of more than 16 rows simply take several passes.
template <typename T, int RB>
__global__ __launch_bounds__(256) void solveElimLumpsLtMulti(const SolveLumpDesc* descs,
                                                             const SolveLumpBlock* blocks,
                                                             SolveRef<T> ref, int numLumps, int nRhs) {
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int idx0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool liveLump = idx0 < numLumps;
  const SolveLumpDesc ld = descs[liveLump ? idx0 : numLumps - 1];
  const int n = ld.n;
  if (n > 4) return;  // (wider lumps of the range: panel kernels; the caller checks the range's width)
  GP<const T> data = solveMat(ref);
  const int rhs0 = RB * blockIdx.y, nR = min(RB, nRhs - rhs0);
  GP<T> vecs[RB];
#pragma unroll
  for (int q = 0; q < RB; q++) vecs[q] = solveVecBase(ref) + (int64_t)(rhs0 + min(q, nR - 1)) * ref.ldc;
  T acc[RB][4];
#pragma unroll
  for (int q = 0; q < RB; q++) {
#pragma unroll
    for (int j = 0; j < 4; j++) acc[q][j] = T(0);
  }
  const int nBlocks = liveLump ? ld.blockEnd - ld.blockBegin : 0;
  // CB blocks per trip, every load of a trip issued before its first product: a wave's time is its
  // chain of dependent round trips (descriptor -> values), one block per trip took ten of them for
  // the typical widest lump of a wave; the next trip's descriptors are requested a trip ahead
  constexpr int CB = RB <= 4 ? 4 : 2;
  SolveLumpBlock nb[CB];
#pragma unroll
  for (int c = 0; c < CB; c++) nb[c] = blocks[ld.blockBegin + min(c, max(nBlocks - 1, 0))];
  for (int e0 = 0; e0 < nBlocks; e0 += CB) {
    SolveLumpBlock b[CB];
#pragma unroll
    for (int c = 0; c < CB; c++) b[c] = nb[c];
    if (e0 + CB < nBlocks) {
#pragma unroll
      for (int c = 0; c < CB; c++) nb[c] = blocks[ld.blockBegin + min(e0 + CB + c, nBlocks - 1)];
    }
    bool tall = false;
    T v[CB][4], y[CB][RB];
#pragma unroll
    for (int c = 0; c < CB; c++) {
      const bool on = e0 + c < nBlocks && sub < b[c].rows;
      tall = tall || (e0 + c < nBlocks && b[c].rows > 16);
      GP<const T> row = data + b[c].dataOff + (on ? sub : 0) * n;
#pragma unroll
      for (int j = 0; j < 4; j++) v[c][j] = (on && j < n) ? row[j] : T(0);
#pragma unroll
      for (int q = 0; q < RB; q++) {
        if (q < nR) y[c][q] = on ? vecs[q][b[c].yOff + sub] : T(0);  // (nR: workgroup-uniform)
      }
    }
#pragma unroll
    for (int c = 0; c < CB; c++) {
#pragma unroll
      for (int q = 0; q < RB; q++) {
        if (q < nR) {
#pragma unroll
          for (int j = 0; j < 4; j++) acc[q][j] += v[c][j] * y[c][q];
        }
      }
    }
    if (__any(tall)) {  // rows 16 .. of blocks taller than a lane group (not the 9 x 3 blocks of BAL)
#pragma unroll
      for (int c = 0; c < CB; c++) {
        const int rowsC = e0 + c < nBlocks ? b[c].rows : 0;
        for (int r = sub + 16; r < rowsC; r += 16) {
          GP<const T> row = data + b[c].dataOff + r * n;
          T vv[4];
#pragma unroll
          for (int j = 0; j < 4; j++) vv[j] = j < n ? row[j] : T(0);
#pragma unroll
          for (int q = 0; q < RB; q++) {
            if (q < nR) {
              const T yy = vecs[q][b[c].yOff + r];
#pragma unroll
              for (int j = 0; j < 4; j++) acc[q][j] += vv[j] * yy;
            }
          }
        }
      }
    }
  }
  // (every lane of the wave from here on: the DPP rotations read all 16 lanes of a row)
  T d[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) d[i][j] = (i < n && j <= i) ? data[ld.diagOff + i * n + j] : (i == j ? T(1) : T(0));
  }
#pragma unroll
  for (int q = 0; q < RB; q++) {
    T x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const T xj = j < n ? vecs[q][ld.xOff + j] : T(0);
      x[j] = xj - rowSum16(acc[q][j]);
    }
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      T sres = x[j];
#pragma unroll
      for (int i = j + 1; i < 4; i++) sres -= d[i][j] * x[i];
      x[j] = sres / d[j][j];
    }
    if (liveLump && sub < n && q < nR) {
      vecs[q][ld.xOff + sub] = sub == 0 ? x[0] : sub == 1 ? x[1] : sub == 2 ? x[2] : x[3];
    }
  }
}

// (round 3, tried and removed: K-S3 with the wave's four columns staged through LDS like the factor's
//  K1s.  PMC says K-S3 is texture-addresser bound -- 12.1 M wave loads per call, TA 76 % busy, 561 us
//  on BAL-871 -- but three staged variants were all SLOWER: blocks walked by the (k, ip) lanes with y
//  from global memory 859 us; a lane per block with its y values prefetched 691 us (64 scattered
//  8-byte gathers per instruction: L1 stalled on pending misses 65 % of the time); lane = row of the
//  block, y read as one contiguous piece per block, 775 us -- 32 KB of LDS per workgroup leaves 20
//  waves per CU for a loop whose every iteration is a dependent shuffle + load + LDS read.)
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
  // Row i (BACKWARD: column i) is divided by its own diagonal entry up front -- lane-local, off the
  // dependent chain -- so that a step of the chain is one broadcast and one fma: lane j's value IS
  // x_j when step j comes.
  const T inv = T(1) / Ls[lane * LD + lane];
  T xi = (lane < nb ? x[lane] : T(0)) * inv;
  if (!BACKWARD) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const T lij = Ls[lane * LD + j] * inv;  // column j of L, lane = row
      const T xj = laneBcast(xi, j);
      xi = lane > j ? xi - lij * xj : xi;
    }
  } else {
#pragma unroll
    for (int j = NB - 1; j >= 0; j--) {
      const T lji = Ls[j * LD + lane] * inv;  // row j of L, lane = column
      const T xj = laneBcast(xi, j);
      xi = lane < j ? xi - lji * xj : xi;
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

// ---- wide lumps: one outer block (up to 256 columns, <= 4 panels) per step ---------------------
// The chain of a wide lump is one panel per level: four launches of four kernels per 256 columns.
// These kernels take a whole outer block: K-B1 solves the block's own triangle in one workgroup
// (x_B stays in LDS across its panels), K-B2 applies the block to the rows below it in one launch.
// `first` = descriptor of the block's first panel, w = columns of the block.
template <typename T>
__device__ __forceinline__ void stageTri64(GP<const T> A, int lda, int nb, T* Ls) {
  constexpr int NB = kPanelWidth, LD = NB + 1;
  const int tid = threadIdx.x;
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

// wave 0: xs[0..nb) <- L^-1 xs (BACKWARD: L^-T), L staged by stageTri64
template <typename T, bool BACKWARD>
__device__ __forceinline__ void triSolve64(const T* Ls, T* xs, int nb) {
  constexpr int NB = kPanelWidth, LD = NB + 1;
  const int lane = threadIdx.x;
  if (lane >= 64) return;
  // (rows / columns pre-divided by their diagonal entry: one broadcast + one fma per step, see
  //  solveTriPanel)
  const T inv = T(1) / Ls[lane * LD + lane];
  T xi = (lane < nb ? xs[lane] : T(0)) * inv;
  if (!BACKWARD) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const T lij = Ls[lane * LD + j] * inv;
      const T xj = laneBcast(xi, j);
      xi = lane > j ? xi - lij * xj : xi;
    }
  } else {
#pragma unroll
    for (int j = NB - 1; j >= 0; j--) {
      const T lji = Ls[j * LD + lane] * inv;
      const T xj = laneBcast(xi, j);
      xi = lane < j ? xi - lji * xj : xi;
    }
  }
  if (lane < nb) xs[lane] = xi;
}

// K-B1: the block's own triangle, one workgroup (256 threads)
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) void solveTriBlock(PanelDesc first, int w, SolveRef<T> ref) {
  constexpr int NB = kPanelWidth, LD = NB + 1;
  __shared__ T Ls[NB * LD];
  __shared__ T xs[kSolveBlock];
  __shared__ T part[4][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lda = first.lda;
  GP<const T> A = solveMat(ref) + first.diagOff;  // (c0, c0) of the block
  GP<T> x = solveVec(ref) + first.vecOff;
  if (tid < w) xs[tid] = x[tid];
  const int nq = (w + NB - 1) / NB;
  if (!BACKWARD) {
    for (int q = 0; q < nq; q++) {
      const int c = q * NB, nb = min(NB, w - c);
      stageTri64<T>(A + (int64_t)c * lda + c, lda, nb, Ls);
      __syncthreads();
      triSolve64<T, false>(Ls, xs + c, nb);
      __syncthreads();
      // rows of the block below the panel: xs[r] -= L[r, c..c+nb) . xs[c..c+nb)
      // (wave w takes rows c+nb+w, +4, ...; 16 rows in flight per lane and round trip)
      const T xk = lane < nb ? xs[c + lane] : T(0);
      for (int r0 = c + nb + wave; r0 < w; r0 += 64) {
        T p[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int r = min(r0 + 4 * u, w - 1);
          p[u] = lane < nb ? A[(int64_t)r * lda + c + lane] : T(0);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) p[u] *= xk;
        const T sres = waveSum16(p, lane);
        const int ur = (lane >> 2) & 15;
        if ((lane & 3) == 0 && r0 + 4 * ur < w) xs[r0 + 4 * ur] -= sres;
      }
      __syncthreads();
    }
  } else {
    for (int q = nq - 1; q >= 0; q--) {
      const int c = q * NB, nb = min(NB, w - c);
      stageTri64<T>(A + (int64_t)c * lda + c, lda, nb, Ls);
      // xs[c + k] -= sum over the block rows r below the panel of L[r, c + k] * xs[r]
      T acc = T(0);
      for (int r0 = c + nb + wave; r0 < w; r0 += 64) {
        T p[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int r = min(r0 + 4 * u, w - 1);
          p[u] = lane < nb ? A[(int64_t)r * lda + c + lane] : T(0);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) acc += (r0 + 4 * u < w) ? p[u] * xs[min(r0 + 4 * u, w - 1)] : T(0);
      }
      part[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && lane < nb) {
        xs[c + lane] -= part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
      }
      __syncthreads();
      triSolve64<T, true>(Ls, xs + c, nb);
      __syncthreads();
    }
  }
  if (tid < w) x[tid] = xs[tid];
}

// ---- round 3: the block's triangle through INVERTED 64 x 64 diagonal blocks ------------------------
// K-B1 spends ~20 us per 256 columns: per panel a global round trip for the triangle, a 64-step
// dependent chain on ONE wave (triSolve64, ~1.8 us), and another round trip for the rows of the block
// below the panel -- eight dependent round trips and four serial chains per block, 62 blocks per
// solve of BAL-871 (1.25 of the 2.95 ms).  Prefetching around the unrolled chain was tried in round
// 2 and lost to register pressure.  Here the chains leave the block kernel altogether:
//   K-B0 (one launch per solve call, one wave per panel of every block group): the inverse of the
//        panel's nb x nb diagonal block of L, lane t = column t by forward substitution in
//        registers, stored row-major (64 x 64 slot, identity-padded);
//   K-B1i: left-looking over the block's panels with everything -- the block's strictly lower 64 x 64
//        tiles and the four inverses, 320 KB -- requested up front (one round trip):
//        x_q = Inv_q (y_q - sum_{p<q} L_qp x_p); both products are row sums over 64 lanes
//        (waveSum16), two barriers per panel.  Backward: x_q = Inv_q^T (y_q - sum_{p>q} L_pq^T x_p),
//        column sums per lane + a cross-wave reduction.
// The solve through explicitly inverted diagonal blocks is the standard GPU formulation (as the
// factor's trsm already does with 16 x 16 blocks): its error grows with the condition number of a
// 64 x 64 diagonal block of L, not with that of the matrix.
template <typename T>
__global__ __launch_bounds__(64) void solveInvertPanels(const PanelDesc* list, T* invOut,
                                                        int64_t batchStride, SolveRef<T> ref,
                                                        unsigned long long* arm = nullptr,
                                                        int64_t armWords = 0) {
  constexpr int NB = kPanelWidth, LD = NB + 1;
  __shared__ T Ls[NB * LD];
  const PanelDesc pd = list[blockIdx.x];
  const int t = threadIdx.x, nb = pd.nb, lda = pd.lda;
  // (round 6: this launch also ARMS the exchange buffer of the persistent sweeps that follow it --
  //  every word all-ones = "not published yet", hip_sweep_kernels.h)
  for (int64_t i = ((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * 64 + t; i < armWords;
       i += (int64_t)gridDim.x * gridDim.z * 64) {
    arm[i] = ~0ull;
  }
  GP<const T> A = solveMat(ref) + pd.diagOff;
  T y[NB];
  {  // row i: lanes = columns; all 64 loads in flight before the first LDS store
#pragma unroll
    for (int i = 0; i < NB; i++) {
      y[i] = A[(int64_t)min(i, nb - 1) * lda + min(t, min(i, nb - 1))];
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      Ls[i * LD + t] = (i < nb && t <= i) ? y[i] : (i == t ? T(1) : T(0));
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // (1 / L_jj once per lane, handed round with readlane: no division inside the chain)
  const T myInvD = T(1) / Ls[t * LD + t];
#pragma unroll
  for (int i = 0; i < NB; i++) y[i] = i == t ? T(1) : T(0);
#pragma unroll
  for (int j = 0; j < NB; j++) {
    y[j] *= readLaneT(myInvD, j);
#pragma unroll
    for (int i = j + 1; i < NB; i++) y[i] -= Ls[i * LD + j] * y[j];
  }
  GP<T> out = (GP<T>)invOut + (int64_t)blockIdx.z * batchStride + (int64_t)blockIdx.x * NB * NB;
#pragma unroll
  for (int i = 0; i < NB; i++) out[i * NB + t] = y[i];  // Inv[i][t]
}

template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void solveTriBlockInv(
    PanelDesc first, int w, const T* invBase, int64_t batchStride, SolveRef<T> ref) {
  constexpr int NB = kPanelWidth, NQ = kSolveBlock / NB;
  __shared__ T xs[kSolveBlock];
  __shared__ T ts[NB];
  __shared__ T part[4][NB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lda = first.lda;
  GP<const T> A = solveMat(ref) + first.diagOff;  // (c0, c0) of the block
  GP<const T> inv = (GP<const T>)invBase + (int64_t)blockIdx.z * batchStride;
  GP<T> x = solveVec(ref) + first.vecOff;
  const int nq = (w + NB - 1) / NB;
  // everything the block needs, requested before the first use: tile (q, p) of the block's strict
  // lower part, rows 16 wv .. 16 wv + 15 of it (lane = column), and the same rows of Inv_q
  T Lt[NQ * (NQ - 1) / 2][16], Iv[NQ][16];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      Iv[q][u] = q < nq ? inv[(int64_t)q * NB * NB + (16 * wv + u) * NB + lane] : T(0);
    }
#pragma unroll
    for (int p = 0; p < q; p++) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int r = NB * q + 16 * wv + u;
        Lt[q * (q - 1) / 2 + p][u] = r < w ? A[(int64_t)r * lda + NB * p + lane] : T(0);
      }
    }
  }
  if (tid < kSolveBlock) xs[tid] = tid < w ? x[tid] : T(0);
  __syncthreads();
  const int ur = (lane >> 2) & 15;
  if (!BACKWARD) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      if (q < nq) {
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = T(0);
#pragma unroll
        for (int p = 0; p < q; p++) {
          const T xp = xs[NB * p + lane];
#pragma unroll
          for (int u = 0; u < 16; u++) v[u] += Lt[q * (q - 1) / 2 + p][u] * xp;
        }
        const T s = waveSum16(v, lane);
        if ((lane & 3) == 0) ts[16 * wv + ur] = xs[NB * q + 16 * wv + ur] - s;
        __syncthreads();
        const T tq = ts[lane];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = Iv[q][u] * tq;
        const T xq = waveSum16(v, lane);
        if ((lane & 3) == 0) xs[NB * q + 16 * wv + ur] = xq;
        __syncthreads();
      }
    }
  } else {
#pragma unroll
    for (int q = NQ - 1; q >= 0; q--) {
      if (q < nq) {
        // t_q[c] = y_q[c] - sum over the tiles (p, q), p > q, of L[row][c] x[row]; lane = c
        T acc = T(0);
#pragma unroll
        for (int p = q + 1; p < NQ; p++) {
#pragma unroll
          for (int u = 0; u < 16; u++) acc += Lt[p * (p - 1) / 2 + q][u] * xs[NB * p + 16 * wv + u];
        }
        part[wv][lane] = acc;
        __syncthreads();
        if (wv == 0) ts[lane] = xs[NB * q + lane] - (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
        __syncthreads();
        // x_q[c] = sum_r Inv_q[r][c] t_q[r]
        acc = T(0);
#pragma unroll
        for (int u = 0; u < 16; u++) acc += Iv[q][u] * ts[16 * wv + u];
        part[wv][lane] = acc;
        __syncthreads();
        if (wv == 0) xs[NB * q + lane] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        __syncthreads();
      }
    }
  }
  if (tid < w) x[tid] = xs[tid];
}

// K-B2 forward: rows below the block, 64 rows per workgroup: x[target(q)] -= L[row q, block] . x_B
// (`last` = descriptor of the block's last panel: its below-rows are the block's below-rows)
template <typename T>
__global__ __launch_bounds__(256) void solveGemvBlockL(PanelDesc first, PanelDesc last, int w,
                                                       const int32_t* rowGlobal, SolveRef<T> ref) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lda = first.lda;
  GP<const T> A = solveMat(ref) + first.diagOff + (int64_t)w * lda;  // first row below, col c0
  GP<T> vec = solveVec(ref);
  T xk[4];
#pragma unroll
  for (int i = 0; i < 4; i++) xk[i] = lane + 64 * i < w ? vec[first.vecOff + lane + 64 * i] : T(0);
  const int rowTile = blockIdx.x * kTile;
  const int rows = min(kTile, last.rowsBelow - rowTile);
  for (int r0 = wave * 16; r0 < wave * 16 + 16; r0 += 8) {
    T p[8][4];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      GP<const T> row = A + (int64_t)(rowTile + min(r0 + u, rows - 1)) * lda;
#pragma unroll
      for (int i = 0; i < 4; i++) p[u][i] = lane + 64 * i < w ? row[lane + 64 * i] : T(0);
    }
    T d[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      d[u] = p[u][0] * xk[0] + p[u][1] * xk[1] + p[u][2] * xk[2] + p[u][3] * xk[3];
    }
    const T sres = waveSum8(d, lane);
    const int ur = (lane >> 3) & 7;
    if ((lane & 7) == 0 && r0 + ur < rows) {
      atomicSub(vec + solveTargetRow(last, rowGlobal, rowTile + r0 + ur), sres);
    }
  }
}

// K-B2 backward: x_B[k] -= sum over a 64-row tile of L[row q, c0 + k] * x[target(q)]; thread = k
template <typename T>
__global__ __launch_bounds__(256) void solveGemvBlockLt(PanelDesc first, PanelDesc last, int w,
                                                        const int32_t* rowGlobal, SolveRef<T> ref) {
  __shared__ T xq[kTile];
  const int tid = threadIdx.x, lda = first.lda;
  GP<const T> A = solveMat(ref) + first.diagOff + (int64_t)w * lda;
  GP<T> vec = solveVec(ref);
  const int rowTile = blockIdx.x * kTile;
  const int rows = min(kTile, last.rowsBelow - rowTile);
  if (tid < kTile) xq[tid] = tid < rows ? vec[solveTargetRow(last, rowGlobal, rowTile + tid)] : T(0);
  __syncthreads();
  T acc = T(0);
  const int k = min(tid, w - 1);
  for (int r0 = 0; r0 < rows; r0 += 16) {
    T p[16];
#pragma unroll
    for (int u = 0; u < 16; u++) p[u] = A[(int64_t)(rowTile + min(r0 + u, rows - 1)) * lda + k];
#pragma unroll
    for (int u = 0; u < 16; u++) acc += (r0 + u < rows) ? p[u] * xq[min(r0 + u, kTile - 1)] : T(0);
  }
  if (tid < w) atomicSub(vec + first.vecOff + tid, acc);
}

// ---- Solver::addMvFrom (Solver.cpp:400-449): out += alpha * A * in on the trailing block from a
// lump on, A symmetric with its lower blocks in skeleton layout.  One workgroup per tile of 64 rows
// x kMvCols columns of a lump's column (a wave takes 16 rows, a lane 4 columns): a row adds
// A[r, cols] . in[cols] to out[r], a column adds A[rows, c] . in[rows] to out[c] (the transposed
// half of the symmetric product; strictly-lower part inside the diagonal block) -- both summed in
// registers over the tile, so that a tile of 16 K values ends in ~1 K atomics instead of one per
// value (the first version: 5.6 ms for the 15 507-wide camera block of BAL-1723, i.e. 120 M atomics).
// Replaces the symm / gemv / assembleVec / assembleVecT / gemvT sequence of the reference.
constexpr int kMvCols = 256;
template <typename T>
__global__ __launch_bounds__(256) void addMvKernel(SkelDev sk, const int64_t* tiles, const T* mat,
                                                   const T* in, int64_t inStride, T* out,
                                                   int64_t outStride, T alpha) {
  // tiles: triples (lump, first row of the tile within the lump column, first column)
  const int64_t lump = tiles[3 * blockIdx.x], r0 = tiles[3 * blockIdx.x + 1];
  const int64_t c0 = tiles[3 * blockIdx.x + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ls = sk.lumpStart[lump], n = sk.lumpStart[lump + 1] - ls;
  const int64_t ch0 = sk.chainColPtr[lump], nCh = sk.chainColPtr[lump + 1] - ch0;
  const int64_t totalRows = sk.chainRowsTillEnd[ch0 + nCh - 1];
  const T* A = mat + sk.chainData[ch0];
  const T* x = in + (int64_t)blockIdx.y * inStride;
  T* y = out + (int64_t)blockIdx.y * outStride;
  // global row index of row r of the column: inside the diagonal block, or through the chains
  auto globalRow = [&](int64_t r) -> int64_t {
    if (r < n) return ls + r;
    int64_t lo = 0, hi = nCh;  // chainRowsTillEnd[c] > r  -> chain holding row r
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (sk.chainRowsTillEnd[ch0 + mid - 1] <= r) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    return sk.spanStart[sk.chainRowSpan[ch0 + lo]] + (r - sk.chainRowsTillEnd[ch0 + lo - 1]);
  };
  constexpr int NQ = kMvCols / 64;
  T xc[NQ], colAcc[NQ];
  bool okc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int64_t c = c0 + lane + 64 * q;
    okc[q] = c < n;
    xc[q] = okc[q] ? x[ls + c] : T(0);
    colAcc[q] = T(0);
  }
  T rowDot[16];
  const int64_t rw = r0 + 16 * wave;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int64_t r = rw + i;
    T d = T(0);
    if (r < totalRows) {  // wave-uniform
      const bool inDiag = r < n;
      const T xr = x[globalRow(r)];
      const T* row = A + r * n + c0 + lane;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int64_t c = c0 + lane + 64 * q;
        // (upper part of the diagonal block: not stored)
        if (okc[q] && !(inDiag && c > r)) {
          const T a = row[64 * q];
          d += a * xc[q];
          if (!(inDiag && c == r)) colAcc[q] += a * xr;
        }
      }
    }
    rowDot[i] = d;
  }
  // row sums: the butterfly leaves the sum of row u in the lanes with ((lane >> 2) & 15) == u
  const T rs = waveSum16(rowDot, lane);
  if ((lane & 3) == 0) {
    const int64_t r = rw + ((lane >> 2) & 15);
    if (r < totalRows) unsafeAtomicAdd(y + globalRow(r), alpha * rs);
  }
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    if (okc[q] && colAcc[q] != T(0)) unsafeAtomicAdd(y + ls + c0 + lane + 64 * q, alpha * colAcc[q]);
  }
}


// ---- per-op SolveCtx boundary (MatOps.h:139-168) -------------------------------------------------
// Kernels behind the reference's op-by-op solve driver (Solver.cpp:303-328,357-381,419-448):
// symm / gemv / assembleVec / gemvT / assembleVecT on a dense temporary of nRows x nRHS values,
// row-major, one slice per batch entry (blockIdx.z) -- the layout of CpuBaseSolveCtx::tmpBuf
// (MatOpsCpuBase.h:381-428) and of the device buffer of MatOpsCuda.cu:1093-1181.  These take the
// place of the cublas gemm / symm calls there; solve() itself never goes through them (fused
// path), so they are written for clarity: one wave per row, lanes over the columns.
template <typename T>
__device__ __forceinline__ GP<T> solveTmp(T* tmp, int64_t tmpStride) {
  return (GP<T>)tmp + (int64_t)blockIdx.z * tmpStride;
}

// tmp[r][rhs] = alpha * M[r][:] . A[offA + :, rhs]      (SolveCtx::gemv)
template <typename T>
__global__ __launch_bounds__(256) void perOpGemv(SolveRef<T> mv, int64_t offM, int64_t nRows,
                                                 int64_t nCols, SolveRef<T> av, int64_t offA, T alpha,
                                                 T* tmp, int64_t tmpStride, int nRHS) {
  // mv: matrices (data); av: the vectors A (.vec / .vecs, .ldc = lda)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= nRows) return;
  GP<const T> M = solveMat(mv) + offM + r * nCols;
  GP<const T> x = (GP<const T>)solveVec(av) + offA;
  T dot = T(0);
  for (int64_t c = lane; c < nCols; c += 64) dot += M[c] * x[c];
  dot = waveSum(dot);
  if (lane == 0) solveTmp(tmp, tmpStride)[r * nRHS + blockIdx.y] = alpha * dot;
}

// A[offA + c, rhs] += alpha * sum_r M[r][c] * tmp[r][rhs]      (SolveCtx::gemvT)
// workgroup = 64 columns x a chunk of 256 rows (wave w takes rows w, w+4, ...), one atomic per
// column and workgroup
constexpr int kPerOpRowChunk = 256;
template <typename T>
__global__ __launch_bounds__(256) void perOpGemvT(SolveRef<T> mv, int64_t offM, int64_t nRows,
                                                  int64_t nCols, SolveRef<T> av, int64_t offA, T alpha,
                                                  const T* tmp, int64_t tmpStride, int nRHS) {
  __shared__ T part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t colTiles = (nCols + 63) / 64;
  const int64_t c = (int64_t)(blockIdx.x % colTiles) * 64 + lane;
  const int64_t rBegin = (int64_t)(blockIdx.x / colTiles) * kPerOpRowChunk;
  const int64_t rEnd = min(rBegin + (int64_t)kPerOpRowChunk, nRows);
  GP<const T> M = solveMat(mv) + offM;
  GP<const T> t = solveTmp(const_cast<T*>(tmp), tmpStride);
  T acc = T(0);
  if (c < nCols) {
    for (int64_t r = rBegin + wave; r < rEnd; r += 4) acc += M[r * nCols + c] * t[r * nRHS + blockIdx.y];
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < nCols) {
    unsafeAtomicAdd((T*)(solveVec(av) + offA + c),
                    alpha * (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]));
  }
}

// D[offC + i, rhs] += alpha * sum_j S(i,j) C[offC + j, rhs],  S = symmetric matrix whose lower
// triangle is the row-major n x n block at offM      (SolveCtx::symm)
template <typename T>
__global__ __launch_bounds__(256) void perOpSymm(SolveRef<T> mv, int64_t offM, int64_t n,
                                                 SolveRef<T> cv, int64_t offC, SolveRef<T> dv, T alpha) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  GP<const T> A = solveMat(mv) + offM;
  GP<const T> x = (GP<const T>)solveVec(cv) + offC;
  T dot = T(0);
  for (int64_t j = lane; j < n; j += 64) dot += (j <= i ? A[i * n + j] : A[j * n + i]) * x[j];
  dot = waveSum(dot);
  if (lane == 0) solveVec(dv)[offC + i] += alpha * dot;
}

// C[spanStart(chain i) + j, rhs] += tmp[rowOffset(i) + j][rhs] over the chains
// [chainColPtr, chainColPtr + numColItems) of one lump column      (SolveCtx::assembleVec);
// GATHER: the transposed copy, tmp <- C      (SolveCtx::assembleVecT).  One workgroup per chain.
template <typename T, bool GATHER>
__global__ __launch_bounds__(256) void perOpAssembleVec(SkelDev sk, int64_t chainColPtr,
                                                        SolveRef<T> cv, T* tmp, int64_t tmpStride,
                                                        int nRHS) {
  const int64_t ch = chainColPtr + blockIdx.x;
  const int64_t startRow = sk.chainRowsTillEnd[chainColPtr - 1];
  const int64_t rowOffset = sk.chainRowsTillEnd[ch - 1] - startRow;
  const int64_t span = sk.chainRowSpan[ch];
  const int64_t s0 = sk.spanStart[span], sz = sk.spanStart[span + 1] - s0;
  GP<T> C = solveVec(cv) + s0;
  GP<T> t = solveTmp(tmp, tmpStride) + rowOffset * nRHS + blockIdx.y;
  for (int64_t j = threadIdx.x; j < sz; j += 256) {
    if (GATHER) {
      t[j * nRHS] = C[j];
    } else {
      C[j] += t[j * nRHS];
    }
  }
}

}  // namespace hipk
}  // namespace BaSpaCho
