// Persistent solve sweeps for SEVERAL right-hand sides: the matrix cores do the products (round 6).
//
// hip_sweep_kernels.h walks a wide run with one right-hand side per set of workgroups; ten right-
// hand sides would be ten sets (1 640 workgroups for BAL-871: not resident, so the multi-launch path
// took over: 1.64 of solve-10's 4.1 ms).  Here ONE set of workgroups carries up to 16 right-hand
// sides: every product is  D(16 rows x 16 rhs) += A(16 x 4) B(4 x 16)  on v_mfma_f64_16x16x4 -- the
// tile of L is the A operand (one value per lane and 4 columns, held in registers for the whole
// launch by the spines), the right-hand sides are the B operand, and the 16 columns of the
// instruction that a single right-hand side wastes are the other fifteen.  No cross-lane sums at all.
// Same roles, tickets, self-validating exchange words, poller waves, staggered start and watchdog as
// the one-RHS sweep; what changes is the layout of everything that is exchanged: x and the far sums
// live as [row][16] (16 right-hand sides interleaved), which IS the B-operand order of the
// instruction (value of lane l for rows 4 kb .. 4 kb + 3 at 64 kb + l) and makes every publish and
// every poll a run of contiguous 512-byte wave accesses.
// Forward far tiles are 48 rows, backward far tiles 48 columns (three compute waves of 16 + the
// poller): both stream A operands of 16 x 192 per source block and differ only in how a tile of L is
// addressed -- backward needs L^T, which in A-operand order is the NATURAL, coalesced read.
#pragma once

#include <hip/hip_runtime.h>

#include "hip_sweep_kernels.h"

namespace BaSpaCho {
namespace hipk {

constexpr int kSweepR = 16;     // right-hand sides per instance
constexpr int kSweepMRing = 4;  // far roles: source blocks of x kept in LDS

// A-operand element of a 64 x 64 tile for (wave w, k-block kb), lane l: tile row 16 w + (l & 15),
// tile column 4 kb + (l >> 4).  TRANSPOSED: the tile's rows are matrix columns.
template <typename T, bool TRANSPOSED>
__device__ __forceinline__ T sweepMLoadA(GP<const T> A, int lda, int w_, int r0, int c0, int wv, int kb,
                                         int lane) {
  // r0 / c0: first matrix row / column of the operand's (tile-row, tile-column) origin
  const int i = 16 * wv + (lane & 15), k = 4 * kb + (lane >> 4);
  const int row = TRANSPOSED ? r0 + k : r0 + i, col = TRANSPOSED ? c0 + i : c0 + k;
  // (no mask and no post-processing, so that all of a spine's loads are in flight together: a clamped
  //  element only ever meets x = 0 or lands in a row nobody stores.  The products must SUBTRACT: it is
  //  x that travels negated -- xs, the exchange slots and the far sums all hold -x / -sum)
  return A[(int64_t)min(row, w_ - 1) * lda + min(col, w_ - 1)];
}

// ---- spine ---------------------------------------------------------------------------------------
template <typename T, bool BACKWARD>
__device__ __forceinline__ void sweepMSpine(const SweepDesc& sd, int b, int nr, GP<const T> A, GP<T> vec0,
                                            int64_t ldc, GP<const T> inv, GP<T> xq, GP<const T> farq,
                                            SweepWatch& watch, T* lds, int fault, long long* trace) {
  constexpr int NB = kPanelWidth, Q = kSweepQ, W = kSweepW, R = kSweepR;
  using Acc = typename Mfma<T>::Acc;
  if (trace && threadIdx.x == 0) trace[8 * b] = (long long)wall_clock64();
  T* IvA = lds;                  // Q x [wave][kb][lane]: inverses in A-operand order
  T* xb = IvA + Q * NB * NB;     // x of the previous block, [W][R]
  T* xs = xb + W * R;            // x of this block, [W][R]
  T* tb = xs + W * R;            // right-hand side of the panel being solved, [NB][R]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lda = sd.lda, w = sd.w;
  const int cb = b * W;
  const int nq = min(Q, (w - cb + NB - 1) / NB);
  const int bp = BACKWARD ? b + 1 : b - 1;
  const bool hasPrev = BACKWARD ? bp < sd.nBlocks : bp >= 0;
  const int cp = bp * W;
  const bool hasFar = BACKWARD ? (sd.rowsBelow > 0 || b + 2 < sd.nBlocks) : b >= 2;
  const int rhs = lane & 15, cpS = hasPrev ? cp : 0;

  // operands on chip
  T Lp[Q][Q][16], Li[Q * (Q - 1) / 2][16];
#pragma unroll
  for (int q = 0; q < Q; q++) {
#pragma unroll
    for (int pp = 0; pp < Q; pp++) {
#pragma unroll
      for (int kb = 0; kb < 16; kb++) {
        // (a first block has no previous one: it loads block 0's tiles and never uses them)
        Lp[q][pp][kb] = BACKWARD ? sweepMLoadA<T, true>(A, lda, w, cpS + NB * pp, cb + NB * q, wv, kb, lane)
                                 : sweepMLoadA<T, false>(A, lda, w, cb + NB * q, cpS + NB * pp, wv, kb, lane);
      }
    }
  }
#pragma unroll
  for (int hi = 1; hi < Q; hi++) {
#pragma unroll
    for (int lo = 0; lo < hi; lo++) {
#pragma unroll
      for (int kb = 0; kb < 16; kb++) {
        // forward: target panel hi, source panel lo (rows of hi, columns of lo); backward: target lo, source hi
        Li[hi * (hi - 1) / 2 + lo][kb] =
            BACKWARD ? sweepMLoadA<T, true>(A, lda, w, cb + NB * hi, cb + NB * lo, wv, kb, lane)
                     : sweepMLoadA<T, false>(A, lda, w, cb + NB * hi, cb + NB * lo, wv, kb, lane);
      }
    }
  }
  {
    GP<const T> src = inv + (int64_t)(sd.invSlot + b * Q) * NB * NB;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      T v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = q < nq ? src[(int64_t)q * NB * NB + tid + 256 * i] : T(0);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int e = tid + 256 * i, r = e >> 6, c = e & 63;
        const int ti = BACKWARD ? c : r, tk = BACKWARD ? r : c;  // A[ti][tk]
        IvA[q * NB * NB + ((ti >> 4) * 16 + (tk >> 2)) * 64 + (ti & 15) + 16 * (tk & 3)] = v[i];
      }
    }
  }
  // right-hand sides of the block's rows, accumulator layout
  Acc D[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int r = cb + NB * q + 16 * wv + Mfma<T>::row(lane, reg);
      const T y = vec0[(int64_t)min(rhs, nr - 1) * ldc + sd.vecOff + min(r, w - 1)];
      D[q][reg] = (r < w && rhs < nr) ? y : T(0);
    }
  }
  for (int i = tid; i < W * R; i += 256) xs[i] = T(0);  // (panels not solved yet / not there read as zero)
  __syncthreads();
  if (trace && tid == 0) trace[8 * b + 1] = (long long)wall_clock64();  // operands on chip

  // wait: the far sums of this block's rows (they had a step of slack), then x of the previous block
  // (this wave's quarter of it, shared through LDS)
  if (hasFar) {
    GP<const T> pp[12];
    bool need[12];
    T got[12];
#pragma unroll
    for (int q = 0; q < Q; q++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int e = 4 * q + reg, r = cb + NB * q + 16 * wv + Mfma<T>::row(lane, reg);
        need[e] = r < w;
        pp[e] = farq + (need[e] ? (int64_t)r * R + rhs : 0);
        got[e] = T(0);
      }
    }
    if (!sweepWait<T, 12>(pp, need, got, watch)) return;
#pragma unroll
    for (int q = 0; q < Q; q++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) D[q][reg] += got[4 * q + reg];  // (the far roles publish -sum)
    }
  }
  if (trace && tid == 0) trace[8 * b + 4] = (long long)wall_clock64();  // far sums in
  {
    GP<const T> pp[12];
    bool need[12];
    T got[12];
#pragma unroll
    for (int e = 0; e < 12; e++) {
      const int kb = 12 * wv + e, row = cp + 4 * kb + (lane >> 4);
      need[e] = hasPrev && row < w;
      pp[e] = (GP<const T>)xq + (need[e] ? (int64_t)row * R + rhs : 0);
      got[e] = T(0);
    }
    if (!sweepWait<T, 12>(pp, need, got, watch)) return;
#pragma unroll
    for (int e = 0; e < 12; e++) xb[(12 * wv + e) * 64 + lane] = got[e];
  }
  __syncthreads();
  if (trace && tid == 0) trace[8 * b + 2] = (long long)wall_clock64();  // x of the previous block in
  if (hasPrev) {
#pragma unroll
    for (int pp = 0; pp < Q; pp++) {
#pragma unroll
      for (int kb = 0; kb < 16; kb++) {
        const T xv = xb[(16 * pp + kb) * 64 + lane];
#pragma unroll
        for (int q = 0; q < Q; q++) D[q] = Mfma<T>::run(Lp[q][pp][kb], xv, D[q]);
      }
    }
  }
#pragma unroll
  for (int qi = 0; qi < Q; qi++) {
    const int q = BACKWARD ? Q - 1 - qi : qi;
    if (q < nq) {
#pragma unroll
      for (int p = 0; p < Q; p++) {
        if (qi > 0 && (BACKWARD ? p > q : p < q)) {
          const int hi = p > q ? p : q, lo = p > q ? q : p;
#pragma unroll
          for (int kb = 0; kb < 16; kb++) {
            D[q] = Mfma<T>::run(Li[hi * (hi - 1) / 2 + lo][kb], xs[(16 * p + kb) * 64 + lane], D[q]);
          }
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++) tb[(16 * wv + Mfma<T>::row(lane, reg)) * R + rhs] = D[q][reg];
      __syncthreads();
      Acc X = {0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < 16; kb++) {
        X = Mfma<T>::run(IvA[q * NB * NB + (wv * 16 + kb) * 64 + lane], tb[kb * 64 + lane], X);
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int rr = 16 * wv + Mfma<T>::row(lane, reg), r = cb + NB * q + rr;
        xs[(NB * q + rr) * R + rhs] = r < w ? -X[reg] : T(0);
        if (r < w) {
          if (fault != b + 1) sweepPublish<T>(xq + (int64_t)r * R + rhs, -X[reg]);
        }
      }
      __syncthreads();
    }
  }
  if (trace && tid == 0) trace[8 * b + 3] = (long long)wall_clock64();  // published
  // the solution itself, AFTER every publish (a scattered store in front of a publish delays it: the
  // memory pipe is in order), from LDS, 16 lanes along a right-hand side's rows: 128-byte runs
#pragma unroll
  for (int e = 0; e < W / 16; e++) {
    const int row = (tid & 15) + 16 * e, c = tid >> 4;
    if (cb + row < w && c < nr) vec0[(int64_t)c * ldc + sd.vecOff + cb + row] = -xs[row * R + c];
  }
}

// ---- far (both directions) ------------------------------------------------------------------------
// Forward (BACKWARD = false): rows [row0, +48) of the run (below = false) or below it against source
// blocks 0 .. nSrc-1; backward: columns [48 tile, +48) against the chunks of rows below the run, then
// the row blocks nB-1 .. b+2.  Wave wv < 3 owns 16 of the 48 targets; wave 3 polls.
template <typename T, bool BACKWARD>
__device__ __forceinline__ void sweepMFar(const SweepDesc& sd, int tile, bool below, int nr, GP<const T> A,
                                          GP<T> vec0, int64_t ldc, GP<const T> xq, GP<T> farq,
                                          const int32_t* rowGlobal, SweepWatch& watch, T* lds) {
  constexpr int W = kSweepW, R = kSweepR, RING = kSweepMRing, KB = W / 4;  // 48 k-blocks per source
  using Acc = typename Mfma<T>::Acc;
  volatile int* seq = reinterpret_cast<volatile int*>(lds);  // [0]: sources staged, [1..3]: consumed per wave
  T* ring = lds + kSweepLdsHead;                                // RING slots of [W][R]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lda = sd.lda, w = sd.w, nB = sd.nBlocks;
  const int rhs = lane & 15;
  int nC = 0, nSrc;
  if (!BACKWARD) {
    nSrc = below ? nB : tile / kSweepFarPerBlock - 1;
    if (!below && kSweepFarRows * tile >= w) return;
  } else {
    nC = (sd.rowsBelow + W - 1) / W;
    nSrc = nC + max(0, nB - tile / kSweepFarPerBlock - 2);
    if (kSweepFarRows * tile >= w) return;
  }
  if (nSrc <= 0) return;
  if (tid < 4) seq[tid] = 0;
  __syncthreads();
  // source s -> first matrix row of its 192 x-rows (backward: rows of L; forward: columns of L)
  auto srcRow = [&](int s) { return !BACKWARD ? W * s : (s < nC ? w + W * s : W * (nB - 1 - (s - nC))); };
  if (wv == 3) {  // the poller: x of source s, in B-operand order, into ring slot s % RING
    for (int s = 0; s < nSrc; s++) {
      while (s >= RING && min(min(seq[1], seq[2]), seq[3]) <= s - RING) __builtin_amdgcn_s_sleep(1);
      T* slot = ring + (s % RING) * W * R;
      const int r0 = srcRow(s);
      if (BACKWARD && s < nC) {  // rows below the run: final since before this launch
        for (int kb = 0; kb < KB; kb++) {
          const int q = r0 - w + 4 * kb + (lane >> 4);
          const T x = (q < sd.rowsBelow && rhs < nr)
                          ? -vec0[(int64_t)rhs * ldc + sweepTargetRow(sd, rowGlobal, q)]
                          : T(0);
          slot[kb * 64 + lane] = x;
        }
      } else {
        // (all 48 words of a lane in ONE wait: four waits of 12 were four fabric round trips per block)
        GP<const T> p[KB];
        bool need[KB];
        T got[KB];
#pragma unroll
        for (int e = 0; e < KB; e++) {
          const int row = r0 + 4 * e + (lane >> 4);
          need[e] = row < w;
          p[e] = xq + (need[e] ? (int64_t)row * R + rhs : 0);
          got[e] = T(0);
        }
        if (!sweepWaitCanary<T, KB>(p, need, got, watch, KB - 1)) {
          if (lane == 0) seq[0] = -1;
          return;
        }
#pragma unroll
        for (int e = 0; e < KB; e++) slot[e * 64 + lane] = got[e];
      }
      sweepLdsWait();
      if (lane == 0) seq[0] = s + 1;
    }
    return;
  }
  // compute waves: 16 targets each
  const int t0 = (BACKWARD ? 0 : (below ? w : 0)) + kSweepFarRows * tile + 16 * wv;  // first target (row / column)
  const int tEnd = BACKWARD ? w : (below ? w + sd.rowsBelow : w);
  T buf[3][KB];
  Acc D = {0, 0, 0, 0};
  auto load = [&](T(&dst)[KB], int s) {
    const int r0 = srcRow(s);
    const int lim = (BACKWARD && s < nC) ? w + sd.rowsBelow : w;  // sources beyond it meet x = 0
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const int i = min(t0 + (lane & 15), tEnd - 1), k = min(r0 + 4 * kb + (lane >> 4), lim - 1);
      dst[kb] = BACKWARD ? A[(int64_t)k * lda + min(i, w - 1)] : A[(int64_t)i * lda + min(k, w - 1)];
    }
  };
  auto step = [&](T(&cur)[KB], T(&nxt)[KB], int s) -> bool {
    int q;
    while ((q = seq[0]) >= 0 && q <= s) __builtin_amdgcn_s_sleep(1);
    if (q < 0) return false;
    if (s + 2 < nSrc) load(nxt, s + 2);
    const T* slot = ring + (s % RING) * W * R;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) D = Mfma<T>::run(cur[kb], slot[kb * 64 + lane], D);
    sweepLdsWait();
    if (lane == 0) seq[1 + wv] = s + 1;
    return true;
  };
  load(buf[0], 0);
  if (nSrc > 1) load(buf[1], 1);
  for (int s = 0; s < nSrc; s += 3) {
    if (!step(buf[0], buf[2], s)) return;
    if (s + 1 >= nSrc) break;
    if (!step(buf[1], buf[0], s + 1)) return;
    if (s + 2 >= nSrc) break;
    if (!step(buf[2], buf[1], s + 2)) return;
  }
#pragma unroll
  for (int reg = 0; reg < 4; reg++) {
    const int t = t0 + Mfma<T>::row(lane, reg);
    if (t < tEnd) {
      if (!BACKWARD && below) {
        if (rhs < nr) atomicSub(vec0 + (int64_t)rhs * ldc + sweepTargetRow(sd, rowGlobal, t - w), -D[reg]);
      } else {
        sweepPublish<T>(farq + (int64_t)t * R + rhs, D[reg]);
      }
    }
  }
}

// blockIdx.y = group of 16 right-hand sides, blockIdx.z = batch entry
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void solveSweepM(
    SweepDesc sd, int nRHS, const T* invBase, int64_t invBatchStride, T* xchg, SweepShared sh,
    const int32_t* rowGlobal, SolveRef<T> ref) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sweepLdsRaw[];
  T* lds = reinterpret_cast<T*>(sweepLdsRaw);
  __shared__ int sTicket;
  const int inst = blockIdx.z * gridDim.y + blockIdx.y;
  if (threadIdx.x == 0) sTicket = (int)(atomicAdd(sh.ctl + 1 + sd.ticketOff + inst, 1u) + 1u);
  __syncthreads();
  const int k = sTicket;
  SweepWatch watch;
  watch.abortWord = (GP<unsigned>)sh.ctl;
  watch.hostErr = sh.hostErr;
  watch.limit = sh.spinLimit;
  watch.reset();
  GP<const T> A = solveMat(ref) + sd.diagOff;
  const int nr = min(kSweepR, nRHS - kSweepR * (int)blockIdx.y);
  GP<T> vec0 = solveVecBase(ref) + (int64_t)kSweepR * blockIdx.y * ref.ldc;  // right-hand side 0 of the group
  GP<T> xq = (GP<T>)xchg + (int64_t)inst * sh.instStride + (int64_t)sd.xchgOff * kSweepR;
  GP<T> farq = xq + (int64_t)sd.nBlocks * kSweepW * kSweepR;
  GP<const T> inv = (GP<const T>)invBase + (int64_t)blockIdx.z * invBatchStride;
  constexpr int G = kSweepFarPerBlock + 1;
  const int blk = k / G, r = k % G;
  if (blk >= sd.nBlocks) {
    if (!BACKWARD) {
      sweepMFar<T, false>(sd, k - G * sd.nBlocks, true, nr, A, vec0, ref.ldc, xq, farq, rowGlobal, watch, lds);
    }
    return;
  }
  const int b = BACKWARD ? sd.nBlocks - 1 - blk : blk;
  {
    const long long wait = r == G - 1 ? (blk >= 3 ? min(blk - 2, 30) * 300ll : 0ll) : 500ll;  // 10-ns ticks
    if (wait > 0) {
      const long long t0 = (long long)wall_clock64();
      while ((long long)wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
  }
  if (r == G - 1) {
    sweepMSpine<T, BACKWARD>(sd, b, nr, A, vec0, ref.ldc, inv, xq, farq, watch, lds, sh.fault,
                             (blockIdx.y | blockIdx.z) ? nullptr : sh.trace);
  } else {
    sweepMFar<T, BACKWARD>(sd, kSweepFarPerBlock * b + r, false, nr, A, vec0, ref.ldc, xq, farq, rowGlobal, watch,
                           lds);
  }
}

template <typename T>
inline size_t sweepMLdsBytes() {
  const size_t spine = (size_t)(kSweepQ * kPanelWidth * kPanelWidth + 2 * kSweepW * kSweepR + kPanelWidth * kSweepR);
  const size_t far = (size_t)kSweepLdsHead + (size_t)kSweepMRing * kSweepW * kSweepR;
  return std::max(spine, far) * sizeof(T);
}

}  // namespace hipk
}  // namespace BaSpaCho
