// Persistent, flag-synchronised triangular sweeps over the serial chain of a WIDE lump (round 6).
//
// The multi-launch path (hip_solve_kernels.h, K-B1i / K-B2) walks a wide lump one 256-column block
// per step with two dependent launches per block: BAL-871's 7839-wide camera lump = 62 + 62
// launches of 7-13 us that move 246 MB -- 0.04 ms of HBM time -- in 1.18 ms.  Here ONE launch per
// direction walks the whole run.  Replaces the cublas trsm / gemv chains of MatOpsCuda.cu:1093-1181.
//
// A run = w consecutive columns of one lump (its one-panel levels) + the R rows below them, cut into
// blocks of kSweepW = 192 columns (three 64-column panels).  Workgroups take ROLES:
//   spine(b)   one per block, the serial chain.  Before its turn it pulls everything its step needs
//              on chip: the three inverted 64 x 64 diagonal blocks (LDS, from K-B0), the block's three
//              strictly-lower tiles and the nine tiles that couple it to the PREVIOUS block of the
//              sweep (registers: 12 tiles x 16 values per lane).  Its step: wait for x(prev block) and
//              for the far sums of its own rows, one 192 x 192 product, three panel solves, publish.
//              The spine never reads the matrix on the critical path.
//   far(t)     streams everything further from the diagonal: forward, a 48-row tile against every
//              block up to two blocks back (row sums, butterfly at the very end); backward, a 64-column
//              tile against every row block from two blocks on (and the rows below the run).  Three
//              compute waves keep 2-3 tiles of prefetch in flight; the fourth wave is the POLLER: it
//              alone spins on the published x and hands it to the others through LDS, because a vector
//              load that polls returns behind every prefetch load issued before it (vmcnt is in order).
// Dependencies go through an exchange buffer of SELF-VALIDATING words: armed to all-ones by K-B0,
// every value is published with one agent-scope relaxed store and consumed by polling the value
// itself (an all-ones result is canonicalised to the default NaN first) -- no flag, no fence, one
// fabric round trip per hop.  Roles are dealt by a ticket taken when a workgroup STARTS, in an order
// in which a role only ever waits for smaller tickets: whatever the dispatcher does, everything a
// workgroup waits for is already resident, so the launch cannot deadlock, co-resident or not.
// WATCHDOG: every spin is bounded by a wall-clock limit; on expiry the wave raises the launch's
// abort word (every other spin sees it within a few polls) and a word in pinned host memory, and
// leaves.  The host reports the failed call and retires the sweeps of that Solver for good
// (multi-launch path from then on).
#pragma once

#include <hip/hip_runtime.h>

#include "hip_solve_kernels.h"

namespace BaSpaCho {
namespace hipk {

constexpr int kSweepQ = 3;                                  // panels per block
constexpr int kSweepW = kSweepQ * kPanelWidth;              // columns per step
constexpr int kSweepFarRows = 48;                           // forward far tile: 3 compute waves x 16 rows
constexpr int kSweepFarPerBlock = kSweepW / kSweepFarRows;  // 4
constexpr int kSweepFwdGroup = kSweepFarPerBlock + 1;       // roles per block, forward
constexpr int kSweepBwdGroup = kSweepQ + 1;                 // ... backward (3 column tiles + spine)
constexpr int kSweepLdsHead = 256;                          // far roles: progress word + partial sums (values)

struct SweepDesc {
  int64_t diagOff;      // data offset of element (0, 0) of the run
  int32_t lda;          // row stride (lump width)
  int32_t w;            // columns of the run
  int32_t rowsBelow;    // rows below the run (rest of the lump + its chain rows)
  int32_t nRest;        // of which inside the lump (linear addressing)
  int32_t lumpRowBase;  // as PanelDesc
  int32_t vecOff;       // position of the run's first column in the vector
  int32_t nBlocks;      // ceil(w / kSweepW)
  int32_t invSlot;      // slot of the run's first panel in the inverse scratch
  int32_t xchgOff;      // element offset of the run's area inside an instance's slice of the exchange buffer
  int32_t ticketOff;    // index of the run's first ticket word (one per instance)
};

struct SweepShared {
  unsigned* ctl;        // word 0: abort (armed = all ones, 0 = abort); words 1..: tickets
  unsigned* hostErr;    // pinned host memory: set on a time-out
  long long spinLimit;  // wall_clock64() ticks (100 MHz)
  int64_t instStride;   // values per instance (right-hand side x batch entry) of the exchange buffer
  int32_t fault;        // TESTING (bsp_test_set_fault): block whose spine never publishes, + 1
  int32_t pad;
  long long* trace;     // developer aid (BSP_SWEEP_TRACE=1): per spine {start, loaded, polled, done}, 100 MHz
};

template <typename T>
struct SweepBits;
template <>
struct SweepBits<double> {
  using U = unsigned long long;
  static constexpr U kArmed = ~0ull, kNan = 0x7ff8000000000000ull;
};
template <>
struct SweepBits<float> {
  using U = unsigned;
  static constexpr U kArmed = ~0u, kNan = 0x7fc00000u;
};

template <typename T>
__device__ __forceinline__ T sweepPeek(GP<const T> p) {
  using U = typename SweepBits<T>::U;
  const U u = __hip_atomic_load((GP<const U>)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(T, u);
}
template <typename T>
__device__ __forceinline__ bool sweepValid(T v) {
  using U = typename SweepBits<T>::U;
  return __builtin_bit_cast(U, v) != SweepBits<T>::kArmed;
}
template <typename T>
__device__ __forceinline__ void sweepPublish(GP<T> p, T v) {
  using U = typename SweepBits<T>::U;
  U u = __builtin_bit_cast(U, v);
  if (u == SweepBits<T>::kArmed) u = SweepBits<T>::kNan;
  __hip_atomic_store((GP<U>)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct SweepWatch {
  GP<unsigned> abortWord;
  unsigned* hostErr;
  long long limit;
  long long t0;
  unsigned it;
  __device__ __forceinline__ void reset() {
    t0 = 0;
    it = 0;
  }
  // after an unsuccessful poll: true = give up (wave-uniform)
  __device__ __forceinline__ bool expired() {
    if ((++it & 15u) != 0u) return false;
    if (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return true;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) {
      t0 = now;
      return false;
    }
    if (now - t0 <= limit) return false;
    __hip_atomic_store(abortWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hostErr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
};

// every lane polls its own N words until all of them (in every lane) have been published
template <typename T, int N>
__device__ __forceinline__ bool sweepWait(GP<const T> (&p)[N], const bool (&need)[N], T (&out)[N],
                                          SweepWatch& watch) {
  watch.reset();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (need[i]) {
        out[i] = sweepPeek<T>(p[i]);
        ok = ok && sweepValid(out[i]);
      }
    }
    if (__all(ok)) return true;
    if (watch.expired()) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// ---- sum 16 per-lane values across the 64 lanes, 16 sums at once, WITHOUT the LDS crossbar ----------
// waveSum16 (hip_solve_kernels.h) is 17 __shfl_xor = 34 ds_bpermute for doubles, a dependent chain
// of crossbar round trips: 0.6 us per call, and a spine step makes eight (first trace of the sweep:
// 5.5 of a step's 6.6 us).  Here the two wide exchanges are gfx950's v_permlane32_swap /
// v_permlane16_swap (tools/permlane_probe.hip: swap(a, b) -> [a.even b.even ...] and [a.odd b.odd ...]
// halves / rows, so that r0 + r1 IS the halving step of the butterfly), the two narrow ones DPP quad
// permutes, and the last two plain row rotations: vector-ALU instructions only.
// The sum of v[sweepRowOf(lane)] ends up in every lane; lanes with sweepRowOwner() write it.
__device__ __forceinline__ int sweepRowOf(int lane) { return ((lane >> 4) << 2) | (lane & 3); }
__device__ __forceinline__ bool sweepRowOwner(int lane) { return (lane & 12) == 0; }
template <int CTRL>
__device__ __forceinline__ double sweepDpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float sweepDpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double sweepHalve32(double a, double b) {  // lanes < 32: a[l] + a[l+32]; else b[l-32] + b[l]
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ float sweepHalve32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double sweepHalve16(double a, double b) {  // even rows: a over {row, row+1}; odd rows: b
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ float sweepHalve16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <typename T>
__device__ __forceinline__ T sweepRowSum16(T (&v)[16], int lane) {
  const bool b1 = lane & 2, b0 = lane & 1;
  T a[8], b[4], c[2];
#pragma unroll
  for (int j = 0; j < 8; j++) a[j] = sweepHalve32(v[j], v[j + 8]);
#pragma unroll
  for (int j = 0; j < 4; j++) b[j] = sweepHalve16(a[j], a[j + 4]);
#pragma unroll
  for (int j = 0; j < 2; j++) c[j] = (b1 ? b[j + 2] : b[j]) + sweepDpp<0x4E>(b1 ? b[j] : b[j + 2]);  // quad_perm [2,3,0,1]
  T d = (b0 ? c[1] : c[0]) + sweepDpp<0xB1>(b0 ? c[0] : c[1]);                                       // quad_perm [1,0,3,2]
  d += sweepDpp<0x124>(d);  // row_ror:4
  d += sweepDpp<0x128>(d);  // row_ror:8
  return d;
}

// ... the same, but spinning on ONE of the words (the one published last) until it is there: a
// poll of N words by a few hundred waiting waves is N times the traffic on the lines everybody is
// waiting for, and slows the very publish it waits for (matrix-core sweep: 48 words per lane and poller)
template <typename T, int N>
__device__ __forceinline__ bool sweepWaitCanary(GP<const T> (&p)[N], const bool (&need)[N], T (&out)[N],
                                                SweepWatch& watch, int canary) {
  watch.reset();
  for (;;) {
    const bool ok = !need[canary] || sweepValid(sweepPeek<T>(p[canary]));
    if (__all(ok)) break;
    if (watch.expired()) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  return sweepWait<T, N>(p, need, out, watch);
}

// position in the vector of row q below the run
__device__ __forceinline__ int sweepTargetRow(const SweepDesc& sd, const int32_t* rowGlobal, int q) {
  return q < sd.nRest ? sd.vecOff + sd.w + q : rowGlobal[sd.lumpRowBase + (q - sd.nRest)];
}

__device__ __forceinline__ void sweepLdsWait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- spine: block b of the run ---------------------------------------------------------------------
// forward:  x_b = T_b^-1  (y_b - far_b - L[b, b-1]   x_{b-1})
// backward: x_b = T_b^-T  (y_b - far_b - L[b+1, b]^T x_{b+1})
// Both as ROW sums (lane = column of the operand tile, 16 rows per wave, one butterfly per product):
// the backward sweep simply loads its tiles, and the inverses, transposed -- uncoalesced, but long
// before its turn comes.
template <typename T, bool BACKWARD>
__device__ __forceinline__ void sweepSpine(const SweepDesc& sd, int b, GP<const T> A, GP<T> vec,
                                           GP<const T> inv, GP<T> xq, GP<const T> farq,
                                           SweepWatch& watch, T* lds, int fault, long long* trace) {
  constexpr int NB = kPanelWidth, Q = kSweepQ, W = kSweepW;
  T* Iv = lds;                // Q inverses, 64 x 64 each (backward: transposed)
  T* xs = Iv + Q * NB * NB;   // x of this block
  T* ts = xs + W;             // right-hand side of the panel being solved
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lda = sd.lda, w = sd.w;
  const int cb = b * W;
  const int nq = min(Q, (w - cb + NB - 1) / NB);
  const int bp = BACKWARD ? b + 1 : b - 1;
  const bool hasPrev = BACKWARD ? bp < sd.nBlocks : bp >= 0;
  const int cp = bp * W;
  const bool hasFar = BACKWARD ? (sd.rowsBelow > 0 || b + 2 < sd.nBlocks) : b >= 2;
  const int ur = sweepRowOf(lane);
  if (trace && tid == 0) trace[8 * b] = (long long)wall_clock64();

  // everything the step needs, requested before the wait.  Forward: tile rows are matrix rows, a lane
  // reads along a row.  Backward the tiles (and the inverses) are needed TRANSPOSED: read the natural,
  // coalesced way (next tile in flight) and turned round through an LDS staging tile -- the first
  // version read them transposed from memory, 64 lines per load: 50-70 us per spine, and the first
  // two spines of the sweep waited for theirs.
  T Lp[Q][Q][16], Li[Q * (Q - 1) / 2][16];
  // (every load unconditional, from a clamped row, masked afterwards: a predicated load is a branch
  //  around it, 600 of them in the first version)
  if (!BACKWARD) {
#pragma unroll
    for (int q = 0; q < Q; q++) {
#pragma unroll
      for (int pp = 0; pp < Q; pp++) {
#pragma unroll
        for (int u = 0; u < 16; u++) Lp[q][pp][u] = T(0);
      }
    }
    if (hasPrev) {
#pragma unroll
      for (int q = 0; q < Q; q++) {
#pragma unroll
        for (int pp = 0; pp < Q; pp++) {
#pragma unroll
          for (int u = 0; u < 16; u++) {
            const int i = cb + NB * q + 16 * wv + u, j = cp + NB * pp + lane;
            const T a = A[(int64_t)min(i, w - 1) * lda + j];
            Lp[q][pp][u] = i < w ? a : T(0);
          }
        }
      }
    }
#pragma unroll
    for (int hi = 1; hi < Q; hi++) {
#pragma unroll
      for (int lo = 0; lo < hi; lo++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {  // target panel hi, source panel lo
          const int i = cb + NB * hi + 16 * wv + u, j = cb + NB * lo + lane;
          const T a = A[(int64_t)min(i, w - 1) * lda + min(j, w - 1)];
          Li[hi * (hi - 1) / 2 + lo][u] = i < w ? a : T(0);
        }
      }
    }
    GP<const T> src = inv + (int64_t)(sd.invSlot + b * Q) * NB * NB;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      T v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = q < nq ? src[(int64_t)q * NB * NB + tid + 256 * i] : T(0);
#pragma unroll
      for (int i = 0; i < 16; i++) Iv[q * NB * NB + tid + 256 * i] = v[i];
    }
  } else {
    constexpr int LD = NB + 1;
    T* St = ts + NB;  // staging tile, NB x LD
    // natural layout: rows = source index (rows of L), lane = target index (its column); every load
    // of the spine in flight at once, then the tiles are turned round in place, one after the other
    auto fetch = [&](T(&dst)[16], int r0, int c0) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int j = r0 + 16 * wv + u;
        const T a = A[(int64_t)min(j, w - 1) * lda + min(c0 + lane, w - 1)];
        dst[u] = j < w ? a : T(0);
      }
    };
    auto turn = [&](T(&tile)[16]) {
#pragma unroll
      for (int u = 0; u < 16; u++) St[(16 * wv + u) * LD + lane] = tile[u];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 16; u++) tile[u] = St[lane * LD + 16 * wv + u];
      __syncthreads();
    };
#pragma unroll
    for (int q = 0; q < Q; q++) {
#pragma unroll
      for (int pp = 0; pp < Q; pp++) {
        if (hasPrev) {
          fetch(Lp[q][pp], cp + NB * pp, cb + NB * q);
        } else {
#pragma unroll
          for (int u = 0; u < 16; u++) Lp[q][pp][u] = T(0);
        }
      }
    }
#pragma unroll
    for (int hi = 1; hi < Q; hi++) {
#pragma unroll
      for (int lo = 0; lo < hi; lo++) fetch(Li[hi * (hi - 1) / 2 + lo], cb + NB * hi, cb + NB * lo);
    }
    if (hasPrev) {
#pragma unroll
      for (int q = 0; q < Q; q++) {
#pragma unroll
        for (int pp = 0; pp < Q; pp++) turn(Lp[q][pp]);
      }
    }
#pragma unroll
    for (int k = 0; k < Q * (Q - 1) / 2; k++) turn(Li[k]);
    GP<const T> src = inv + (int64_t)(sd.invSlot + b * Q) * NB * NB;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      T vi[16];
#pragma unroll
      for (int i = 0; i < 16; i++) vi[i] = q < nq ? src[(int64_t)q * NB * NB + tid + 256 * i] : T(0);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int e = tid + 256 * i;
        Iv[q * NB * NB + (e & 63) * NB + (e >> 6)] = vi[i];
      }
    }
  }
  T yv[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int r = cb + NB * q + 16 * wv + ur;
    const T y = vec[sd.vecOff + min(r, w - 1)];
    yv[q] = r < w ? y : T(0);
  }
  if (tid < W) xs[tid] = T(0);
  __syncthreads();
  if (trace && tid == 0) trace[8 * b + 1] = (long long)wall_clock64();

  GP<const T> pp[2 * Q];
  bool need[2 * Q];
  T got[2 * Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int j = cp + NB * q + lane, r = cb + NB * q + 16 * wv + ur;
    need[q] = hasPrev && j < w;
    pp[q] = (GP<const T>)xq + (need[q] ? j : 0);
    need[Q + q] = hasFar && r < w;
    pp[Q + q] = farq + (need[Q + q] ? r : 0);
    got[q] = T(0);
    got[Q + q] = T(0);
  }
  if (!sweepWait<T, 2 * Q>(pp, need, got, watch)) return;
  if (trace && tid == 0) trace[8 * b + 2] = (long long)wall_clock64();

  T t[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    T v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      v[u] = Lp[q][0][u] * got[0] + Lp[q][1][u] * got[1] + Lp[q][2][u] * got[2];
    }
    t[q] = yv[q] - got[Q + q] - sweepRowSum16(v, lane);
  }
#pragma unroll
  for (int qi = 0; qi < Q; qi++) {
    const int q = BACKWARD ? Q - 1 - qi : qi;
    if (q < nq) {
      if (qi > 0) {
        // panels of this block solved before this one (x of the others is still zero)
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = T(0);
#pragma unroll
        for (int p = 0; p < Q; p++) {
          if (BACKWARD ? p > q : p < q) {
            const int hi = p > q ? p : q, lo = p > q ? q : p;
            const T xp = xs[NB * p + lane];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] += Li[hi * (hi - 1) / 2 + lo][u] * xp;
          }
        }
        t[q] -= sweepRowSum16(v, lane);
      }
      if (sweepRowOwner(lane)) ts[16 * wv + ur] = t[q];
      __syncthreads();
      const T tq = ts[lane];
      T v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = Iv[q * NB * NB + (16 * wv + u) * NB + lane] * tq;
      const T xv = sweepRowSum16(v, lane);
      const int r = cb + NB * q + 16 * wv + ur;
      if (sweepRowOwner(lane)) {
        xs[NB * q + 16 * wv + ur] = xv;
        if (r < w) {
          if (fault != b + 1) sweepPublish<T>(xq + r, xv);
          vec[sd.vecOff + r] = xv;
        }
      }
      __syncthreads();
    }
  }
  if (trace && tid == 0) trace[8 * b + 3] = (long long)wall_clock64();
}

// ---- far, forward: rows [row0, row0 + 48) against source blocks 0 .. nSrc-1 ------------------------
// below = false: rows of the run itself, the sum goes to farq (the spine of their block waits for it);
// below = true: rows below the run, the sum is subtracted from the vector.
template <typename T>
__device__ __forceinline__ void sweepFarL(const SweepDesc& sd, int tile, bool below, GP<const T> A,
                                          GP<T> vec, GP<const T> xq, GP<T> farq,
                                          const int32_t* rowGlobal, SweepWatch& watch, T* lds) {
  constexpr int W = kSweepW;
  volatile int* seq = reinterpret_cast<volatile int*>(lds);
  T* xall = lds + kSweepLdsHead;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lda = sd.lda, w = sd.w;
  const int nSrc = below ? sd.nBlocks : tile / kSweepFarPerBlock - 1;
  if (nSrc <= 0) return;
  if (!below && kSweepFarRows * tile >= w) return;
  if (tid == 0) *seq = 0;
  __syncthreads();
  if (wv == 3) {  // the poller
    for (int s = 0; s < nSrc; s++) {
      GP<const T> p[3];
      bool need[3];
      T got[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int c = W * s + 64 * i + lane;
        need[i] = c < w;
        p[i] = xq + (need[i] ? c : 0);
        got[i] = T(0);
      }
      if (!sweepWait<T, 3>(p, need, got, watch)) {
        if (lane == 0) *seq = -1;
        return;
      }
#pragma unroll
      for (int i = 0; i < 3; i++) xall[W * s + 64 * i + lane] = got[i];
      sweepLdsWait();
      if (lane == 0) *seq = s + 1;
    }
    return;
  }
  const int rowBase = (below ? w : 0) + kSweepFarRows * tile + 16 * wv;
  const int rowEnd = below ? w + sd.rowsBelow : w;
  T buf[3][16][3], acc[16];
#pragma unroll
  for (int u = 0; u < 16; u++) acc[u] = T(0);
  auto load = [&](T(&dst)[16][3], int s) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      GP<const T> row = A + (int64_t)min(rowBase + u, rowEnd - 1) * lda;
#pragma unroll
      // (no mask: a clamped element meets x = 0 for a column beyond the run, and a clamped row is
      //  never published -- a select here would make every prefetch wait for its own loads)
      for (int i = 0; i < 3; i++) dst[u][i] = row[min(W * s + 64 * i + lane, w - 1)];
    }
  };
  auto step = [&](T(&cur)[16][3], T(&nxt)[16][3], int s) -> bool {
    int q;
    while ((q = *seq) >= 0 && q <= s) __builtin_amdgcn_s_sleep(1);
    if (q < 0) return false;
    if (s + 2 < nSrc) load(nxt, s + 2);
    const T x0 = xall[W * s + lane], x1 = xall[W * s + 64 + lane], x2 = xall[W * s + 128 + lane];
#pragma unroll
    for (int u = 0; u < 16; u++) acc[u] += cur[u][0] * x0 + cur[u][1] * x1 + cur[u][2] * x2;
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
  const T sum = sweepRowSum16(acc, lane);
  const int r = rowBase + sweepRowOf(lane);
  if (sweepRowOwner(lane) && r < rowEnd) {
    if (below) {
      atomicSub(vec + sweepTargetRow(sd, rowGlobal, r - w), sum);
    } else {
      sweepPublish<T>(farq + r, sum);
    }
  }
}

// ---- far, backward: columns [64 ctile, +64) against the rows below the run, then the row blocks
// nB-1 .. b+2 (b = the tile's block): lane = column, a wave takes a third of the rows of a unit
// (96 rows: half a block, so that four units of prefetch fit the registers), column sums per lane,
// one cross-wave reduction at the end
template <typename T>
__device__ __forceinline__ void sweepFarLt(const SweepDesc& sd, int ctile, GP<const T> A, GP<T> vec,
                                           GP<const T> xq, GP<T> farq, const int32_t* rowGlobal,
                                           SweepWatch& watch, T* lds) {
  constexpr int W = kSweepW, H = W / 2, RW = H / 3;  // 96-row units, 32 rows per wave
  volatile int* seq = reinterpret_cast<volatile int*>(lds);
  T* part = lds + 64;  // [3][64]
  T* xall = lds + kSweepLdsHead;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lda = sd.lda, w = sd.w;
  const int nB = sd.nBlocks, b = ctile / kSweepQ, wPad = nB * W;
  const int nC = (sd.rowsBelow + W - 1) / W;
  const int nBlk = max(0, nB - b - 2);
  const int nSrc = nC + nBlk;
  if (nSrc <= 0 || 64 * ctile >= w) return;
  if (tid == 0) *seq = 0;
  __syncthreads();
  if (wv == 3) {
    for (int s = 0; s < nSrc; s++) {
      if (s < nC) {  // rows below the run: final since the kernels before this launch
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int q = W * s + 64 * i + lane;
          xall[wPad + q] = q < sd.rowsBelow ? vec[sweepTargetRow(sd, rowGlobal, q)] : T(0);
        }
      } else {
        const int p = nB - 1 - (s - nC);
        GP<const T> pt[3];
        bool need[3];
        T got[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int c = W * p + 64 * i + lane;
          need[i] = c < w;
          pt[i] = xq + (need[i] ? c : 0);
          got[i] = T(0);
        }
        if (!sweepWait<T, 3>(pt, need, got, watch)) {
          if (lane == 0) *seq = -1;
          return;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) xall[W * p + 64 * i + lane] = got[i];
      }
      sweepLdsWait();
      if (lane == 0) *seq = s + 1;
    }
  } else {
    const int col = 64 * ctile + lane;
    T buf[4][RW], acc = T(0);
    const int nUnits = 2 * nSrc;
    // unit -> first matrix row, first x slot, row limit
    auto unitRow = [&](int unit, int& m0, int& x0, int& lim) {
      const int s = unit >> 1, h = unit & 1;
      if (s < nC) {
        m0 = w + W * s + H * h;
        x0 = wPad + W * s + H * h;
        lim = w + sd.rowsBelow;
      } else {
        const int p = nB - 1 - (s - nC);
        m0 = W * p + H * h;
        x0 = m0;
        lim = w;
      }
    };
    auto load = [&](T(&dst)[RW], int unit) {
      int m0, x0, lim;
      unitRow(unit, m0, x0, lim);
#pragma unroll
      for (int u = 0; u < RW; u++) {
        const int m = m0 + RW * wv + u;
        // (no mask: x is zero for a row beyond the unit's limit, a column beyond the run is not published)
        dst[u] = A[(int64_t)min(m, lim - 1) * lda + min(col, w - 1)];
      }
    };
    auto step = [&](T(&cur)[RW], T(&nxt)[RW], int unit) -> bool {
      int q;
      while ((q = *seq) >= 0 && q <= (unit >> 1)) __builtin_amdgcn_s_sleep(1);
      if (q < 0) return false;
      if (unit + 3 < nUnits) load(nxt, unit + 3);
      int m0, x0, lim;
      unitRow(unit, m0, x0, lim);
      const T* xr = xall + x0 + RW * wv;
#pragma unroll
      for (int u = 0; u < RW; u++) acc += cur[u] * xr[u];
      return true;
    };
    load(buf[0], 0);
    load(buf[1], 1);
    if (nUnits > 2) load(buf[2], 2);
    bool ok = true;
    for (int unit = 0; unit < nUnits && ok; unit += 4) {
      ok = step(buf[0], buf[3], unit);
      if (!ok || unit + 1 >= nUnits) break;
      ok = step(buf[1], buf[0], unit + 1);
      if (!ok || unit + 2 >= nUnits) break;
      ok = step(buf[2], buf[1], unit + 2);
      if (!ok || unit + 3 >= nUnits) break;
      ok = step(buf[3], buf[2], unit + 3);
    }
    if (!ok) return;
    part[wv * 64 + lane] = acc;
  }
  __syncthreads();
  if (wv == 0) {
    const int col = 64 * ctile + lane;
    if (col < w) sweepPublish<T>(farq + col, part[lane] + part[64 + lane] + part[128 + lane]);
  }
}

// One launch = one direction of one run; blockIdx.y = right-hand side, blockIdx.z = batch entry.
// Ticket k -> role.  Forward, per block b: far tiles 4b .. 4b+3 (they wait for spines <= b-2), then
// spine b (waits for spine b-1 and its block's far tiles); the tiles of the rows below the run
// last.  Backward the same over b = nB-1 .. 0 with three column tiles per block.
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void solveSweep(
    SweepDesc sd, const T* invBase, int64_t invBatchStride, T* xchg, SweepShared sh,
    const int32_t* rowGlobal, SolveRef<T> ref) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sweepLdsRaw[];
  T* lds = reinterpret_cast<T*>(sweepLdsRaw);
  __shared__ int sTicket;
  const int inst = blockIdx.z * gridDim.y + blockIdx.y;
  if (threadIdx.x == 0) {
    sTicket = (int)(atomicAdd(sh.ctl + 1 + sd.ticketOff + inst, 1u) + 1u);  // armed to all ones
  }
  __syncthreads();
  const int k = sTicket;
  SweepWatch watch;
  watch.abortWord = (GP<unsigned>)sh.ctl;
  watch.hostErr = sh.hostErr;
  watch.limit = sh.spinLimit;
  watch.reset();
  GP<const T> A = solveMat(ref) + sd.diagOff;
  GP<T> vec = solveVec(ref);
  GP<T> xq = (GP<T>)xchg + (int64_t)inst * sh.instStride + sd.xchgOff;
  GP<T> farq = xq + (int64_t)sd.nBlocks * kSweepW;
  GP<const T> inv = (GP<const T>)invBase + (int64_t)blockIdx.z * invBatchStride;
  constexpr int G = BACKWARD ? kSweepBwdGroup : kSweepFwdGroup;
  const int blk = k / G, r = k % G;
  if (blk >= sd.nBlocks) {
    if (!BACKWARD) sweepFarL<T>(sd, k - G * sd.nBlocks, true, A, vec, xq, farq, rowGlobal, watch, lds);
    return;
  }
  const int b = BACKWARD ? sd.nBlocks - 1 - blk : blk;
  // STAGGERED START.  Every role begins by pulling its operands on chip: 40 MB requested in the same
  // microsecond, and the spines of the first steps -- the only ones anybody is waiting for -- queued
  // behind all of it (first trace: 21 us before spine 1 had its tiles, 47 backward).  Spine k is not
  // needed before ~5 k us, the far tiles not before the first x is out: they start a little later.
  {
    const long long wait = r == G - 1 ? (blk >= 3 ? min(blk - 2, 30) * 200ll : 0ll) : 500ll;  // 10-ns ticks
    if (wait > 0) {
      const long long t0 = (long long)wall_clock64();
      while ((long long)wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
  }
  if (r == G - 1) {
    sweepSpine<T, BACKWARD>(sd, b, A, vec, inv, xq, farq, watch, lds, sh.fault,
                               (blockIdx.y | blockIdx.z) ? nullptr : sh.trace);
  } else if (!BACKWARD) {
    sweepFarL<T>(sd, kSweepFarPerBlock * b + r, false, A, vec, xq, farq, rowGlobal, watch, lds);
  } else {
    sweepFarLt<T>(sd, kSweepQ * b + r, A, vec, xq, farq, rowGlobal, watch, lds);
  }
}

// dynamic LDS of a launch, bytes
template <typename T>
inline size_t sweepLdsBytes(int w, int rowsBelow, bool backward) {
  const size_t spine = (size_t)(kSweepQ * kPanelWidth * kPanelWidth + kSweepW + kPanelWidth +
                                (backward ? kPanelWidth * (kPanelWidth + 1) : 0)) * sizeof(T);
  const size_t nB = (size_t)(w + kSweepW - 1) / kSweepW;
  size_t far = (size_t)kSweepLdsHead + nB * kSweepW;
  if (backward) far += (size_t)((rowsBelow + kSweepW - 1) / kSweepW) * kSweepW;
  return std::max(spine, far * sizeof(T));
}

}  // namespace hipk
}  // namespace BaSpaCho
