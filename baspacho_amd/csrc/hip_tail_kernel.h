// Persistent, flag-synchronised Cholesky of the TAIL of a wide root lump (round 6).
//
// The last outer blocks of a wide lump that has nothing below it (the camera block of a Schur
// complement, the root front of FLAT) are a plain dense Cholesky whose trailing matrix no longer
// fills the GPU: the level schedule walks them with one chainStep launch per 64-column panel, 17-19
// us each plus the event gaps of the lookahead streams -- ~95 us per 256 columns with a tenth of the
// machine working.  Here ONE launch factors the whole K x K tail.  Replaces the cusolverDnDpotrf +
// cublasDtrsm + cublasDgemm chain of MatOpsCuda.cu:508-590 for those columns.
//
// Workgroups take ROLES over the 64 x 64 tiles (i, j), i >= j, of the tail's lower triangle; a role
// keeps the rank-64 updates its tile has received so far in MFMA ACCUMULATORS for the whole launch
// (no read-modify-write of the trailing matrix at all) and meets the others only through flags:
//   tile (i, j), i >= j + 2:  for p < j: wait X(p, i), X(p, j) -> D += X_i^p (X_j^p)^T;  then wait
//        diag(j) -> X_i^j = (A_ij - D) L_jj^-T (register trsm through the inverted 16 x 16 diagonal
//        blocks, as chainStep), stored in place = the final entries of L -> flag X(j, i); done.
//   spine q:  owns BOTH the diagonal tile (q, q) and the tile left of it (q, q-1), so that the
//        serial chain crosses workgroups ONCE per panel: it accumulates both tiles for p < q-1, and
//        when diag(q-1) arrives it solves X_q^{q-1} itself, applies it to its diagonal tile, runs the
//        blocked panel Cholesky (potrfPanelTiles, with the pending update folded into its load) and
//        raises diag(q).  Its X_q^{q-1} is stored and flagged for the tiles of column q.
// Round-6 refinement (second trace): the spine also owns the tile TWO left of its diagonal, (q, q-2), and
// receives X_{q-1}^{q-2} from spine q-1 through an exchange slot of self-validating words (published with
// plain write-through stores the moment it is solved, polled as data: no store-release-flag on anybody's
// critical path).  The serial chain then runs from spine to spine only -- one flag hop per panel -- and
// every other tile has two full steps to deliver; tiles (q, q-1) and (q, q-2) exist as roles too: they
// repeat the solve and STORE the result in place for everybody else.
// Roles are dealt by ticket in an order in which a role only waits for smaller tickets (spine 0; then
// per column j: spine j+1, tiles (j+2.., j)), so the launch cannot deadlock whether or not all of it
// is resident (one exception, bounded: the tile roles (q, q-1) and (q, q-2) also wait for spine q's word
// that it has taken its copy of the unsolved tile, and spine q sits at most one column group later in
// the order -- the plan caps a tail at 128 panels, far inside the 512 roles the GPU holds at once);
// a role that starts late replays the finished panels at L2 speed.  Flags: release
// (agent) after the stores, relaxed polls, one acquire fence per wait (tools/flag_hop_probe.hip: 2.05
// us per hop with a 32-KB tile read and written).  Every spin is bounded by the watchdog of the
// persistent sweeps (SweepWatch): on expiry the launch aborts and the host retires the kernel.
#pragma once

#include <hip/hip_runtime.h>

#include "hip_kernels.h"
#include "hip_sweep_kernels.h"

namespace BaSpaCho {
namespace hipk {

struct TailDesc {
  int64_t diagOff;  // data offset of element (0, 0) of the tail
  int32_t lda;      // row stride (lump width)
  int32_t K;        // order of the tail
  int32_t nP;       // ceil(K / 64)
  int32_t ctlStride;  // control words per matrix: abort, ticket, nP diag flags, nP x nP X flags, nP copy-taken, yield
  int32_t flags;      // bit 0: tiles yield the CU of a spine that is in its panel Cholesky
  int32_t pad;
};
inline int tailCtlWords(int nP) { return 2 + nP + nP * nP + nP + 1; }  // (+ copy-taken words, yield word)
inline int tailRoles(int nP) { return 1 + (nP - 1) + nP * (nP - 1) / 2; }

// control words are armed to all ones (one memset): a flag is raised by storing 0
__device__ __forceinline__ void tailRaise(GP<unsigned> f) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every wave waits for itself (no workgroup barrier inside a wait); false = abort
__device__ __forceinline__ bool tailWait(GP<const unsigned> f0, GP<const unsigned> f1, SweepWatch& watch,
                                         GP<const unsigned> f2 = nullptr, bool acquire = true) {
  watch.reset();
  for (;;) {
    const unsigned a = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned b = f1 ? __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const unsigned c = f2 ? __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (a == 0u && b == 0u && c == 0u) break;
    if (watch.expired()) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

template <typename T>
struct TailTile {
  using Acc = typename Mfma<T>::Acc;
  // D[t] += X_i (X_j rows 16 t .. 16 t + 15)^T for source panel p: X_i = rows of tile row `ti`, X_j =
  // rows of tile row `tj`, both columns 64 p .. 64 p + 63 of the tail (final entries of L, in place)
  // (Exchanging the solved rows through agent-scope relaxed loads and stores instead of plain accesses
  //  between release / acquire fences was tried -- an acquire fence invalidates the whole L2 of its XCD,
  //  and with the tiles' fences switched off for a timing experiment 12 blocks took 920 instead of
  //  1180-1260 us.  It is WRONG on this memory system: a coherent load can return a line that a plain
  //  load of the unsolved tile left in the L2 (vector probe 4e-7), and it was slower besides, every
  //  operand coming from the far side of the fabric: 1600 us.)
  static __device__ __forceinline__ void fetchA(GP<const T> A, int lda, int K, int ti, int p, Acc (&xi)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15;
    const int ri = kTile * ti + 16 * w + n;
    trsmLoadRows<T>(A + (int64_t)min(ri, K - 1) * lda + kTile * p, kTile, lane, xi);
    trsmMaskRows<T>(ri < K, kTile, lane, xi);
  }
  static __device__ __forceinline__ void fetchB(GP<const T> A, int lda, int K, int tj, int p, T (&v)[16]) {
    const int tid = threadIdx.x;
    const int rj = kTile * tj + (tid >> 2);
    GP<const T> rowJ = A + (int64_t)min(rj, K - 1) * lda + kTile * p + 16 * (tid & 3);
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const T a = rowJ[c];
      v[c] = rj < K ? a : T(0);
    }
  }
  static __device__ __forceinline__ void fetch(GP<const T> A, int lda, int K, int ti, int tj, int p,
                                               Acc (&xi)[4], T (&v)[16]) {
    fetchA(A, lda, K, ti, p, xi);
    fetchB(A, lda, K, tj, p, v);
  }
  // D = -(tile (ti, tj)) in accumulator layout: with the products added on top, the solve's
  // right-hand side is -D and no copy of the unsolved tile has to stay in registers
  static __device__ __forceinline__ void loadNeg(GP<const T> A, int lda, int K, int ti, int tj, Acc (&D)[4]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int row = kTile * ti + 16 * w + Mfma<T>::row(lane, reg);
        const T a = A[(int64_t)min(row, K - 1) * lda + kTile * tj + 16 * t + n];
        D[t][reg] = row < K ? -a : T(0);
      }
    }
  }
  // x (row layout of trsmStages) -> XB as the B operand of a product against itself / another tile
  static __device__ __forceinline__ void stageX(const Acc (&x)[4], T* XB) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, qq = lane >> 4;
    ldsBarrier();
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
#pragma unroll
      for (int r = 0; r < 4; r++) XB[(16 * w + n) * kXbLd + 16 * jj + 4 * qq + r] = x[jj][r];
    }
    ldsBarrier();
  }
  static __device__ __forceinline__ void stageB(const T (&v)[16], T* XB) {
    const int tid = threadIdx.x;
    ldsBarrier();  // XB free
#pragma unroll
    for (int c = 0; c < 16; c++) XB[(tid >> 2) * kXbLd + 16 * (tid & 3) + c] = v[c];
    ldsBarrier();
  }
  static __device__ __forceinline__ void multiply(const Acc (&xi)[4], const T* XB, bool diag, Acc (&D)[4]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (!diag || t <= w) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            D[t] = Mfma<T>::run(xi[j][r], XB[(16 * t + n) * kXbLd + 16 * j + 4 * q + r], D[t]);
          }
        }
      }
      asm volatile("" ::: "memory");
    }
  }
  // x = (A_ij - D) L_jj^-T for the rows of tile row ti against panel j (nb columns): in the layout of
  // trsmStages (lane (q, n): row n of the wave's 16, columns 16 jj + 4 q + r), stored in place
  static __device__ __forceinline__ void solve(GP<T> A, GP<const T> dinv, int lda, int K, int ti, int j,
                                               int nb, const Acc (&D)[4], const Acc* raw, T* XB,
                                               Acc (&x)[4], bool store = true) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    TrsmOps<T> o;
    trsmLoadOps<T>((GP<const T>)A + (int64_t)kTile * j * lda + kTile * j, dinv, lda, nb, lane, o);
    // the accumulated update, from accumulator layout to row layout through XB
    ldsBarrier();
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) XB[(16 * w + Mfma<T>::row(lane, reg)) * kXbLd + 16 * t + n] = D[t][reg];
    }
    ldsBarrier();
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        x[jj][r] = (raw ? raw[jj][r] : T(0)) - XB[(16 * w + n) * kXbLd + 16 * jj + 4 * q + r];
      }
    }
    const int ri = kTile * ti + 16 * w + n;
    trsmMaskRows<T>(ri < K, nb, lane, x);
    trsmMaskOps<T>(nb, lane, o);
    trsmStages<T>(o, nb, x);
    if (store) trsmStoreRows<T>(A + (int64_t)min(ri, K - 1) * lda + kTile * j, ri < K, nb, lane, x);
  }
};

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void tailFactor(
    TailDesc td, DataRef<T> dref, T* dinvBase, unsigned* ctlBase, T* xchBase, unsigned* hostErr,
    long long spinLimit, long long* trace) {
  __shared__ T XB[kTile * kXbLd];
  __shared__ int sTicket;
  static_assert(4 * kPanelWidth * 4 + kPanelWidth * kInvLd + 4 * kPanelWidth * 4 <= kTile * kXbLd, "potrf buffers fit in XB");
  using Acc = typename Mfma<T>::Acc;
  using TT = TailTile<T>;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15;
  const int nP = td.nP, lda = td.lda, K = td.K;
  GP<unsigned> ctl = (GP<unsigned>)ctlBase + (size_t)blockIdx.y * td.ctlStride;
  if (tid == 0) sTicket = (int)(atomicAdd((unsigned*)ctl + 1, 1u) + 1u);  // armed to all ones
  __syncthreads();
  int k = sTicket, i = 0, j = 0;
  bool spine = true;
  if (k > 0) {  // column j: spine j + 1, then tiles (j + 1 .., j)
    k -= 1;
    for (;;) {
      const int cnt = nP - j;
      if (k < cnt) break;
      k -= cnt;
      j++;
    }
    spine = k == 0;
    i = spine ? j + 1 : j + k;
  }
  SweepWatch watch;
  watch.abortWord = ctl;
  watch.hostErr = hostErr;
  watch.limit = spinLimit;
  watch.reset();
  GP<T> A = pickData(dref) + td.diagOff;
  GP<T> dinvAll = (GP<T>)dinvBase + (size_t)blockIdx.y * nP * kDinvSlot;
  GP<unsigned> diagFlag = ctl + 2;
  GP<unsigned> xFlag = ctl + 2 + nP;  // [p * nP + i]: rows of tile row i of panel p are final
  GP<unsigned> taken2 = ctl + 2 + nP + nP * nP;  // [q]: spine q has its copy of the unsolved tile (q, q-2)
  GP<T> xch = (GP<T>)xchBase + (size_t)blockIdx.y * nP * kTile * kTile;  // slot q: X_q^{q-1}, per-lane order
  auto nbOf = [&](int p) { return min(kTile, K - kTile * p); };

  // (cooperative CU yield, hip_kernels.h: a tile that shares its CU with a spine pauses while the
  //  spine's panel Cholesky runs -- fp64 MFMA and fp64 VALU share one pipe)
  GP<unsigned> yieldWord = ctl + td.ctlStride - 1;
  if (!spine) {
    // ---- tile (i, j), i >= j + 1 (tile (j + 1, j) repeats what spine j + 1 solves for itself and
    // is the one that STORES it: the spine's critical path has no store and no release in it)
    const unsigned myCu = cuKey();
    Acc D[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, xi[4];
    T v[16];
    for (int p = 0; p < j; p++) {
      if (!tailWait(xFlag + p * nP + i, xFlag + p * nP + j, watch)) return;
      TT::fetch(A, lda, K, i, j, p, xi, v);
      TT::stageB(v, XB);
      if ((td.flags & 1) && __hip_atomic_load(yieldWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == myCu) {
        yieldWhile((const unsigned*)yieldWord, myCu);
      }
      TT::multiply(xi, XB, false, D);
    }
    Acc raw[4], x[4];
    const int ri = kTile * i + 16 * w + n, nb = nbOf(j);
    trsmLoadRows<T>((GP<const T>)A + (int64_t)min(ri, K - 1) * lda + kTile * j, nb, lane, raw);
    // (tiles (j + 1, j) and (j + 2, j): the spine of their row has taken its copy of the unsolved tile
    //  before it is overwritten)
    if (!tailWait(diagFlag + j, i == j + 1 ? xFlag + j * nP + j : (i == j + 2 ? taken2 + i : nullptr), watch)) return;
    TT::solve(A, dinvAll + (size_t)j * kDinvSlot, lda, K, i, j, nb, D, raw, XB, x);
    __syncthreads();  // every wave's stores issued
    if (tid == 0) tailRaise(xFlag + j * nP + i);
    return;
  }

  // ---- spine q: diagonal tile (q, q) and, for q >= 1, the tile left of it
  __builtin_amdgcn_s_setprio(3);
  const int q = i;  // (spine 0: i = j = 0)
  if (blockIdx.y) trace = nullptr;
  if (trace && tid == 0) trace[8 * q] = (long long)wall_clock64();
  Acc Dd[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  Acc Ds[4], D3[4], x[4];
  if (q >= 1) {
    // the unsolved tiles (q, q-1) and (q, q-2) FIRST: their tile roles overwrite them with the solved
    // rows later, and wait for this spine's word that its copies are taken
    TT::loadNeg(A, lda, K, q, q - 1, Ds);
    if (q >= 2) TT::loadNeg(A, lda, K, q, q - 2, D3);
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(xFlag + (q - 1) * nP + (q - 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (q >= 2) __hip_atomic_store(taken2 + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    Acc xi[4];
    T v[16];
    for (int p = 0; p + 2 < q; p++) {
      if (!tailWait(xFlag + p * nP + q, xFlag + p * nP + q - 1, watch, xFlag + p * nP + q - 2)) return;
      TT::fetch(A, lda, K, q, q - 2, p, xi, v);
      TT::stageB(v, XB);
      TT::multiply(xi, XB, false, D3);
      TT::fetchB(A, lda, K, q - 1, p, v);
      TT::stageB(v, XB);
      TT::multiply(xi, XB, false, Ds);
      TT::fetchB(A, lda, K, q, p, v);
      TT::stageB(v, XB);
      TT::multiply(xi, XB, true, Dd);
    }
    if (q >= 2) {
      // panel q-2: its own copy of X_q^{q-2}; X_{q-1}^{q-2} from spine q-1's exchange slot
      if (!tailWait(diagFlag + q - 2, nullptr, watch)) return;
      TT::solve(A, dinvAll + (size_t)(q - 2) * kDinvSlot, lda, K, q, q - 2, kTile, D3, nullptr, XB, x, /*store=*/false);
      {
        GP<const T> slot = xch + (size_t)(q - 1) * kTile * kTile + tid;  // (value e of thread t at [e][t]: coalesced)
        GP<const T> pp[16];
        bool need[16];
        T got[16];
#pragma unroll
        for (int e = 0; e < 16; e++) {
          pp[e] = slot + 256 * e;
          need[e] = true;
          got[e] = T(0);
        }
        if (!sweepWait<T, 16>(pp, need, got, watch)) return;
        const int qq = lane >> 4;
        ldsBarrier();
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
#pragma unroll
          for (int r = 0; r < 4; r++) XB[(16 * w + n) * kXbLd + 16 * jj + 4 * qq + r] = got[4 * jj + r];
        }
        ldsBarrier();
      }
      TT::multiply(x, XB, false, Ds);
      TT::stageX(x, XB);
      TT::multiply(x, XB, true, Dd);
    }
  }
  // the diagonal tile itself, in the accumulator layout of the panel Cholesky (identity beyond nb,
  // zero above the diagonal: potrfTilesBlocked's own load, done here BEFORE the wait for the previous
  // panel -- its loads inside the call are dead once `pre` overwrites the accumulators)
  Acc own[4];
  {
    GP<const T> Aq = (GP<const T>)A + (int64_t)kTile * q * lda + kTile * q;
    const int nbq = nbOf(q), li = lane & 15;
#pragma unroll
    for (int tj = 0; tj < 4; tj++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * w + Mfma<T>::row(lane, r), col = 16 * tj + li;
        const int rl = min(row, nbq - 1);
        const T val = Aq[(int64_t)rl * lda + min(col, rl)];
        own[tj][r] = (row < nbq && col <= row) ? val : ((row >= nbq && col == row) ? T(1) : T(0));
      }
    }
  }
  if (q >= 1) {
    if (!tailWait(diagFlag + q - 1, nullptr, watch)) return;
    if (trace && tid == 0) trace[8 * q + 2] = (long long)wall_clock64();  // diag(q-1) seen
    TT::solve(A, dinvAll + (size_t)(q - 1) * kDinvSlot, lda, K, q, q - 1, kTile, Ds, nullptr, XB, x, /*store=*/false);
    if (trace && tid == 0) trace[8 * q + 4] = (long long)wall_clock64();  // solved
    if (q + 1 < nP) {  // for spine q+1: plain write-through stores of self-validating words
      GP<T> slot = xch + (size_t)q * kTile * kTile + tid;
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
#pragma unroll
        for (int r = 0; r < 4; r++) sweepPublish<T>(slot + 256 * (4 * jj + r), x[jj][r]);
      }
    }
    if (trace && tid == 0) trace[8 * q + 5] = (long long)wall_clock64();  // published
    TT::stageX(x, XB);
    TT::multiply(x, XB, true, Dd);
    ldsBarrier();  // XB free for the potrf
  }
  {
    if (trace && tid == 0) trace[8 * q + 1] = (long long)wall_clock64();  // the panel Cholesky begins
    yieldPublish((unsigned*)yieldWord, cuKey());
    T(*blk)[4] = reinterpret_cast<T(*)[4]>(XB);
    T(*sol)[4] = blk + 3 * kPanelWidth;
    T* Ld = XB + 4 * kPanelWidth * 4;
    auto pre = [&](Acc* acc) {
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = t <= w ? own[t] - Dd[t] : own[t];
    };
    potrfPanelTiles<T>(A + (int64_t)kTile * q * lda + kTile * q, nbOf(q), lda, blk, sol,
                       XB + 4 * kPanelWidth * 4 + kPanelWidth * kInvLd, pre, Ld,
                       dinvAll + (size_t)q * kDinvSlot);
    yieldPublish((unsigned*)yieldWord, 0u);
    __syncthreads();
    if (tid == 0 && q + 1 < nP) tailRaise(diagFlag + q);
    if (trace && tid == 0) trace[8 * q + 3] = (long long)wall_clock64();  // diag(q) raised
  }
}

}  // namespace hipk
}  // namespace BaSpaCho
