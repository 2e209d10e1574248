// Hand-written HIP kernels of the MI355X (gfx950, CDNA4) numeric factor path.
// wave = 64 lanes; fp64 MFMA v_mfma_f64_16x16x4_f64 for the rank-nb update; LDS-staged
// panels; native fp64 atomics (global_atomic_add_f64) for concurrent scatter targets.
// Each kernel names the reference op it replaces (file:line in /root/reference/baspacho).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include <cstdint>

#include "hip_plan.h"

namespace BaSpaCho {
namespace hipk {

// One matrix (single) or a batch of identical-structure matrices (many, indexed by blockIdx.y);
// replaces the Plain/Batched policy structs of MatOpsCuda.cu:345-368.
#ifndef BSP_BULK_AUX
#define BSP_BULK_AUX 0  // cache policy of the bulk tile's operand loads (global_load_lds aux: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef BSP_TILE_PRIO
#define BSP_TILE_PRIO 2  // s_setprio of the chain launches' tile workgroups (the potrf / trsm ones: 3)
#endif

template <typename T>
struct DataRef {
  T* single;
  T* const* many;
};

// Pointers that come out of memory (the batch's pointer array) or out of a select are "generic" to
// the compiler, and every access through them becomes a FLAT instruction, which counts against the
// LDS counter as well: each wait for an LDS read then also waits for the global loads in flight
// (the prefetch of the next K chunk, the operands of the next pairs ...).  The round trip through
// the global address space tells the compiler where the data lives; accesses become global_*.
// (a cast to the global address space and back is folded away: the pointers have to KEEP the
// address-space type down to the access, hence GP<T> throughout the kernels)
template <typename T>
using GP = __attribute__((address_space(1))) T*;

template <typename T>
__device__ __forceinline__ GP<T> pickData(const DataRef<T>& d) {
  return (GP<T>)(d.many ? d.many[blockIdx.y] : d.single);
}
__device__ __forceinline__ void atomicSub(GP<double> p, double v) { unsafeAtomicAdd((double*)p, -v); }
__device__ __forceinline__ void atomicSub(GP<float> p, float v) { unsafeAtomicAdd((float*)p, -v); }

// COOPERATIVE CU YIELD.  fp64 MFMA and fp64 VALU instructions go through the same pipe of a SIMD,
// and an MFMA in flight is not pre-empted: while a bulk-update wave runs its K loop on a SIMD, every
// dependent fp64 instruction of the panel Cholesky's pivot chain waits for the MFMA ahead of it
// (in-situ trace, tools/trace_potrf.py: median step unchanged, 90th percentile 3.5x -- the chain
// workgroup's loop took 40 k clocks beside the bulk tiles against 28 k alone).  Neither stream
// priorities nor CU masks keep the bulk tiles off the chain's CU, so the chain workgroup says where
// it is: it publishes the identity of its CU in a word of device memory, and bulk waves that find
// themselves on that CU sleep (bounded) until the word changes.  Three bulk workgroups pause for
// the ~15 us of a potrf; the critical path gets its pipe back.
__device__ __forceinline__ unsigned cuKey() {
  // HW_REG_HW_ID (4): CU_ID[11:8] SH_ID[12] SE_ID[15:13]; HW_REG_XCC_ID (20): XCC_ID[3:0]
  const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
  return 0x80000000u | (xcc << 8) | hw;
}
__device__ __forceinline__ void yieldPublish(unsigned* flag, unsigned key) {
  if (flag && threadIdx.x == 0) {
    __hip_atomic_store((GP<unsigned>)flag, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ unsigned yieldPeek(const unsigned* flag) {
  return __hip_atomic_load((GP<const unsigned>)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bounded: at most ~48 x (sleep + one load) = a few tens of microseconds, whatever happens to the word
__device__ __forceinline__ void yieldWhile(const unsigned* flag, unsigned key) {
  for (int spin = 0; spin < 48; spin++) {
    __builtin_amdgcn_s_sleep(64);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)yieldPeek(flag)) != key) break;
  }
}

__device__ __forceinline__ void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory
// counter, i.e. it waits for every global store in flight (hundreds of cycles when a kernel
// stores to memory between barriers, as the panel Cholesky does with its finished columns)
__device__ __forceinline__ void ldsBarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 1/sqrt(x): hardware estimate (v_rsq_f64 / v_rsq_f32) refined by Newton steps to full precision.
// The library rsqrt() expands to a full-precision sqrt plus a division (hundreds of dependent
// cycles), which sits on the serial critical path of the panel Cholesky.
__device__ __forceinline__ double fastRsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const double r = fma(-0.5 * x * y, y, 0.5);  // 0.5 - 0.5 x y^2
    y = fma(y, r, y);
  }
  return y;
}
__device__ __forceinline__ float fastRsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  const float r = fmaf(-0.5f * x * y, y, 0.5f);
  return fmaf(y, r, y);
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
  GP<T> data = pickData(dref);
  const int64_t c0 = sk.chainColPtr[l];
  const int64_t nCh = sk.chainColPtr[l + 1] - c0;
  const int64_t diagCh = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
  GP<T> D = data + sk.chainData[c0];
  GP<T> B = data + sk.chainData[c0 + diagCh];
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
    GP<T> row = B + (int64_t)r * n;
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

// lumps of width <= 4 (the 3-wide point columns of bundle adjustment) are driven by ElimLumpDesc: one
// descriptor load, the n x n Cholesky redundantly in every lane's registers (hardware rsq + Newton,
// no wave synchronisation, no division), kTinyPerWave lumps per wave.
#ifndef BSP_TINY_PER_WAVE
#define BSP_TINY_PER_WAVE 4
#endif
// (round 6, BAL-871: 8 per wave in two passes of four, twice the bytes in flight per wave: 0.393 against
//  0.301-0.309 ms; profiles/r06_ab_elim_factor_tiny.txt)
constexpr int kTinyPerWave = BSP_TINY_PER_WAVE;
// K1s  the eliminated columns STAGED through LDS.  A wave's four lumps are neighbours in
// memory (an elimination range is laid out lump after lump, each column one dense (n + rows) x n
// block), so the wave reads its whole stretch -- typically ~600 values -- with fully coalesced
// 512-byte wave loads, all issued back to back, factors and solves out of LDS (16 lanes per lump,
// the n x n Cholesky redundantly in registers) and writes the stretch back the same way.
// (The direct form of rounds 1-2 read every 24-byte row with three 8-byte loads per lane and the
// diagonal block with six more per lane: ~18 wave instructions of ~270 useful bytes each per four
// lumps, 2.9 TB/s on the 527 480 point columns of BAL-871.)  A stretch that does not fit
// the scratch (heavy-tail points seen by dozens of cameras) takes the direct path lump by lump.
template <typename T>
using LP = __attribute__((address_space(3))) T*;
// (round 4, BAL-871, kernel time by cap: 640 0.316 ms, 768 0.310, 896 0.320, 1024 0.335, 1280 0.305,
//  1536 0.339, 2048 0.40 -- profiles/r04_ab_elim_factor_staged_cap.txt: fewer stretches fall back to
//  the direct path as the cap grows, fewer workgroups fit a CU)
constexpr int kStagedCap = 1280;  // scratch values per wave (10 KB fp64; 40 KB per workgroup)

// one lump of width n <= 4 by 16 lanes (sub = 0..15): Cholesky of the diagonal block at D, rows
// below it (at D + n * n) solved against it; P: global or LDS pointer
template <typename T, typename P>
__device__ __forceinline__ void tinyLumpBody(P D, int n, int rowsBelow, int sub) {
  constexpr int G = 16;
  P B = D + n * n;
  T a[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      const T v = D[min(i, n - 1) * n + min(j, n - 1)];
      a[i][j] = (i < n && j < n) ? v : (i == j ? T(1) : T(0));
    }
  }
  T inv[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    T d = a[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= a[j][k] * a[j][k];
    inv[j] = fastRsqrt(d);
    a[j][j] = d * inv[j];
#pragma unroll
    for (int i = j + 1; i < 4; i++) {
      T s = a[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= a[i][k] * a[j][k];
      a[i][j] = s * inv[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      if (i < n && sub == i * n + j) D[i * n + j] = a[i][j];
    }
  }
  for (int r = sub; r < rowsBelow; r += G) {
    P row = B + r * n;
    T y[4];
#pragma unroll
    for (int j = 0; j < 4; j++) y[j] = row[min(j, n - 1)];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      T s = y[j];
#pragma unroll
      for (int i = 0; i < j; i++) s -= y[i] * a[j][i];
      y[j] = s * inv[j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j < n) row[j] = y[j];
    }
  }
}

// Round 6: kTinyPerWave lumps per wave in PASSES of four (16 lanes per lump), every load of the whole
// stretch issued before the first LDS store, and the copy loops bounded by the stretch's own length
// (uniform) instead of 20 predicated iterations -- a predicated load is a branch around it.
template <typename T>
__global__ __launch_bounds__(256) void elimFactorTinyStaged(const ElimLumpDesc* descs,
                                                            DataRef<T> dref, int numLumps) {
  __shared__ T scratch[4][kStagedCap];
  constexpr int PASSES = kTinyPerWave / 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, sub = lane & 15, q = lane >> 4;
  const int first = (blockIdx.x * 4 + wave) * kTinyPerWave;
  if (first >= numLumps) return;
  const int last = min(first + kTinyPerWave - 1, numLumps - 1);
  ElimLumpDesc ld[PASSES];
  bool live[PASSES];
  bool narrow = true;
#pragma unroll
  for (int ps = 0; ps < PASSES; ps++) {
    live[ps] = first + 4 * ps + q <= last;
    ld[ps] = descs[live[ps] ? first + 4 * ps + q : last];
    narrow = narrow && ld[ps].n <= 4;
  }
  const ElimLumpDesc ld0 = descs[first], ldL = descs[last];  // (wave-uniform)
  const int64_t start = ld0.diagOff;
  const int64_t len = ldL.diagOff + (int64_t)ldL.n * (ldL.n + ldL.rowsBelow) - start;
  GP<T> data = pickData(dref);
  if (!narrow) return;  // (the caller checks the range's maximum width)
  // the lumps of a range follow one another in memory; anything else takes the direct path
  if (len <= 0 || len > kStagedCap) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
      if (live[ps]) tinyLumpBody<T>(data + ld[ps].diagOff, ld[ps].n, ld[ps].rowsBelow, sub);
    }
    return;
  }
  // (Round 4: 16 bytes per lane in and out -- half the wave loads and stores; the stretch starts at
  //  an 8-byte boundary, which global accesses tolerate -- measured SLOWER: 0.364 against 0.335 ms.)
  constexpr int IT = kStagedCap / 64;
  const int E = (int)len;
  const int nIt = __builtin_amdgcn_readfirstlane((E + 63) >> 6);
  GP<T> D0 = data + start;
  LP<T> sc = (LP<T>)scratch[wave];
  T v[IT];
#pragma unroll
  for (int it = 0; it < IT; it++) {
    if (it < nIt) v[it] = D0[min(it * 64 + lane, E - 1)];
  }
#pragma unroll
  for (int it = 0; it < IT; it++) {
    if (it < nIt) sc[it * 64 + lane] = v[it];  // (values beyond E: copies of the last one, never stored back)
  }
  waveSync();
#pragma unroll
  for (int ps = 0; ps < PASSES; ps++) {
    if (live[ps]) tinyLumpBody<T>(sc + (int)(ld[ps].diagOff - start), ld[ps].n, ld[ps].rowsBelow, sub);
  }
  waveSync();
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int e = it * 64 + lane;
    if (it < nIt && e < E) D0[e] = sc[e];
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
  GP<T> data = pickData(dref);
  const int n = (int)(sk.lumpStart[l + 1] - sk.lumpStart[l]);
  const int64_t si = sk.chainRowSpan[c];
  const int siSize = (int)(sk.spanStart[si + 1] - sk.spanStart[si]);
  GP<const T> Bi = data + sk.chainData[c];
  const int64_t t = sk.spanToLump[si];
  const int64_t tStride = sk.lumpStart[t + 1] - sk.lumpStart[t];
  const int64_t colOff = sk.spanOffsetInLump[si];
  const int64_t t0 = sk.chainColPtr[t], tCount = sk.chainColPtr[t + 1] - t0;

  int64_t lo = 0;  // target chains are visited in increasing order: resume the search
  for (int64_t j = c; j < cEnd; j++) {
    const int64_t sj = sk.chainRowSpan[j];
    const int sjSize = (int)(sk.spanStart[sj + 1] - sk.spanStart[sj]);
    GP<const T> Bj = data + sk.chainData[j];
    int64_t hi = tCount;
    while (hi - lo > 1) {
      int64_t mid = lo + (hi - lo) / 2;
      if (sk.chainRowSpan[t0 + mid] <= sj) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    GP<T> tgt = data + sk.chainData[t0 + lo] + colOff;
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
  // inverse of (q, reg) -> row: the m with row(16 q + *, reg) == m has 4 q + reg == colOfRow(m)
  static __device__ __forceinline__ int colOfRow(int m) { return 4 * (m & 3) + (m >> 2); }
};
template <>
struct Mfma<float> {
  using Acc = float4_t;
  static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
  static __device__ __forceinline__ int colOfRow(int m) { return m; }
};


// ------------------------------------------------------------------------------------------
// K2g  sparse-elimination update, GATHER form (atomic-free, deterministic).  One wave owns one
// target block (sj,si) and the list of source chain pairs that contribute to it (sorted by target
// at plan time): it accumulates  sum_l B_j(l) * B_i(l)^T  in registers (each lane up to 4 output
// elements) and subtracts the sum from the target once.  Pair offsets are fetched 64 at a time
// (one coalesced load per lane) and broadcast with readlane; the small source blocks are read
// straight from HBM/L2 (each is |s| x n contiguous values).  Replaces the per-pair atomics of
// sparse_elim_straight_kernel (MatOpsCuda.cu:235-331) for ranges whose blocks fit a wave.
// ------------------------------------------------------------------------------------------
constexpr int kGatherStage = 1024;  // staged values per wave (8 KB fp64)
constexpr int kGatherLoads = 16;    // global loads in flight per lane and batch

template <typename T>
__global__ __launch_bounds__(256) void elimGather(const ElimGatherItem* items, const uint32_t* offJ,
                                                  const uint32_t* offI, DataRef<T> dref,
                                                  int numItems) {
  // The source blocks of a batch of pairs are fetched with one coalesced load per 64 values
  // (a pair's two blocks are |sj| x n and |si| x n contiguous values), parked in this wave's LDS
  // slice and consumed from there; the next batch is already in flight in registers while the
  // current one is multiplied (the 8-byte per-lane operand fetches straight from L1 were
  // address-unit bound).
  __shared__ T stageAll[4][kGatherStage];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + wave;
  if (idx >= numItems) return;
  const ElimGatherItem it = items[idx];
  GP<T> data = pickData(dref);
  T* stage = stageAll[wave];
  const int rows = it.rows, cols = it.cols, n = it.n;
  const int total = rows * cols;
  const int EJ = rows * n, E = (rows + cols) * n;  // values per pair: B_j then B_i
  const int loadsPerPair = (E + 63) >> 6;
  const int batch = max(1, min(kGatherLoads / loadsPerPair, kGatherStage / E));
  // per-lane output elements e = lane + 64*s  ->  (r, q)
  int rowK[4], colK[4], tgt[4];
  bool ok[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int e = lane + 64 * s;
    ok[s] = e < total;
    const int ee = ok[s] ? e : 0;
    const int r = ee / cols, q = ee - r * cols;
    rowK[s] = r * n;
    colK[s] = EJ + q * n;
    tgt[s] = r * it.tgtStride + q;
    if ((it.flags & 2) && q > r) ok[s] = false;  // diagonal target block: lower triangle only
  }
  const int slots = (total + 63) >> 6;
  T acc[4] = {T(0), T(0), T(0), T(0)};

  for (int base = it.pairBegin; base < it.pairEnd; base += 64) {
    const int cnt = min(64, it.pairEnd - base);
    const uint32_t myJ = lane < cnt ? offJ[base + lane] : 0u;
    const uint32_t myI = lane < cnt ? offI[base + lane] : 0u;
    T v[kGatherLoads];
    auto fetch = [&](int t0) {  // issue the loads of pairs [t0, t0+batch) of this group of 64
#pragma unroll
      for (int u = 0; u < kGatherLoads; u++) {
        const int t = t0 + u / loadsPerPair;
        const int e = (u % loadsPerPair) * 64 + lane;
        const bool live = u < batch * loadsPerPair && t < cnt && e < E;
        const int tt = min(t, cnt - 1);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)myJ, tt);
        const uint32_t oi = (uint32_t)__builtin_amdgcn_readlane((int)myI, tt);
        const uint32_t off = e < EJ ? oj + e : oi + (e - EJ);
        v[u] = live ? data[off] : T(0);
      }
    };
    auto park = [&]() {
#pragma unroll
      for (int u = 0; u < kGatherLoads; u++) {
        const int e = (u % loadsPerPair) * 64 + lane;
        if (u < batch * loadsPerPair && e < E) stage[(u / loadsPerPair) * E + e] = v[u];
      }
    };
    fetch(0);
    park();
    waveSync();
    for (int t0 = 0; t0 < cnt; t0 += batch) {
      const bool more = t0 + batch < cnt;
      if (more) fetch(t0 + batch);
      const int nb = min(batch, cnt - t0);
      for (int t = 0; t < nb; t++) {
        const T* blk = stage + t * E;
        if (n == 3) {
#pragma unroll
          for (int sl = 0; sl < 2; sl++) {
            const T* a = blk + rowK[sl];
            const T* b = blk + colK[sl];
            acc[sl] += a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
          }
          if (slots > 2) {
#pragma unroll
            for (int sl = 2; sl < 4; sl++) {
              const T* a = blk + rowK[sl];
              const T* b = blk + colK[sl];
              acc[sl] += a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
            }
          }
        } else {
#pragma unroll
          for (int sl = 0; sl < 4; sl++) {
            if (sl < slots) {
              const T* a = blk + rowK[sl];
              const T* b = blk + colK[sl];
              T d = T(0);
              for (int k = 0; k < n; k++) d += a[k] * b[k];
              acc[sl] += d;
            }
          }
        }
      }
      waveSync();
      if (more) {
        park();
        waveSync();
      }
    }
  }
  GP<T> target = data + it.tgtOff;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    if (ok[s]) {
      if (it.flags & 1) {
        atomicSub(target + tgt[s], acc[s]);
      } else {
        target[tgt[s]] -= acc[s];
      }
    }
  }
}

// K2m  the gather on the matrix cores, for target blocks of at most 16 x 16 (the 9x9 camera blocks
// of bundle adjustment).  A pair's product B_j * B_i^T is ONE v_mfma 16x16x4 per 4 source columns:
// lane (i, k) = (lane & 15, lane >> 4) supplies B_j[i][k] and B_i[i][k], i.e. every lane issues one
// 8-byte load per operand, the 64 lanes together read the two contiguous source blocks, and nothing
// goes through LDS (K2g spends 13 LDS wave-instructions per pair and is LDS-bandwidth bound at
// ~12 G pairs/s).  The sum over the pairs stays in the 4 accumulator registers; loads of U pairs
// are issued before their MFMAs.
template <typename T>
__global__ __launch_bounds__(256) void elimGatherMfma(const ElimGatherItem* items,
                                                      const uint32_t* offJ, const uint32_t* offI,
                                                      DataRef<T> dref, int numItems) {
  // pairs whose operand loads are in flight together.  With items of at most 128 pairs the kernel is
  // not bound by a wave's own round trips any more: 4 / 8 / 16 give 6.87-6.92 / 6.95-6.99 / 7.03 ms on
  // BAL-871 (fewer registers, more waves ready to issue)
  constexpr int U = 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const int idx = blockIdx.x * 4 + wave;
  if (idx >= numItems) return;
  const ElimGatherItem it = items[idx];
  GP<T> data = pickData(dref);
  GP<const T> src = (GP<const T>)data;
  const int rows = it.rows, cols = it.cols, n = it.n;
  using Acc = typename Mfma<T>::Acc;
  Acc acc = {0, 0, 0, 0};
  // ONE load instruction per pair (round 3).  PMC on BAL-871 (profiles/r03_pmc_ta.txt): the texture
  // addresser is busy 91-100 % of this kernel, ~20 cycles per wave load whatever the 27 active lanes
  // fetch, while the L1 -> L2 read latency averages 460 cycles: the kernel is bound by the NUMBER of
  // vector-memory instructions, not by bytes or latency.  When both blocks of a pair fit half a wave
  // (rows * n, cols * n <= 32 and n <= 4: the 9x3 blocks of bundle adjustment), lanes 0-31 fetch B_j
  // and lanes 32-63 fetch B_i with the same instruction -- element e of a block in lane e -- and the
  // MFMA operand layout (lane (i, k) wants element i * n + k) is restored through the LDS crossbar
  // (ds_bpermute: no LDS memory, four 32-bit permutes per pair).
  const int EA = rows * n, EB = cols * n;
  // (two pairs per load instruction -- 16 bytes per lane through a private LDS slot -- was measured at
  //  1.15 against 1.17 ms for the kernel and nothing for factor(): round 3, removed in round 4)
  if (n <= 4 && EA <= 32 && EB <= 32) {
    const int half = lane >> 5, e = lane & 31;
    const bool ldOk = half ? e < EB : e < EA;
    const uint32_t eOff = ldOk ? (uint32_t)e : 0u;
    const bool okA = li < rows && lk < n, okB = li < cols && lk < n;
    const int selA = 4 * (okA ? li * n + lk : 0), selB = 4 * (32 + (okB ? li * n + lk : 0));
    for (int base = it.pairBegin; base < it.pairEnd; base += 64) {
      const int cnt = min(64, it.pairEnd - base);
      const uint32_t myJ = lane < cnt ? offJ[base + lane] : 0u;
      const uint32_t myI = lane < cnt ? offI[base + lane] : 0u;
      for (int t0 = 0; t0 < cnt; t0 += U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int t = min(t0 + u, cnt - 1);
          const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)myJ, t);
          const uint32_t oi = (uint32_t)__builtin_amdgcn_readlane((int)myI, t);
          v[u] = src[(half ? oi : oj) + eOff];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const bool live = t0 + u < cnt;
          T a, b;
          if constexpr (sizeof(T) == 8) {
            const int lo = __double2loint(v[u]), hi = __double2hiint(v[u]);
            a = __hiloint2double(__builtin_amdgcn_ds_bpermute(selA, hi), __builtin_amdgcn_ds_bpermute(selA, lo));
            b = __hiloint2double(__builtin_amdgcn_ds_bpermute(selB, hi), __builtin_amdgcn_ds_bpermute(selB, lo));
          } else {
            const int w = __float_as_int(v[u]);
            a = __int_as_float(__builtin_amdgcn_ds_bpermute(selA, w));
            b = __int_as_float(__builtin_amdgcn_ds_bpermute(selB, w));
          }
          acc = Mfma<T>::run((okA && live) ? a : T(0), okB ? b : T(0), acc);
        }
      }
    }
  } else {
  for (int k0 = 0; k0 < n; k0 += 4) {  // one pass per 4 source columns (n <= 4: a single pass)
    const int k = k0 + lk;
    const bool okA = li < rows && k < n, okB = li < cols && k < n;
    const uint32_t eA = okA ? (uint32_t)(li * n + k) : 0u, eB = okB ? (uint32_t)(li * n + k) : 0u;
    for (int base = it.pairBegin; base < it.pairEnd; base += 64) {
      const int cnt = min(64, it.pairEnd - base);
      const uint32_t myJ = lane < cnt ? offJ[base + lane] : 0u;
      const uint32_t myI = lane < cnt ? offI[base + lane] : 0u;
      for (int t0 = 0; t0 < cnt; t0 += U) {
        T a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int t = min(t0 + u, cnt - 1);
          const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)myJ, t);
          const uint32_t oi = (uint32_t)__builtin_amdgcn_readlane((int)myI, t);
          a[u] = src[oj + eA];
          b[u] = src[oi + eB];
        }
        // (no branch per pair: pairs past the end of the list multiply zeros)
#pragma unroll
        for (int u = 0; u < U; u++) {
          const bool live = t0 + u < cnt;
          acc = Mfma<T>::run((okA && live) ? a[u] : T(0), okB ? b[u] : T(0), acc);
        }
      }
    }
  }
  }
  GP<T> target = data + it.tgtOff;
  GP<T> ptr[4];
  bool ok[4];
  T old[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int r = Mfma<T>::row(lane, g);
    ok[g] = r < rows && li < cols && !((it.flags & 2) && li > r);
    ptr[g] = target + (ok[g] ? (int64_t)r * it.tgtStride + li : 0);
  }
  if (it.flags & 1) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (ok[g]) atomicSub(ptr[g], acc[g]);
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; g++) old[g] = *ptr[g];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (ok[g]) *ptr[g] = old[g] - acc[g];
    }
  }
}

// K2t  the same gather for TINY target blocks (<= 16 elements, e.g. the 3x3 blocks of automatically
// detected elimination ranges), operands straight from global memory (a pair's blocks are a few
// dozen bytes).  Such items hold 1.4 pairs on average (GRID 82x82) and the kernel is bound by the
// number of scattered vector-memory instructions a CU can address, not by bytes (PMC: 22 loads
// per wave, 1.8 TB/s), so everything is arranged to need few of them:
//  * G lanes per item, G = 9 (everything about the item fits 9 lanes: 7 items per wave) or 16 (4);
//  * the descriptor is read one dword per lane and handed round through the LDS crossbar;
//  * lane e fetches element e of B_j and of B_i (one contiguous block per group and operand), the
//    rows a lane needs come from its neighbours' registers;
//  * the first pair's offsets ride in the descriptor, those of further pairs are read G at a time;
//  * the target is read while the blocks are on their way.
static_assert(sizeof(ElimGatherItem) == 40 && offsetof(ElimGatherItem, pairBegin) == 8 &&
                  offsetof(ElimGatherItem, tgtStride) == 16 && offsetof(ElimGatherItem, rows) == 20 &&
                  offsetof(ElimGatherItem, n) == 24 && offsetof(ElimGatherItem, flags) == 26 &&
                  offsetof(ElimGatherItem, firstJ) == 28 && offsetof(ElimGatherItem, firstI) == 32,
              "elimGatherTiny reads the descriptor as 9 dwords");
template <typename T, int G>
__global__ __launch_bounds__(256) void elimGatherTiny(const ElimGatherItem* items,
                                                      const uint32_t* offJ, const uint32_t* offI,
                                                      DataRef<T> dref, int numItems) {
  constexpr int IPW = 64 / G;
  const int lane = threadIdx.x & 63, grp = lane / G, sub = lane - grp * G, gbase = grp * G;
  const int idx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * IPW + grp;
  if (grp >= IPW || idx >= numItems) return;  // whole groups leave; all exchanges are in-group
  const uint32_t myW = reinterpret_cast<const uint32_t*>(items + idx)[min(sub, 8)];
  auto field = [&](int k) { return (uint32_t)__shfl((int)myW, gbase + k, 64); };
  const int64_t tgtOff = (int64_t)((uint64_t)field(0) | ((uint64_t)field(1) << 32));
  const int pairBegin = (int)field(2), pairEnd = (int)field(3), tgtStride = (int)field(4);
  const uint32_t rc = field(5), nf = field(6);
  const int rows = int(rc & 0xffffu), cols = int(rc >> 16), n = int(nf & 0xffffu);
  const int flags = int(nf >> 16);
  GP<T> data = pickData(dref);
  const int total = rows * cols;
  const bool live = sub < total;
  const int e = live ? sub : 0;
  const int r = e / cols, q = e - r * cols;
  const bool writes = live && !((flags & 2) && q > r);
  const bool atomic = flags & 1;
  GP<T> target = data + tgtOff + (int64_t)r * tgtStride + q;
  const T old = (writes && !atomic) ? *target : T(0);
  const int rn = r * n, qn = q * n;
  // (G = 9: guaranteed by the plan; G = 16: sources wider than 16 / rows take the row loads)
  const bool small = G == 9 || __all(rows * n <= 16 && cols * n <= 16);
  const int ej = min(sub, rows * n - 1), ei = min(sub, cols * n - 1);
  auto dot = [&](uint32_t oj, uint32_t oi) -> T {
    T d = T(0);
    if (small) {  // wave-uniform
      const T vj = data[oj + ej];
      const T vi = data[oi + ei];
      if (n <= 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int kk = min(k, n - 1);
          const T a = __shfl(vj, gbase + rn + kk, 64);
          const T b = __shfl(vi, gbase + qn + kk, 64);
          d += k < n ? a * b : T(0);
        }
      } else {
        for (int k = 0; k < n; k++) d += __shfl(vj, gbase + rn + k, 64) * __shfl(vi, gbase + qn + k, 64);
      }
      return d;
    }
    GP<const T> Bj = data + oj + rn;
    GP<const T> Bi = data + oi + qn;
    for (int k = 0; k < n; k++) d += Bj[k] * Bi[k];
    return d;
  };
  T acc = dot(field(7), field(8));
  const int more = pairEnd - pairBegin - 1;
  for (int base = 0; base < more; base += G) {
    const int cnt = min(G, more - base);
    const int pi = pairBegin + 1 + base + min(sub, cnt - 1);
    const int myJ = (int)offJ[pi], myI = (int)offI[pi];
    for (int t = 0; t < cnt; t += 2) {
      const int t1 = min(t + 1, cnt - 1);
      const T d0 = dot((uint32_t)__shfl(myJ, gbase + t, 64), (uint32_t)__shfl(myI, gbase + t, 64));
      const T d1 = dot((uint32_t)__shfl(myJ, gbase + t1, 64), (uint32_t)__shfl(myI, gbase + t1, 64));
      acc += d0;
      if (t + 1 < cnt) acc += d1;
    }
  }
  if (writes) {
    if (atomic) {
      atomicSub(target, acc);
    } else {
      *target = old - acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K3  panel potrf: in-place Cholesky of the nb x nb (nb <= 64) diagonal block of a panel, one
// workgroup per panel, whole block in LDS.  Replaces cusolverDn?potrf / potrfBatched
// (MatOpsCuda.cu:508-548, 727-755) on the panel granularity.
// ------------------------------------------------------------------------------------------
#if defined(BSP_KTRACE) || defined(BSP_TRACE_UPD)
// in-situ trace builds (build.sh with BSP_KTRACE=1, or BSP_EXTRA_DEFS=-DBSP_TRACE_UPD for the
// updateTile trace alone): every launch of a stamped kernel appends one record of clock values;
// read back with hipBackendReadTrace()
constexpr int kTraceW = 8;  // clock values per record (slots 0-3: kernel phases, 4-7: inside a potrf step)
__device__ long long bspTrace[8192 * kTraceW];
__device__ unsigned bspTraceCount;
__shared__ unsigned bspTraceSlot;
#endif
#if defined(BSP_KTRACE)
#ifdef BSP_TRACE_TILE
#define BSP_STEP_STAMPS(slot) ((slot) < 4)
#define BSP_CLOCK() wall_clock64()  // device-wide constant-rate counter (the shader clocks of
                                    // different XCDs are not aligned): 100 MHz
#else
#define BSP_STEP_STAMPS(slot) true
#define BSP_CLOCK() clock64()
#endif
#ifndef BSP_TRACE_TID
#define BSP_TRACE_TID 0  // thread that writes the in-step stamps (slots 4-7): 0, 64, 128, 192 = wave 0..3
#endif
#define BSP_STAMP(slot)                                                        \
  if (threadIdx.x == ((slot) >= 4 ? BSP_TRACE_TID : 0) && blockIdx.x == 0 && BSP_STEP_STAMPS(slot)) { \
    if ((slot) == 0) bspTraceSlot = atomicAdd(&bspTraceCount, 1u) & 8191u;     \
    bspTrace[(bspTraceSlot & 8191u) * kTraceW + (slot)] = BSP_CLOCK();         \
  }
// BSP_TRACE_TILE builds: per chain-step launch (ordinal passed by the host), the earliest / latest
// start and the latest end over ALL its workgroups (wall clock), besides workgroup 0's record
__device__ unsigned long long bspLaunchExtent[2048 * 4];  // start min, start max, end max, wg0 end
#define BSP_EXTENT_BEGIN(id)                                                                   \
  if (threadIdx.x == 0) {                                                                      \
    const unsigned long long t_ = wall_clock64();                                              \
    atomicMin(&bspLaunchExtent[((id) & 2047) * 4 + 0], t_);                                     \
    atomicMax(&bspLaunchExtent[((id) & 2047) * 4 + 1], t_);                                     \
  }
#define BSP_EXTENT_END(id, wg0)                                                                \
  if (threadIdx.x == 0) {                                                                      \
    const unsigned long long t_ = wall_clock64();                                              \
    atomicMax(&bspLaunchExtent[((id) & 2047) * 4 + 2], t_);                                     \
    if (wg0) bspLaunchExtent[((id) & 2047) * 4 + 3] = t_;                                       \
  }
// BSP_TRACE_TILE builds: the LAST tile workgroup of a chain-step launch writes a record of its own
// (slots 4-7: start / after solve / after multiply / end; slots 0-3 stay zero), and the in-step
// stamps of workgroup 0 are off
#define BSP_STAMP_TILE(slot)                                                   \
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.x != 0) {    \
    if ((slot) == 4) bspTraceSlot = atomicAdd(&bspTraceCount, 1u) & 8191u;     \
    bspTrace[bspTraceSlot * kTraceW + (slot)] = BSP_CLOCK();                   \
  }
#elif defined(BSP_KDEBUG)
#define BSP_STAMP(slot) if (threadIdx.x == 0 && blockIdx.x == 0) bspDebugStamps[slot] = clock64()
__device__ long long bspDebugStamps[16];
#else
#define BSP_STAMP(slot)
#endif
#if !defined(BSP_KTRACE) || !defined(BSP_TRACE_TILE)
#undef BSP_STAMP_TILE
#define BSP_STAMP_TILE(slot)
#undef BSP_EXTENT_BEGIN
#undef BSP_EXTENT_END
#define BSP_EXTENT_BEGIN(id)
#define BSP_EXTENT_END(id, wg0)
#endif

// NT = tiles per dimension (NT = 4: 64x64 block, 256 threads).  A 256x256 single-workgroup variant
// (NT = 16 / paired tile rows, 512 threads) and a fused 256-column trsm were built and measured:
// 184 us + 77 us per outer block against ~165 us for the four 64-wide panel steps they would
// replace -- the serial dependent-latency per 4-column step dominates either way -- so the
// 64-wide panel chain stays.
// f(integral_constant<int, I>) for I = BEGIN .. END-1, everything inlined
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void staticFor(F& f) {
  if constexpr (BEGIN < END) {
    f(std::integral_constant<int, BEGIN>{});
    staticFor<BEGIN + 1, END>(f);
  }
}

struct NoPreUpdate {
  template <typename A>
  __device__ __forceinline__ void operator()(A*) const {}
};

// `pre(acc)` runs between the load of the block and the factorization: the fused kernel
// (updateTileDirectPotrf) applies the pending rank-K update of the block there.
// blk: 3 x (16 NT) rows of 4 values (ring of column blocks), sol: 16 NT rows of 4 values.
// Ld / dinvOut (both or neither): the four 16x16 diagonal blocks of L are also collected in LDS
// (Ld: 16 NT rows of kInvLd values, row i = its own block's 16 columns) and, after the last step,
// wave w inverts block w and writes it to dinvOut[w][16][16] -- the trsm of this panel then
// multiplies by the inverses (trsmStages) instead of substituting.
constexpr int kInvLd = 17;
constexpr int kDinvSlot = 4 * 16 * 16;        // one panel's inverted diagonal blocks
constexpr int kDinvBatchStride = 2 * kDinvSlot;  // two slots (alternating panels) per matrix
template <typename T>
__device__ __forceinline__ T readLaneT(T v, int srcLane);
template <>
__device__ __forceinline__ double readLaneT<double>(double v, int srcLane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
  return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float readLaneT<float>(float v, int srcLane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srcLane));
}
// lane n (= lane % 16, every group of 16 lanes does the same work) solves L y = e_n for the 16x16
// lower-triangular block at Lb (row stride ldl), right-looking, in registers: y = column n of L^-1
template <typename T>
__device__ __forceinline__ void invertColumn16(const T* Lb, int ldl, int n, T (&y)[16]) {
  const T rd = T(1) / Lb[n * ldl + n];
#pragma unroll
  for (int i = 0; i < 16; i++) y[i] = (i == n) ? T(1) : T(0);
#pragma unroll
  for (int i = 0; i < 16; i++) {
    y[i] *= readLaneT(rd, i);
#pragma unroll
    for (int k = i + 1; k < 16; k++) y[k] -= Lb[k * ldl + i] * y[i];
  }
}

// BLOCKED form of the panel potrf (round 3; the 4-column-step form of rounds 1-2 is in the history).
// 16-column blocks, two LDS buffers of 64 rows x 16 columns that alternate between blocks.  Per block J:
//   * ONE wave factors the whole block column in registers, lane = row (diagonal tile and the rows
//     below alike): right-looking, pivot and multipliers broadcast with v_readlane -- the Cholesky
//     of the diagonal tile and the solve of the rows below are the same 16 column steps, nothing
//     waits for another wave inside them; the result goes back to the buffer.  MEANWHILE the other
//     waves give the tiles right of the block column the PREVIOUS block's rank-16 update (four
//     v_mfma 16x16x4 per tile, operands from the other buffer);                   -- barrier --
//   * thread (row, 4-column group) stores the final entries of L; every wave applies this block's
//     update to its tile of the NEXT block column only and copies it out of the accumulators
//     into the other buffer.                                                       -- barrier --
// 8 barriers per panel instead of 32, no redundant pivot factorizations, the matrix cores work
// behind the serial path.  In situ (tools/trace_potrf.py): 26.9 k -> see DESIGN.md clocks per panel.
// Contract: accumulator-layout input after `pre`, identity padding beyond nb, Ld / dinvOut.  Buffer layout: row R, column c at R * 16 + (c ^ key(R)), key = 2 ((R / 2) % 8) --
// the lane-per-row walk and the MFMA operand fetch are both (nearly) conflict-free without padding.
__device__ __forceinline__ int potrfColbufAt(int R, int c) { return R * 16 + (c ^ (((R >> 1) & 7) << 1)); }

// One wave factors a block column of 16 columns held one row per lane (rows R of buf, lanes beyond
// the last row idle along on a copy of it): the Cholesky of the diagonal tile and the solve of the
// rows below are the same right-looking column steps.  Per column j the serial path is: pivot
// (v_readlane) -> 1/sqrt (hardware estimate + Newton) -> scale -> update of column j + 1.  The
// instruction order is pinned so that everything else fills its latencies: the multipliers of
// columns j + 3 .. 15 come back from LDS as broadcast reads of the column just written (one LDS
// instruction instead of two v_readlane on the vector pipe), and their updates are issued between
// the Newton steps of the NEXT column.
template <typename T>
struct RsqrtChain;
template <>
struct RsqrtChain<double> {
  static constexpr int kOps = 6;
};
template <>
struct RsqrtChain<float> {
  static constexpr int kOps = 3;
};
__device__ __forceinline__ double rsqEstimate(double x) { return __builtin_amdgcn_rsq(x); }
__device__ __forceinline__ float rsqEstimate(float x) { return __builtin_amdgcn_rsqf(x); }

// (orders two values' definitions for the compiler: whatever is computed from `b` afterwards cannot be
//  moved in front of the computation of `a` -- the DAG linearisation ignores sched_barrier for
//  plain arithmetic)
template <typename T>
__device__ __forceinline__ void orderAfter(const T& a, T& b) {
  asm volatile("" : "+v"(b) : "v"(a));  // (`a` is only read: its own consumers do not wait for `b`)
}

template <typename T, int J>
__device__ __forceinline__ void potrfBlockColumn(T* buf, int R) {
  T a[16];
#pragma unroll
  for (int c = 0; c < 16; c++) a[c] = buf[potrfColbufAt(R, c)];
  // multipliers that came back from LDS: column j's are requested at the end of step j and used
  // between the Newton steps of column j + 2 (a full step of latency budget); columns j + 1 and
  // j + 2 get column j's update straight away, through v_readlane.
  // (A variant with the pivots as a recurrence on uniform values, d_{j+1} = p - u^2 / d_j, and all
  //  vector work as filler one column behind was measured slower: 18.3 k against 16.9 k clocks per
  //  panel -- three columns of v_readlane updates per step instead of two.)
  T m[3][16];
  auto column = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int src = j >= 2 ? j - 2 : 0;      // column whose LDS-fed updates are applied now
    constexpr int nDef = j >= 2 ? 15 - (src + 3) + 1 : 0;  // columns src + 3 .. 15
    constexpr int nOps = RsqrtChain<T>::kOps;
    constexpr int perOp = nDef > 0 ? (nDef + nOps - 1) / nOps : 0;
    const T d = readLaneT(a[j], j);  // pivot (16 J + j, 16 J + j): lane j
    T y = rsqEstimate(d);
    const T h = T(-0.5) * d;
    T t = T(0);
#pragma unroll
    for (int op = 0; op < nOps; op++) {
      switch (op % 3) {
        case 0: t = h * y; break;
        case 1: t = fma(t, y, T(0.5)); break;
        default: y = fma(y, t, y); break;
      }
#pragma unroll
      for (int q = 0; q < 16; q++) {
        if (q >= op * perOp && q < (op + 1) * perOp && q < nDef) {
          const int k = src + 3 + q;
          orderAfter(op % 3 == 2 ? y : t, a[k]);
          a[k] -= a[src] * m[src % 3][k];
        }
      }
    }
    a[j] *= y;  // lane j: the diagonal entry; lanes > j: the multipliers; lanes < j: unused
    buf[potrfColbufAt(R, j)] = a[j];  // (every lane: clamped lanes rewrite the last row's value)
    if constexpr (j + 1 < 16) a[j + 1] -= a[j] * readLaneT(a[j], j + 1);
    if constexpr (j + 2 < 16) {
      orderAfter(a[j + 1], a[j + 2]);
      a[j + 2] -= a[j] * readLaneT(a[j], j + 2);
    }
#pragma unroll
    for (int k = j + 3; k < 16; k++) m[j % 3][k] = buf[potrfColbufAt(16 * J + k, j)];
  };
  staticFor<0, 16>(column);
}

// GLOBAL = false (trsmPanelPotrf): nothing is written to memory; the factor goes to `Lfull` (LDS, row
// stride ldFull) only.
template <typename T, typename Pre = NoPreUpdate, bool GLOBAL = true>
__device__ __forceinline__ void potrfTilesBlocked(GP<T> A, int nb, int lda, T* colbuf0, T* colbuf1,
                                                  Pre pre = Pre(), T* Ld = nullptr,
                                                  GP<T> dinvOut = nullptr, T* Lfull = nullptr,
                                                  int ldFull = 0) {
  constexpr int NT = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int i = tid >> 2, g = tid & 3;
  using Acc = typename Mfma<T>::Acc;
  Acc acc[NT];
#pragma unroll
  for (int tj = 0; tj < NT; tj++) {
    if (tj <= w) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * w + Mfma<T>::row(lane, r), col = 16 * tj + li;
        const int rl = min(row, nb - 1);
        const T v = A[(int64_t)rl * lda + min(col, rl)];
        acc[tj][r] = (row < nb && col <= row) ? v : ((row >= nb && col == row) ? T(1) : T(0));
      }
    } else {
      acc[tj] = Acc{0, 0, 0, 0};
    }
  }
  pre(acc);
  BSP_STAMP(1);
  if (Ld) {  // identity beyond nb, zero above the diagonal; the blocks fill in the rest
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int col = 4 * g + c;
      Ld[i * kInvLd + col] = (i >= nb && (i & 15) == col) ? T(1) : T(0);
    }
  }
  // tile (w, TJ) of this wave -> buf
  auto publish = [&](auto TJc, T* buf) __attribute__((always_inline)) {
    constexpr int TJ = decltype(TJc)::value;
#pragma unroll
    for (int r = 0; r < 4; r++) buf[potrfColbufAt(16 * w + Mfma<T>::row(lane, r), li)] = acc[TJ][r];
  };
  // acc[c] -= X_w X_c^T for the tiles c = C0 .. NT-1 of this wave (c <= w), X = solved block column in buf
  auto update = [&](int c0, int c1, const T* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
      const T xa = -buf[potrfColbufAt(16 * w + li, k0 + lk)];
#pragma unroll
      for (int tj = 0; tj < NT; tj++) {
        if (tj >= c0 && tj <= c1 && tj <= w) {
          acc[tj] = Mfma<T>::run(xa, buf[potrfColbufAt(16 * tj + li, k0 + lk)], acc[tj]);
        }
      }
    }
  };
  publish(std::integral_constant<int, 0>{}, colbuf0);
  ldsBarrier();
  auto block = [&](auto Jc) __attribute__((always_inline)) {
    constexpr int J = decltype(Jc)::value;
    if (16 * J >= nb) return;  // (uniform: the rest is identity padding)
    T* cur = (J & 1) ? colbuf1 : colbuf0;
    T* other = (J & 1) ? colbuf0 : colbuf1;
    if (J == 1) { BSP_STAMP(4); }
    auto storePrev = [&](int item) __attribute__((always_inline)) {  // block J - 1, from `other`
      const int row = 16 * (J - 1) + (item >> 2), g4 = item & 3;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int col = 16 * (J - 1) + 4 * g4 + c;
        if (row < nb && col < nb && col <= row) {
          const T v = other[potrfColbufAt(row, 4 * g4 + c)];
          if (GLOBAL) A[(int64_t)row * lda + col] = v;
          if (Lfull) Lfull[row * ldFull + col] = v;
          if (Ld && (row >> 4) == J - 1) Ld[row * kInvLd + 4 * g4 + c] = v;
        }
      }
    };
    if (w == J) {
      // wave J factors rows 16 J .. 63 of the block column, lane = row
      const int R = min(16 * J + lane, 16 * NT - 1);
      potrfBlockColumn<T, J>(cur, R);
    } else if (J >= 1) {
      // meanwhile: the previous block's update of the tiles right of the block column, and its
      // final entries to memory (three waves, (64 - 16 (J - 1)) x 4 items)
      if (w > J) update(J + 1, NT - 1, other);
      const int idx = 64 * (w < J ? w : w - 1) + lane;
      for (int item = idx; item < 4 * (16 * NT - 16 * (J - 1)); item += 192) storePrev(item);
    }
    if (J == 1) { BSP_STAMP(5); }
    ldsBarrier();
    if (J == 1) { BSP_STAMP(6); }
    // Final entries to memory (and to Ld): thread (row, 4-column group).  Off the serial path: when
    // a next block follows, the waves that do not factor it store while it is factored.
    auto storeBlock = [&](int item) __attribute__((always_inline)) {
      const int row = 16 * J + (item >> 2), g4 = item & 3;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int col = 16 * J + 4 * g4 + c;
        if (row < nb && col < nb && col <= row) {
          const T v = cur[potrfColbufAt(row, 4 * g4 + c)];
          if (GLOBAL) A[(int64_t)row * lda + col] = v;
          if (Lfull) Lfull[row * ldFull + col] = v;
          if (Ld && (row >> 4) == J) Ld[row * kInvLd + 4 * g4 + c] = v;
        }
      }
    };
    const bool nextBlock = J + 1 < NT && 16 * (J + 1) < nb;
    if (!nextBlock) {
      if (i >= 16 * J) storeBlock(tid - 64 * J);
    }
    if constexpr (J + 1 < NT) {
      if (16 * (J + 1) < nb) {
        // the next block column gets this block's update and is published; the tiles right of it
        // wait until the next block's factorization runs
        if (w > J) {
          update(J + 1, J + 1, cur);
          publish(std::integral_constant<int, J + 1>{}, other);
        }
        ldsBarrier();
      }
    }
    if (J == 1) { BSP_STAMP(7); }
  };
  staticFor<0, NT>(block);
  BSP_STAMP(2);
  if (Ld) {
    ldsBarrier();
    T y[16];
    invertColumn16(Ld + 16 * w * kInvLd, kInvLd, li, y);
    if (lane < 16) {
#pragma unroll
      for (int r = 0; r < 16; r++) dinvOut[(16 * w + r) * 16 + li] = y[r];
    }
  }
}

// the panel potrf the kernels call: blk .. sol are 4 x 64 x 4 contiguous values, buf2 another 1024
template <typename T, typename Pre = NoPreUpdate>
__device__ __forceinline__ void potrfPanelTiles(GP<T> A, int nb, int lda, T (*blk)[4], T (*sol)[4],
                                                T* buf2, Pre pre = Pre(), T* Ld = nullptr,
                                                GP<T> dinvOut = nullptr) {
  (void)sol;
  potrfTilesBlocked<T, Pre>(A, nb, lda, &blk[0][0], buf2, pre, Ld, dinvOut);
}

template <typename T>
__global__ __launch_bounds__(256) void potrfPanel(const PanelDesc* levelPanelDescs,
                                                  DataRef<T> dref) {
  __shared__ T blk[8 * kPanelWidth][4];  // (ring of three column blocks + sol / the blocked form's two buffers)
  T(*sol)[4] = blk + 3 * kPanelWidth;
  BSP_STAMP(0);
  const PanelDesc pd = levelPanelDescs[blockIdx.x];  // (the level's descriptors in launch order)
  potrfPanelTiles<T>(pickData(dref) + pd.diagOff, pd.nb, pd.lda, blk, sol, &blk[4 * kPanelWidth][0]);
  BSP_STAMP(3);
}

// DIRECT variants (potrfPanelDirect / trsmPanelDirect / updateTileDirect) serve the levels that
// hold ONE panel -- the serial chain of a wide lump.  Their descriptors travel in the kernel
// arguments instead of behind two or three dependent table loads, and every global load of a
// workgroup is issued up front: the chain is bound by memory round trips (2-3x longer while a
// bulk update saturates the memory system beside it), so each kernel is cut to one load round
// trip plus the store.
template <typename T>
__global__ __launch_bounds__(256) void potrfPanelDirect(PanelDesc pd, DataRef<T> dref,
                                                        T* dinvOut) {
  __shared__ T blk[8 * kPanelWidth][4];
  T(*sol)[4] = blk + 3 * kPanelWidth;
  __shared__ T Ld[kPanelWidth * kInvLd];
  __builtin_amdgcn_s_setprio(3);
  BSP_STAMP(0);
  potrfPanelTiles<T>(pickData(dref) + pd.diagOff, pd.nb, pd.lda, blk, sol, &blk[4 * kPanelWidth][0], NoPreUpdate(), Ld,
                   (GP<T>)dinvOut + (size_t)blockIdx.y * kDinvBatchStride);
  BSP_STAMP(3);
}

// ------------------------------------------------------------------------------------------
// K4  panel trsm: X * L^T = B for a tile of 64 rows below the panel's diagonal block.
// One wave per task; L and the row tile live in LDS (tile transposed, xs[j][row], padded so
// that both the coalesced fill and the per-lane column walk are bank-conflict free).
// Replaces cublas?trsm LEFT/UPPER/OP_C (MatOpsCuda.cu:550-566, 757-781).
// ------------------------------------------------------------------------------------------
// MFMA form of the panel trsm for one tile of 64 rows:  X L^T = B, solved transposed,
//   X_j^T = Dinv_j (B_j^T - sum_{l<j} L_jl X_l^T)     (16x16 blocks, Dinv_j = L_jj^-1)
// Wave w owns rows 16w..16w+15 of the tile, entirely in registers: lane (q, n) = (lane/16,
// lane%16) holds, for row n, the columns 16j + 4q + r (j, r < 4) -- chosen so that accumulator
// register r of a finished block IS the B operand of K-chunk r of the next product (the A
// operand, read from LDS, is indexed to match through Mfma::colOfRow): no shuffles, no LDS round
// trip for X, 40 MFMAs per wave in a dependent chain of 4 x 8.  The four 16x16 inverses are
// computed once per workgroup (wave j inverts block j, lane c < 16 solves column c in registers);
// the solve through inverted 16x16 diagonal blocks is the standard GPU formulation (error grows
// with the condition of a 16x16 block of L, not of the panel).
// Ls: 64 x kTrsmLd (lower triangle of L, identity beyond nb; the inverted diagonal blocks take
// the place of the diagonal blocks).
constexpr int kTrsmLd = 66;
constexpr int kTrsmLdsElems = kPanelWidth * kTrsmLd;
// the solve of trsmTileMfma once L is in Ls and the tile's rows are in x (wave w: rows 16 w ..):
// inversion of the four diagonal blocks, the four dependent stages, the store
template <typename T>
__device__ __forceinline__ void trsmTileSolveStore(GP<T> row, bool active, int nb, T* Ls,
                                                   typename Mfma<T>::Acc (&x)[4]) {
  constexpr int LDT = kTrsmLd;
  typedef T TV2 __attribute__((ext_vector_type(2), aligned(sizeof(T))));
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, q = lane >> 4;
  using Acc = typename Mfma<T>::Acc;
  __syncthreads();
  {  // wave w inverts diagonal block w
    T y[16];
    invertColumn16(Ls + (16 * w) * LDT + 16 * w, LDT, n, y);
    // (round 4: written over the diagonal block itself -- only this wave reads it, all its reads
    //  precede these writes, and the stages below use the off-diagonal blocks only: 33.8 instead of
    //  43 KB of LDS, four workgroups per CU instead of three)
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; i++) Ls[(16 * w + i) * LDT + 16 * w + n] = y[i];
    }
  }
  __syncthreads();
  const int pm = Mfma<T>::colOfRow(n);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (16 * j < nb) {
#pragma unroll
      for (int l = 0; l < j; l++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          x[j] = Mfma<T>::run(-Ls[(16 * j + pm) * LDT + 16 * l + 4 * q + r], x[l][r], x[j]);
        }
      }
      Acc y = {0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        y = Mfma<T>::run(Ls[(16 * j + pm) * LDT + 16 * j + 4 * q + r], x[j][r], y);
      }
      x[j] = y;
    }
  }
  typedef __attribute__((address_space(1))) TV2* GP2w;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = 16 * j + 4 * q;
    if (active && c + 3 < nb) {
      *(GP2w)(row + c) = TV2{x[j][0], x[j][1]};
      *(GP2w)(row + c + 2) = TV2{x[j][2], x[j][3]};
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (active && c + r < nb) row[c + r] = x[j][r];
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void trsmTileMfma(GP<const T> A, GP<T> P, int lda, int nb, int rows,
                                             T* Ls) {
  constexpr int LDT = kTrsmLd, N = kPanelWidth;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, q = lane >> 4;
  const bool active = 16 * w + n < rows;
  GP<T> row = P + (int64_t)(active ? 16 * w + n : 0) * lda;
  using Acc = typename Mfma<T>::Acc;
  Acc x[4];
  // (round 3: two values per lane and load where the panel is at least two columns wide -- on the
  //  batched GRID workload the texture addresser is busy 72 % of this kernel, ~22 cycles per wave
  //  load whatever it fetches; rows are only sizeof(T)-aligned, the pair type says so)
  typedef T TV2 __attribute__((ext_vector_type(2), aligned(sizeof(T))));
  typedef __attribute__((address_space(1))) const TV2* GP2;
  if (nb >= 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = 16 * j + 4 * q;
      if (c + 3 < nb) {
        const TV2 lo = *(GP2)(row + c), hi = *(GP2)(row + c + 2);
        x[j][0] = lo.x;
        x[j][1] = lo.y;
        x[j][2] = hi.x;
        x[j][3] = hi.y;
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) x[j][r] = row[min(c + r, nb - 1)];
      }
    }
    // L: lane -> (row tid / 32 + 8 it, columns 2 (tid % 32) and + 1); the strictly upper entries of
    // the square diagonal block are stored too (unspecified values, masked below)
    TV2 v2[8];
    const int jj = 2 * (tid & 31), jc = min(jj, nb - 2);
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int i = min((tid >> 5) + 8 * it, nb - 1);
      v2[it] = *(GP2)(A + (int64_t)i * lda + jc);
    }
    const bool sh = jj > nb - 2;  // (pair clamped back by one: column jj is its second value)
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int i = (tid >> 5) + 8 * it;
      const T e0 = sh ? v2[it].y : v2[it].x, e1 = v2[it].y;
      Ls[i * LDT + jj] = (i < nb && jj <= i) ? e0 : ((i >= nb && i == jj) ? T(1) : T(0));
      Ls[i * LDT + jj + 1] = (i < nb && jj + 1 <= i && jj + 1 < nb) ? e1 : ((i >= nb && i == jj + 1) ? T(1) : T(0));
    }
  } else {
    T v[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int r = 0; r < 4; r++) x[j][r] = row[min(16 * j + 4 * q + r, nb - 1)];
    }
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int e = tid + 256 * it, i = min(e / N, nb - 1), jj = e % N;
      v[it] = A[(int64_t)i * lda + min(jj, i)];
    }
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int e = tid + 256 * it, i = e / N, jj = e % N;
      Ls[i * LDT + jj] = (i < nb && jj <= i) ? v[it] : ((i >= nb && i == jj) ? T(1) : T(0));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 4; r++) x[j][r] = (active && 16 * j + 4 * q + r < nb) ? x[j][r] : T(0);
  }
  trsmTileSolveStore<T>(row, active, nb, Ls, x);
}

// Register-only form for the chain (single-panel levels): the inverses of the diagonal blocks
// come from the panel's potrf (dinv[4][16][16]), the off-diagonal blocks of L straight from the
// matrix; every operand is loaded in MFMA A-operand layout, no LDS, no barrier.
template <typename T>
struct TrsmOps {
  T L[6][4];  // -(L_jl), (j,l) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
  T D[4][4];  // Dinv_j
};
template <typename T>
__device__ __forceinline__ void trsmLoadOps(GP<const T> A, GP<const T> dinv, int lda, int nb,
                                            int lane, TrsmOps<T>& o) {
  const int pm = Mfma<T>::colOfRow(lane & 15), q = lane >> 4;
#pragma unroll
  for (int j = 1; j < 4; j++) {
#pragma unroll
    for (int l = 0; l < j; l++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        o.L[j * (j - 1) / 2 + l][r] = A[(int64_t)min(16 * j + pm, nb - 1) * lda + 16 * l + 4 * q + r];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 4; r++) o.D[j][r] = dinv[(16 * j + pm) * 16 + 4 * q + r];
  }
}
template <typename T>
__device__ __forceinline__ void trsmMaskOps(int nb, int lane, TrsmOps<T>& o) {
  const int pm = Mfma<T>::colOfRow(lane & 15);
#pragma unroll
  for (int j = 1; j < 4; j++) {
#pragma unroll
    for (int l = 0; l < j; l++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        T& v = o.L[j * (j - 1) / 2 + l][r];
        v = (16 * j + pm < nb) ? -v : T(0);
      }
    }
  }
}
// x: rows of the tile in the layout of trsmTileMfma (lane (q, n): row n, columns 16j + 4q + r)
template <typename T>
__device__ __forceinline__ void trsmLoadRows(GP<const T> row, int nb, int lane,
                                             typename Mfma<T>::Acc (&x)[4]) {
  const int q = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 4; r++) x[j][r] = row[min(16 * j + 4 * q + r, nb - 1)];
  }
}
template <typename T>
__device__ __forceinline__ void trsmMaskRows(bool active, int nb, int lane,
                                             typename Mfma<T>::Acc (&x)[4]) {
  const int q = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 4; r++) x[j][r] = (active && 16 * j + 4 * q + r < nb) ? x[j][r] : T(0);
  }
}
template <typename T>
__device__ __forceinline__ void trsmStoreRows(GP<T> row, bool active, int nb, int lane,
                                              const typename Mfma<T>::Acc (&x)[4]) {
  const int q = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int col = 16 * j + 4 * q + r;
      if (active && col < nb) row[col] = x[j][r];
    }
  }
}
// PAIR: two independent row sets go through the stages together (twice the MFMA parallelism, and
// the operands of a stage die after it)
template <typename T, bool PAIR = false>
__device__ __forceinline__ void trsmStages(const TrsmOps<T>& o, int nb,
                                           typename Mfma<T>::Acc (&x)[4],
                                           typename Mfma<T>::Acc* x2 = nullptr) {
  using Acc = typename Mfma<T>::Acc;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (16 * j < nb) {
#pragma unroll
      for (int l = 0; l < j; l++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          x[j] = Mfma<T>::run(o.L[j * (j - 1) / 2 + l][r], x[l][r], x[j]);
          if (PAIR) x2[j] = Mfma<T>::run(o.L[j * (j - 1) / 2 + l][r], x2[l][r], x2[j]);
        }
      }
      Acc y = {0, 0, 0, 0}, y2 = {0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        y = Mfma<T>::run(o.D[j][r], x[j][r], y);
        if (PAIR) y2 = Mfma<T>::run(o.D[j][r], x2[j][r], y2);
      }
      x[j] = y;
      if (PAIR) x2[j] = y2;
    }
  }
}
// one tile of 64 rows: wave w takes rows 16w..16w+15
template <typename T>
__device__ __forceinline__ void trsmTileRegs(GP<const T> A, GP<const T> dinv, GP<T> P, int lda,
                                             int nb, int rows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15;
  const bool active = 16 * w + n < rows;
  GP<T> row = P + (int64_t)(active ? 16 * w + n : 0) * lda;
  typename Mfma<T>::Acc x[4];
  TrsmOps<T> o;
  trsmLoadRows<T>(row, nb, lane, x);
  trsmLoadOps<T>(A, dinv, lda, nb, lane, o);
  trsmMaskRows<T>(active, nb, lane, x);
  trsmMaskOps<T>(nb, lane, o);
  trsmStages<T>(o, nb, x);
  trsmStoreRows<T>(row, active, nb, lane, x);
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void trsmPanel(const TrsmTaskFat* tasks, DataRef<T> dref) {
  __shared__ T lds[kTrsmLdsElems];
  const TrsmTaskFat pd = tasks[blockIdx.x];  // (panel fields + row tile in one uniform load)
  const TrsmTaskFat& task = pd;
  GP<T> data = pickData(dref);
  const int nb = pd.nb, lda = pd.lda;
  GP<T> P = data + pd.diagOff + (int64_t)(nb + task.rowTile) * lda;
  const int rows = min(kTile, pd.rowsBelow - task.rowTile);
  trsmTileMfma<T>(data + pd.diagOff, P, lda, nb, rows, lds);
}

// K4p  (round 6) the trsm of a SMALL tree level with the level's potrf folded in: every row tile factors
// its own copy of the panel's diagonal block in LDS (same code as potrfPanel: bitwise the same factor)
// before it solves, so the level is two launches instead of three; nothing is written to the diagonal
// block here -- its workgroups all read the unfactored block -- and ONE potrfPanel launch over all such
// panels stores the factors at the end of the factorisation (nothing reads them before: update tiles
// take the solved rows, the chain its own panel).  Taken for multi-panel levels whose trsm launch is
// one round of workgroups (tasks x batch <= 512; launchLevels).  Measured in round 4 (-4 % on GRID
// 82x82, profiles/r04_ab_potrf_in_trsm.txt), built into the library in round 6.
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void trsmPanelPotrf(
    const TrsmTaskFat* tasks, DataRef<T> dref) {
  constexpr int LDT = kTrsmLd;
  __shared__ T Ls[kTrsmLdsElems];
  __shared__ T blk[8 * kPanelWidth][4];
  const TrsmTaskFat pd = tasks[blockIdx.x];
  GP<T> data = pickData(dref);
  const int nb = pd.nb, lda = pd.lda;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, q = lane >> 4;
  const int rows = min(kTile, pd.rowsBelow - pd.rowTile);
  const bool active = 16 * w + n < rows;
  GP<T> row = data + pd.diagOff + (int64_t)(nb + pd.rowTile + (active ? 16 * w + n : 0)) * lda;
  using Acc = typename Mfma<T>::Acc;
  Acc x[4];
  // the tile's rows first: they travel while the diagonal block is factored
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 4; r++) x[j][r] = row[min(16 * j + 4 * q + r, nb - 1)];
  }
  for (int e = tid; e < kPanelWidth * LDT; e += 256) {
    const int i = e / LDT, jj = e - i * LDT;
    Ls[e] = (i >= nb && i == jj) ? T(1) : T(0);
  }
  __syncthreads();
  potrfTilesBlocked<T, NoPreUpdate, false>(data + pd.diagOff, nb, lda, &blk[0][0], &blk[4 * kPanelWidth][0],
                                           NoPreUpdate(), nullptr, nullptr, Ls, LDT);
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int r = 0; r < 4; r++) x[j][r] = (active && 16 * j + 4 * q + r < nb) ? x[j][r] : T(0);
  }
  trsmTileSolveStore<T>(row, active, nb, Ls, x);
}

template <typename T>
__global__ __launch_bounds__(256) void trsmPanelDirect(PanelDesc pd, DataRef<T> dref,
                                                       const T* dinv) {
  __builtin_amdgcn_s_setprio(3);
  GP<T> data = pickData(dref);
  const int nb = pd.nb, lda = pd.lda, rowTile = blockIdx.x * kTile;
  GP<T> P = data + pd.diagOff + (int64_t)(nb + rowTile) * lda;
  trsmTileRegs<T>(data + pd.diagOff, (GP<const T>)dinv + (size_t)blockIdx.y * kDinvBatchStride, P,
                  lda, nb, min(kTile, pd.rowsBelow - rowTile));
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
#if defined(BSP_TRACE_UPD)
// in-situ trace of ONE workgroup (the middle one, first matrix) of every updateTile launch: wall
// clock (100 MHz) at start / tables + first fetch issued / first chunk in LDS / K loop done / old
// values arrived / end; slot 6 = K, slot 7 = -(workgroups of the launch) marks the record
#define UPD_STAMP(slot, val)                                                            \
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0) {             \
    if ((slot) == 0) bspTraceSlot = atomicAdd(&bspTraceCount, 1u) & 8191u;              \
    bspTrace[bspTraceSlot * kTraceW + (slot)] = (val);                                  \
  }
#else
#define UPD_STAMP(slot, val)
#endif
constexpr int kUpdChunk = 32;  // K chunk of updateTile: 2 x 64 x 34 doubles = 35 KB LDS -> 4 WG/CU
// PREFETCH: the next K chunk is requested before the current one is multiplied.  For launches of
// at most ~2 rounds of workgroups (small batches, the top of an elimination tree) a tile's time is
// its chain of dependent memory round trips -- one per K chunk -- and this overlaps them with the
// multiplies; in saturated launches the other workgroups of the CU already do, and the staging
// registers held across the multiplies cost more than they bring (64 x GRID: 11.67 against 11.20 ms).
template <typename T, bool PREFETCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PREFETCH ? 3 : 4, 4))) void updateTile(const UpdTaskWide* tasks, const int64_t* chainOffTab,
                                                  const int32_t* rowChain, const int32_t* rowLocal,
                                                  const int32_t* rowColOff, DataRef<T> dref,
                                                  T* altTarget = nullptr, int64_t altStride = 0,
                                                  int atomicMask = 3) {
  // altTarget: write the (negated) product into a separate buffer instead of `data`
  // (per-op saveSyrkGemm: the frontal temp buffer, one slice per batch entry)
  constexpr int KC = kUpdChunk, LD = KC + 2;
  __shared__ T As[kTile * LD];
  __shared__ T Bs[kTile * LD];
  __shared__ int64_t rowBase[kTile];
  __shared__ int32_t colOff[kTile];

  UPD_STAMP(0, (long long)wall_clock64());
  // Round 4: ONE uniform load (task, segment and source fields side by side, hip_plan.h) instead of
  // the task -> segment -> source chain of three dependent ones.
  const UpdTaskWide w = tasks[blockIdx.x];
  if (w.K <= 0) return;
  GP<T> data = pickData(dref);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = w.K, lda = w.lda;
  UPD_STAMP(6, (long long)K);
  UPD_STAMP(7, -(long long)(gridDim.x * gridDim.y));
  UPD_STAMP(1, (long long)wall_clock64());
  GP<const T> P = data + w.srcOff;  // first row below the source columns
  const bool diagTile = w.rowTile == w.colTile;
  const int segEnd = w.segEnd;

  // K loop in chunks of KC source columns.  Staging map: k = tid % KC, rows (tid / KC) + (256/KC)*it;
  // all loads of a chunk are issued before the first LDS write (memory-level parallelism), then
  // every wave runs its 2x2 MFMA tiles over the chunk.
  // Round 3: TWO source columns per lane and load (PMC on the batched GRID workload: the texture
  // addresser is busy 54 % of this kernel and ~22 cycles per wave load whatever it fetches, so the
  // operands come with half as many loads).  Staging map: k = 2 (tid % 16) and k + 1, rows
  // (tid / 16) + 16 it.  Rows are only 8-byte aligned: the pair type says so.  The pair (k, k + 1) is
  // clamped into [0, K - 2] and sorted out when it is stored (a source of one column keeps the
  // one-column map).
  typedef T TV2 __attribute__((ext_vector_type(2), aligned(sizeof(T))));
  typedef __attribute__((address_space(1))) const TV2* GP2;
  constexpr int RSTEP = 16, NIT = kTile / RSTEP;
  const int sk = 2 * (tid % 16), sr = tid / 16;
  const bool wide = K >= 2;
  TV2 va[NIT], vb[NIT];
  auto fetch = [&](int kBase) {
    const int kcl = wide ? min(kBase + sk, K - 2) : 0;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int qa = min(w.rowTile + sr + RSTEP * it, w.rowsBelow - 1);
      GP<const T> pa = P + (int64_t)qa * lda + kcl;
      if (wide) {
        va[it] = *(GP2)pa;
      } else {
        va[it] = TV2{*pa, T(0)};
      }
    }
    if (!diagTile) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int qb = min(w.colTile + sr + RSTEP * it, segEnd - 1);
        GP<const T> pb = P + (int64_t)qb * lda + kcl;
        if (wide) {
          vb[it] = *(GP2)pb;
        } else {
          vb[it] = TV2{*pb, T(0)};
        }
      }
    }
  };
  // Round 4: the first chunk is requested BEFORE the row / column tables are looked up (a board
  // segment's tables are two more dependent loads: row -> chain -> chain offset), not after them.
  fetch(0);

  // per-row / per-column target addressing of this tile
  if (tid < kTile) {
    const int q = w.rowTile + tid;
    int64_t base = 0;
    if (q < w.rowsBelow) {
      if (w.kind == kSegIntra) {
        base = w.tgtBase + (int64_t)q * w.tgtStride;
      } else {
        const int rr = w.lumpRowBase + (q - w.nRest);
        base = chainOffTab[w.chainTabPtr + (rowChain[rr] - w.firstChainOrd)] +
               (int64_t)rowLocal[rr] * w.tgtStride;
      }
    }
    rowBase[tid] = base;
  } else if (tid < 2 * kTile) {
    const int cidx = tid - kTile;
    const int q = w.colTile + cidx;
    int32_t off = 0;
    if (q < segEnd) off = w.kind == kSegIntra ? q : rowColOff[w.lumpRowBase + (q - w.nRest)];
    colOff[cidx] = off;
  }
  const T* Bt = diagTile ? As : Bs;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  using Acc = typename Mfma<T>::Acc;
  Acc acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  // a diagonal tile only needs sub-tiles on or below the diagonal
  const bool skipUpper = diagTile && wr < wc;
  // (Round 3 skipped the MFMA steps of 16 x 16 sub-tiles that only reach masked-off entries --
  //  a quarter of the steps on GRID 82 x 82 -- behind wave-uniform flags: measured neutral, 11.20
  //  against 11.19 ms on the batched GRID workload, and the second form of the loop cost 23
  //  registers; removed in round 4.)
  GP<T> tbase = altTarget ? (GP<T>)altTarget + (int64_t)blockIdx.y * altStride : data;

  for (int kBase = 0; kBase < K; kBase += KC) {
    const int kc = min(KC, K - kBase);
    const int kPad = (kc + 3) & ~3;
    if (!PREFETCH && kBase > 0) fetch(kBase);
    if (kBase > 0) __syncthreads();  // the previous chunk has been consumed
    // element k of the chunk sits in .x of the loaded pair unless the pair was clamped back by one
    // (k = K - 1 with K odd)
    const bool shifted = wide && kBase + sk > K - 2;
    const bool ok0 = sk < kc, ok1 = sk + 1 < kc;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int r = sr + RSTEP * it;
      const bool rowOk = w.rowTile + r < w.rowsBelow;
      As[r * LD + sk] = (ok0 && rowOk) ? (shifted ? va[it].y : va[it].x) : T(0);
      As[r * LD + sk + 1] = (ok1 && rowOk) ? va[it].y : T(0);
    }
    if (!diagTile) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int r = sr + RSTEP * it;
        const bool rowOk = w.colTile + r < segEnd;
        Bs[r * LD + sk] = (ok0 && rowOk) ? (shifted ? vb[it].y : vb[it].x) : T(0);
        Bs[r * LD + sk + 1] = (ok1 && rowOk) ? vb[it].y : T(0);
      }
    }
    __syncthreads();
    if (kBase == 0) { UPD_STAMP(2, (long long)wall_clock64()); }
    if (PREFETCH && kBase + KC < K) fetch(kBase + KC);
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
    }
  }
  UPD_STAMP(3, (long long)wall_clock64());
  if (!skipUpper) {
    // Scatter.
    // Round 3, built and measured on the batched GRID workload, not kept:
    //  * pair form (the lanes of a column pair swap one accumulator per two rows through DPP, each
    //    then owns both columns of a row: one 16-byte read and write instead of two 8-byte ones,
    //    wave-uniformly per sub-tile when every pair is adjacent in the target): 11.19 against
    //    11.15 ms -- the instruction count of the read-modify-write is not the bound;
    //  * old values requested BEFORE the K loop, or right after chunk 0 went to LDS (overlapping
    //    one of the tile's dependent memory round trips with the operand fetch): 16 values held
    //    across the loop do not fit the 128 registers of 4 waves per SIMD in fp64;
    //  * round 4: requested before the K loop straight INTO the accumulator registers (which then
    //    hold -old + sum a b, negated once when chunk 0 is in LDS; no register added): GRID 82 x 82
    //    1.200 against 1.192 ms, batch of 64 10.82 against 10.78 -- hiding this round trip buys
    //    nothing (and in fp32 the products, added one by one to a large old value, are each rounded
    //    at its magnitude: the per-op boundary differed from the fused path by 2e-6);
    //  * next chunk fetched during the multiplies in EVERY launch: 11.67 against 11.20 ms (now the
    //    PREFETCH variant, for launches of a few rounds only).
    const Acc* accs[4] = {&acc00, &acc01, &acc10, &acc11};
    const bool atomicTile = (w.atomic & atomicMask) != 0;
    T old[16];
    if (!atomicTile) {  // gather all 16 old values first (independent loads in flight together)
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int32_t co = colOff[wc + (t & 1) * 16 + li];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          // (masked-off entries point at valid memory)
          old[t * 4 + reg] = *(tbase + rowBase[wr + (t >> 1) * 16 + Mfma<T>::row(lane, reg)] + co);
        }
      }
#if defined(BSP_TRACE_UPD)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      UPD_STAMP(4, (long long)wall_clock64());
#endif
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int r0 = wr + (t >> 1) * 16, c0 = wc + (t & 1) * 16;
      const int cIn = c0 + li;
      const int qc = w.colTile + cIn;
      const int32_t co = colOff[cIn];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int rIn = r0 + Mfma<T>::row(lane, reg);
        const int qr = w.rowTile + rIn;
        const bool ok = qc < segEnd && qr < w.rowsBelow && qr >= qc && qr >= w.rowMin;
        GP<T> ptr = tbase + rowBase[rIn] + co;
        if (ok) {
          if (atomicTile) {
            atomicSub(ptr, (*accs[t])[reg]);
          } else {
            *ptr = old[t * 4 + reg] - (*accs[t])[reg];
          }
        }
      }
    }
  }
#if defined(BSP_TRACE_UPD)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  UPD_STAMP(5, (long long)wall_clock64());
#endif
}

// K5b  the bulk of the bulk: plain intra-lump tiles whose source width is a multiple of the K
// chunk (every lookahead unit, rank 256).  Against updateTile:
//   * the task is self-contained: one uniform 64-byte load instead of the task -> segment ->
//     source chain of dependent loads, no row / column tables in LDS;
//   * operands go straight from global memory to LDS (global_load_lds_dwordx4: lane t of a wave
//     lands at base + 16 t bytes, so one wave instruction fills 4 rows of 32 doubles / 8 rows of
//     32 floats): no staging registers, no LDS stores, no masking -- rows beyond the end are
//     clamped onto valid ones, their products only reach entries the scatter masks off;
//   * rows are unpadded in LDS; instead the 16-byte slot index of a row is XOR-swizzled with the
//     row (fp64: slot ^ (r & 15), fp32: slot ^ ((r >> 1) & 7)), which keeps the MFMA operand
//     fetch (lane (li, lk) reads [li][k0 + lk]) at the two-lanes-per-8-bytes minimum;
//   * the old target values are requested before the K loop.
// tools/tile_update_probe.hip has the structure study: MFMA + LDS-read loop alone 66-71 TF/s at
// 3-4 workgroups per CU, register staging 55-60, direct-to-LDS 61-65, + read-modify-write of the
// target ~ -15 %.
template <typename T>
struct BulkSwizzle {
  static constexpr int E = 16 / (int)sizeof(T);  // elements per 16-byte slot
  static __device__ __forceinline__ int key(int r) { return sizeof(T) == 8 ? (r & 15) : ((r >> 1) & 7); }
  // element offset of (row r, column k) inside an unpadded [64][kUpdChunk] operand
  static __device__ __forceinline__ int at(int r, int k) {
    return r * kUpdChunk + E * ((k / E) ^ key(r)) + (k % E);
  }
};
// As, Bs: kTile * kUpdChunk values each, 16-byte aligned.
template <typename T>
__device__ __forceinline__ void bulkTileBody(const UpdTaskFat& t, GP<T> data, T* As, T* Bs,
                                             const unsigned* yieldFlag = nullptr, int atomicMask = 3) {
  constexpr int KC = kUpdChunk, E = BulkSwizzle<T>::E, SLOTS = KC / E, RPI = 64 / SLOTS;
  constexpr int NI = kTile / (4 * RPI);  // wave instructions per operand and wave
  typedef __attribute__((address_space(1))) const void* GV;
  typedef __attribute__((address_space(3))) void* LV;
  GP<const T> P = data + t.srcOff;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = t.K, lda = t.lda;
  const bool diagTile = t.rowTile == t.colTile;
  const T* Bt = diagTile ? As : Bs;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  using Acc = typename Mfma<T>::Acc;
  Acc acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const bool skipUpper = diagTile && wr < wc;

  // wave w, instruction it: rows RPI * (4 it + w) .. + RPI - 1; lane -> (row, slot)
  GP<const T> srcA[NI], srcB[NI];
#pragma unroll
  for (int it = 0; it < NI; it++) {
    const int r = RPI * (4 * it + wave) + lane / SLOTS;
    const int slot = (lane % SLOTS) ^ BulkSwizzle<T>::key(r);
    srcA[it] = P + (int64_t)min(t.rowTile + r, t.rowsBelow - 1) * lda + E * slot;
    srcB[it] = P + (int64_t)min(t.colTile + r, t.segEnd - 1) * lda + E * slot;
  }
  // The old target values are NOT requested before the K loop (their 32 registers would be live
  // through it): the kernel stays within 96 registers, so that a chain workgroup (224) fits on a
  // CU beside three of these.  With 120 the chain's workgroups queued behind the remaining rounds
  // of a bulk launch (17 us per chain launch on average, 100+ us in the worst launches:
  // tools/trace_extents.py).  No-return atomics for every tile (no old values at all) cost the
  // bulk 11 % (3.74 -> 4.15 ms serialised).
  // (Round 4: requesting them before the loop straight INTO the accumulator registers -- which
  //  then hold -old + sum a b, no register added -- measured 0.7 % slower on BAL-871, 6.37-6.44
  //  against 6.33-6.38 ms: the first chunk then waits for sixteen HBM misses instead of an L2 hit.)
  GP<T> tgt = data + t.tgtBase;
  const int oa0 = BulkSwizzle<T>::at(wr + li, lk), oa1 = BulkSwizzle<T>::at(wr + 16 + li, lk);
  const int ob0 = BulkSwizzle<T>::at(wc + li, lk), ob1 = BulkSwizzle<T>::at(wc + 16 + li, lk);
  const int ka0 = BulkSwizzle<T>::key(wr + li), ka1 = BulkSwizzle<T>::key(wr + 16 + li);
  const int kb0 = BulkSwizzle<T>::key(wc + li), kb1 = BulkSwizzle<T>::key(wc + 16 + li);
  const unsigned myCu = yieldFlag ? cuKey() : 0u;
  for (int kBase = 0; kBase < K; kBase += KC) {
    if (kBase > 0) __syncthreads();  // the previous chunk has been consumed
    // (cooperative CU yield: the word is fetched with the chunk and looked at after it has landed)
    const unsigned yf = yieldFlag ? yieldPeek(yieldFlag) : 0u;
#pragma unroll
    for (int it = 0; it < NI; it++) {
      __builtin_amdgcn_global_load_lds((GV)(srcA[it] + kBase), (LV)(As + RPI * (4 * it + wave) * KC),
                                       16, 0, BSP_BULK_AUX);
    }
    if (!diagTile) {
#pragma unroll
      for (int it = 0; it < NI; it++) {
        __builtin_amdgcn_global_load_lds((GV)(srcB[it] + kBase),
                                         (LV)(Bs + RPI * (4 * it + wave) * KC), 16, 0, BSP_BULK_AUX);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (yieldFlag && (unsigned)__builtin_amdgcn_readfirstlane((int)yf) == myCu) {
      yieldWhile(yieldFlag, myCu);
    }
    if (!skipUpper) {
#pragma unroll
      for (int k0 = 0; k0 < KC; k0 += 4) {
        // (row r, column k0 + lk): slot (k0 + lk) / E moves by k0 / E under the XOR key.  (Reading
        //  the operands of step s+1 before the MFMAs of step s -- two register sets, pinned with a
        //  sched_barrier -- was measured slower: 4.02 against 3.8 ms for the serialised bulk.  Round 4
        //  again, letting the compiler pipeline the unrolled loop -- reads one step ahead,
        //  `s_waitcnt lgkmcnt(2)`, 86 registers: BAL-871 6.43-6.50 against 6.31-6.37 ms, FLAT-50k 27.27
        //  against 26.86, profiles/r04_ab_bulk_pipelined_reads.txt.  A wave that never waits for its
        //  own operands takes the matrix pipe from the waves whose chunk loads it should be hiding.)
        const int s = k0 / E;
        const T a0 = As[oa0 + E * (((lk / E + s) ^ ka0) - ((lk / E) ^ ka0))];
        const T a1 = As[oa1 + E * (((lk / E + s) ^ ka1) - ((lk / E) ^ ka1))];
        const T b0 = Bt[ob0 + E * (((lk / E + s) ^ kb0) - ((lk / E) ^ kb0))];
        const T b1 = Bt[ob1 + E * (((lk / E + s) ^ kb1) - ((lk / E) ^ kb1))];
        acc00 = Mfma<T>::run(a0, b0, acc00);
        acc01 = Mfma<T>::run(a0, b1, acc01);
        acc10 = Mfma<T>::run(a1, b0, acc10);
        acc11 = Mfma<T>::run(a1, b1, acc11);
      }
    }
  }
  if (!skipUpper) {
    const Acc* accs[4] = {&acc00, &acc01, &acc10, &acc11};
    if (t.atomic & atomicMask) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int qc = t.colTile + wc + (q & 1) * 16 + li;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int qr = t.rowTile + wr + (q >> 1) * 16 + Mfma<T>::row(lane, reg);
          if (qc < t.segEnd && qr < t.rowsBelow && qr >= qc && qr >= t.rowMin) {
            atomicSub(tgt + (int64_t)qr * t.tgtStride + qc, (*accs[q])[reg]);
          }
        }
      }
    } else {
      // single writer: read - subtract - write, the 16 old values requested together AFTER the K
      // loop (masked-off entries clamped onto valid ones)
      T old[16];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int qc = min(t.colTile + wc + (q & 1) * 16 + li, t.segEnd - 1);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int qr = min(t.rowTile + wr + (q >> 1) * 16 + Mfma<T>::row(lane, reg), t.rowsBelow - 1);
          old[q * 4 + reg] = tgt[(int64_t)qr * t.tgtStride + qc];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int qc = t.colTile + wc + (q & 1) * 16 + li;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int qr = t.rowTile + wr + (q >> 1) * 16 + Mfma<T>::row(lane, reg);
          if (qc < t.segEnd && qr < t.rowsBelow && qr >= qc && qr >= t.rowMin) {
            tgt[(int64_t)qr * t.tgtStride + qc] = old[q * 4 + reg] - (*accs[q])[reg];
          }
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void updateTileBulk(
    const UpdTaskFat* tasks, DataRef<T> dref, const unsigned* yieldFlag, int atomicMask) {
  __shared__ __attribute__((aligned(16))) T As[kTile * kUpdChunk];
  __shared__ __attribute__((aligned(16))) T Bs[kTile * kUpdChunk];
  const UpdTaskFat t = tasks[blockIdx.x];
  bulkTileBody<T>(t, pickData(dref), As, Bs, yieldFlag, atomicMask);
}

// K5d  direct variant for the update tiles of a one-panel level that all belong to ONE
// intra-lump segment: descriptors by value, the tile is decoded from blockIdx.x (same order as
// the plan's task list: column tiles of the first `mNow` columns, each with its row tiles, then
// the XCD-contiguous permutation), the old target values are fetched together with the first K
// chunk and the next chunk is fetched while the current one is multiplied.
// XCD-contiguous permutation of a launch's tile indices (workgroup b lands on XCD b % 8): XCD x
// gets a contiguous run of the tile list
__device__ __forceinline__ int xcdContiguous(int b, int n) {
  if (n < 64) return b;
  const int base = n >> 3, extra = n & 7, x = b & 7;
  return x * base + min(x, extra) + (b >> 3);
}

// tile `idx` of the segment in the order: column tiles from sd.q0 on, each with its row tiles
template <typename T>
__device__ __forceinline__ void updateTileDirectBody(const SrcDesc& pd, const SegDesc& sd, int idx,
                                                     GP<T> data, T* As, T* Bs,
                                                     GP<T> rawOut = nullptr, int nbNext = 0) {
  constexpr int KC = kUpdChunk, LD = KC + 2;
  int colTile = sd.q0, rowTile;
  for (;;) {
    const int cnt = (pd.rowsBelow - colTile + kTile - 1) / kTile;
    if (idx < cnt) {
      rowTile = colTile + kTile * idx;
      break;
    }
    idx -= cnt;
    colTile += kTile;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = pd.K, lda = pd.lda;
  GP<const T> P = data + pd.off;
  const bool diagTile = rowTile == colTile;
  const int segEnd = sd.q0 + sd.m;
  const T* Bt = diagTile ? As : Bs;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  using Acc = typename Mfma<T>::Acc;
  Acc acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const bool skipUpper = diagTile && wr < wc;

  constexpr int RSTEP = 256 / KC, NIT = kTile / RSTEP;
  const int sk = tid % KC, sr = tid / KC;
  T va[NIT], vb[NIT];
  auto fetch = [&](int kBase) {
    const int kcl = min(kBase + sk, K - 1);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int qa = min(rowTile + sr + RSTEP * it, pd.rowsBelow - 1);
      va[it] = P[(int64_t)qa * lda + kcl];
    }
    if (!diagTile) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int qb = min(colTile + sr + RSTEP * it, segEnd - 1);
        vb[it] = P[(int64_t)qb * lda + kcl];
      }
    }
  };
  fetch(0);
  // old target values (masked-off entries are clamped onto valid ones)
  GP<T> tgt = data + sd.tgtBase;
  T old[16];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int qc = min(colTile + wc + (t & 1) * 16 + li, segEnd - 1);
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int qr = min(rowTile + wr + (t >> 1) * 16 + Mfma<T>::row(lane, reg), pd.rowsBelow - 1);
      old[t * 4 + reg] = tgt[(int64_t)qr * sd.tgtStride + qc];
    }
  }
  for (int kBase = 0; kBase < K; kBase += KC) {
    const int kc = min(KC, K - kBase);
    const int kPad = (kc + 3) & ~3;
    if (kBase > 0) __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int r = sr + RSTEP * it;
      As[r * LD + sk] = (sk < kc && rowTile + r < pd.rowsBelow) ? va[it] : T(0);
    }
    if (!diagTile) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int r = sr + RSTEP * it;
        Bs[r * LD + sk] = (sk < kc && colTile + r < segEnd) ? vb[it] : T(0);
      }
    }
    __syncthreads();
    if (kBase + KC < K) fetch(kBase + KC);
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
    }
  }
  if (!skipUpper) {
    const Acc* accs[4] = {&acc00, &acc01, &acc10, &acc11};
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int qc = colTile + wc + (t & 1) * 16 + li;
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int qr = rowTile + wr + (t >> 1) * 16 + Mfma<T>::row(lane, reg);
        if (qc < segEnd && qr < pd.rowsBelow && qr >= qc && qr >= sd.rowMin) {
          const T val = old[t * 4 + reg] - (*accs[t])[reg];
          tgt[(int64_t)qr * sd.tgtStride + qc] = val;
          // rows below the next panel's diagonal block, also to the chain's staging buffer
          if (rawOut && colTile == 0 && qc < nbNext && qr >= nbNext) {
            rawOut[(int64_t)(qr - nbNext) * kTile + qc] = val;
          }
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void updateTileDirect(SrcDesc pd, SegDesc sd, int nTasks,
                                                        DataRef<T> dref, T* rawOut, int nbNext,
                                                        int64_t rawStride) {
  constexpr int LD = kUpdChunk + 2;
  __shared__ T As[kTile * LD];
  __shared__ T Bs[kTile * LD];
  __builtin_amdgcn_s_setprio(BSP_TILE_PRIO);
  updateTileDirectBody<T>(pd, sd, xcdContiguous(blockIdx.x, nTasks), pickData(dref), As, Bs,
                          rawOut ? (GP<T>)rawOut + blockIdx.y * rawStride : nullptr, nbNext);
}

// K5f  the same launch with the NEXT panel's potrf fused in.  Tile 0 of the segment is the next
// panel's diagonal block: workgroup 0 applies that tile's update inside the potrf (the block is
// already in MFMA accumulator layout there) and factors it, while the other workgroups update
// the remaining tiles -- the potrf (the longest kernel of the chain) overlaps the update instead
// of following it, with no second stream and no event.
// trsm launch of the LAST panel of an outer block, plus one workgroup (the last) that applies to
// tile 0 of the block-wide segment the update by the block's first `part.K` source columns (they
// are final for the rows of that tile); the potrf workgroup of the following update launch then
// starts at source column part.K (kStart) instead of summing all 256 columns on its own.
template <typename T>
__global__ __launch_bounds__(256) void trsmPanelDirectPlus(PanelDesc pd, SrcDesc part, SegDesc sd,
                                                           DataRef<T> dref, const T* dinv) {
  constexpr int LD = kUpdChunk + 2;
  __shared__ T lds[2 * kTile * LD];
  __builtin_amdgcn_s_setprio(3);
  GP<T> data = pickData(dref);
  if (blockIdx.x == gridDim.x - 1) {
    updateTileDirectBody<T>(part, sd, 0, data, lds, lds + kTile * LD);
    return;
  }
  const int nb = pd.nb, lda = pd.lda, rowTile = blockIdx.x * kTile;
  GP<T> P = data + pd.diagOff + (int64_t)(nb + rowTile) * lda;
  trsmTileRegs<T>(data + pd.diagOff, (GP<const T>)dinv + (size_t)blockIdx.y * kDinvBatchStride, P,
                  lda, nb, min(kTile, pd.rowsBelow - rowTile));
}

template <typename T>
__global__ __launch_bounds__(256) void updateTileDirectPotrf(SrcDesc pd, SegDesc sd, int nTasks,
                                                             PanelDesc next, DataRef<T> dref,
                                                             int kStart, T* dinvOut, T* rawOut,
                                                             int64_t rawStride) {
  constexpr int KC = kUpdChunk, LD = KC + 2;
  __shared__ __attribute__((aligned(16))) T As[kTile * LD];
  __shared__ __attribute__((aligned(16))) T Bs[kTile * LD];
  GP<T> data = pickData(dref);
  if (blockIdx.x != 0) {
    __builtin_amdgcn_s_setprio(BSP_TILE_PRIO);
    const int idx = 1 + xcdContiguous(blockIdx.x - 1, nTasks - 1);
    GP<T> raw = rawOut ? (GP<T>)rawOut + blockIdx.y * rawStride : nullptr;
    // (the bulk tile body -- operands straight to LDS, no register prefetch -- was measured
    //  slower for these tiles: they run on the chain's stream, where latency counts)
    updateTileDirectBody<T>(pd, sd, idx, data, As, Bs, raw, next.nb);
    return;
  }
  __builtin_amdgcn_s_setprio(3);
  T(*blk)[4] = reinterpret_cast<T(*)[4]>(Bs);
  T(*sol)[4] = blk + 3 * kPanelWidth;
  T* Ld = Bs + 4 * kPanelWidth * 4;
  static_assert(4 * kPanelWidth * 4 + kPanelWidth * kInvLd <= kTile * LD, "potrf LDS fits in Bs");
  const int K = pd.K - kStart, lda = pd.lda, nb = next.nb;
  // rows of the next panel, source columns from kStart on
  GP<const T> X = data + pd.off + (int64_t)sd.q0 * lda + kStart;
  using Acc = typename Mfma<T>::Acc;
  auto pre = [&](Acc* acc) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    constexpr int RSTEP = 256 / KC, NIT = kTile / RSTEP;
    const int sk = tid % KC, sr = tid / KC;
    T v[NIT];
    // the product is summed on its own and subtracted once, as the unfused tile update does
    Acc upd[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto fetch = [&](int kBase) {
      const int kcl = min(kBase + sk, K - 1);
#pragma unroll
      for (int it = 0; it < NIT; it++) v[it] = X[(int64_t)min(sr + RSTEP * it, nb - 1) * lda + kcl];
    };
    fetch(0);
    for (int kBase = 0; kBase < K; kBase += KC) {
      const int kc = min(KC, K - kBase);
      const int kPad = (kc + 3) & ~3;
      if (kBase > 0) __syncthreads();
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int r = sr + RSTEP * it;
        As[r * LD + sk] = (sk < kc && r < nb) ? v[it] : T(0);
      }
      __syncthreads();
      if (kBase + KC < K) fetch(kBase + KC);
      for (int k0 = 0; k0 < kPad; k0 += 4) {
        const T a = As[(16 * w + li) * LD + k0 + lk];
#pragma unroll
        for (int tj = 0; tj < 4; tj++) {
          if (tj <= w) upd[tj] = Mfma<T>::run(a, As[(16 * tj + li) * LD + k0 + lk], upd[tj]);
        }
      }
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++) {
      if (tj <= w) acc[tj] -= upd[tj];
    }
  };
  BSP_STAMP(0);
  // (As is free once `pre` has run: the blocked form's second buffer)
  potrfPanelTiles<T>(data + next.diagOff, nb, next.lda, blk, sol, As, pre, Ld,
                     (GP<T>)dinvOut + (size_t)blockIdx.y * kDinvBatchStride);
  BSP_STAMP(3);
}

// ------------------------------------------------------------------------------------------
// K6  one launch per chain step inside an outer block: trsm of the panel + rank-nb update of the
// rest of the block's columns (+ the next panel's potrf), replacing the trsm launch AND the update
// launch of the step.  Workgroup = one 64x64 target tile (i, j) of the segment:
//   * it solves its OWN copies of the panel rows it needs -- X_i (tile rows) and X_j (tile
//     columns) -- with the register-only MFMA trsm (40 MFMAs per wave each), reading the unsolved
//     rows from the chain's staging buffer `rawIn` (64 values per row, written by the previous
//     step's update next to its in-place store): the matrix itself cannot be the source, because
//     the workgroups of column tile 0 store X_i in place while others still need row tile i raw;
//   * X_j goes through LDS once (XB); X_i stays in registers: in the layout of trsmStages it IS
//     the MFMA A operand of the tile product, and XB rows read as 4 consecutive values are the B
//     operand.  Wave w owns target rows 16w..16w+15 x 64 columns (4 MFMA tiles, K = 64: 64 MFMAs);
//   * column tile 0 is the next panel: its tiles are also written to `rawOut` for the next step;
//   * fuse: workgroup 0 (tile (0,0) = the next panel's diagonal block) continues into the potrf.
// ------------------------------------------------------------------------------------------
constexpr int kXbLd = kTile + 2;

// rows ri (tile rows) and rj (tile columns, unless the tile is diagonal) of the staged panel ->
// X_i in registers, X_j in XB (X_i itself for a diagonal tile), X_i stored in place when asked,
// and the tile product D[t] = X_i (X_j rows 16t..16t+15)^T  (diagonal tile: t <= w only)
template <typename T>
struct ChainTile {
  using Acc = typename Mfma<T>::Acc;
  Acc xi[4], xj[4];
  TrsmOps<T> o;
  bool actI, actJ, diagTile;
  __device__ __forceinline__ void load(GP<const T> rawIn, GP<const T> Lkk, GP<const T> dinv, int lda,
                                       int nb, int ri, int rj, int rowsBelow, int segEnd,
                                       bool diag) {
    const int lane = threadIdx.x & 63;
    diagTile = diag;
    actI = ri < rowsBelow;
    actJ = rj < segEnd;
    // (staging rows are 64 values long whatever nb is: no clamping of the column index)
    trsmLoadRows<T>(rawIn + (int64_t)(actI ? ri : 0) * kTile, kTile, lane, xi);
    if (!diagTile) trsmLoadRows<T>(rawIn + (int64_t)(actJ ? rj : 0) * kTile, kTile, lane, xj);
    trsmLoadOps<T>(Lkk, dinv, lda, nb, lane, o);
  }
  __device__ __forceinline__ void solve(int nb, T* XB, GP<T> storeRow /* or null */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    trsmMaskRows<T>(actI, nb, lane, xi);
    trsmMaskOps<T>(nb, lane, o);
    if (!diagTile) {
      trsmMaskRows<T>(actJ, nb, lane, xj);
      trsmStages<T>(o, nb, xj);
    }
    trsmStages<T>(o, nb, xi);
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        XB[(16 * w + n) * kXbLd + 16 * j + 4 * q + r] = diagTile ? xi[j][r] : xj[j][r];
      }
    }
    if (storeRow) trsmStoreRows<T>(storeRow, actI, nb, lane, xi);
    ldsBarrier();
  }
  // D[t] += (rows ri of Pm, columns 0..kMem) (rows rj of Pm)^T: source columns of the outer block
  // that earlier panels already solved (block-last step).  64 columns at a time: the tile-row
  // operand straight into registers (MFMA A-operand layout), the tile-column operand through XB.
  // Rows beyond the end are clamped: their products only reach entries the scatter masks off.
  static __device__ __forceinline__ void multiplyMem(GP<const T> Pm, int lda, int kMem, int rowTile,
                                                     int colTile, int rowsBelow, int segEnd,
                                                     bool diag, T* XB, Acc (&D)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, q = lane >> 4;
    GP<const T> rowI = Pm + (int64_t)min(rowTile + 16 * w + n, rowsBelow - 1) * lda + 4 * q;
    GP<const T> rowJ = Pm + (int64_t)min(colTile + (tid >> 2), segEnd - 1) * lda + 16 * (tid & 3);
    Acc am[4];
    T v[16];
    auto fetch = [&](int kb) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int r = 0; r < 4; r++) am[j][r] = rowI[kb + 16 * j + r];
      }
#pragma unroll
      for (int c = 0; c < 16; c++) v[c] = rowJ[kb + c];
    };
    if (kMem > 0) fetch(0);
    for (int kb = 0; kb < kMem; kb += kTile) {
      Acc a[4];
#pragma unroll
      for (int j = 0; j < 4; j++) a[j] = am[j];
      ldsBarrier();  // XB free
#pragma unroll
      for (int c = 0; c < 16; c++) XB[(tid >> 2) * kXbLd + 16 * (tid & 3) + c] = v[c];
      ldsBarrier();
      if (kb + kTile < kMem) fetch(kb + kTile);  // (the next 64 columns while these are multiplied)
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (!diag || t <= w) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              D[t] = Mfma<T>::run(a[j][r], XB[(16 * t + n) * kXbLd + 16 * j + 4 * q + r], D[t]);
            }
          }
        }
        asm volatile("" ::: "memory");
      }
    }
    if (kMem > 0) ldsBarrier();  // XB is rewritten by solve()
  }
  // D[t] += X_i (X_j rows 16t..16t+15)^T
  __device__ __forceinline__ void multiply(const T* XB, Acc (&D)[4]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (!diagTile || t <= w) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            D[t] = Mfma<T>::run(xi[j][r], XB[(16 * t + n) * kXbLd + 16 * j + 4 * q + r], D[t]);
          }
        }
      }
      // (keeps the 16 LDS reads of the next column group from being hoisted above this one:
      //  all 64 in flight at once cost 128 registers)
      asm volatile("" ::: "memory");
    }
  }
};

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void chainStep(
    PanelDesc pd, SegDesc sd, int nTasks, PanelDesc next, int fuse, DataRef<T> dref,
    const T* rawInBase, T* rawOutBase, int64_t rawStride, const T* dinvInBase, T* dinvOutBase,
    int64_t memOff, int kMem, unsigned* yieldFlag, int traceId, int kMem0, int extraDiag) {
  // kMem0 <= kMem: source columns workgroup 0 still has to apply from memory to tile (0,0), the
  // LAST kMem0 of the kMem ones -- the earlier panels of the block applied theirs already, each in
  // its own step (extraDiag: one more workgroup, the diagonal tile just past the segment's columns
  // = the next outer block's tile (0,0), rank-nb, with atomics: lookahead units of the side stream
  // may be working on that tile too).  The block-last step's potrf workgroup then starts from a
  // plain rank-nb update like every other step instead of a rank-256 one (10-25 us per block).
  __shared__ T XB[kTile * kXbLd];
  BSP_EXTENT_BEGIN(traceId);
  static_assert(4 * kPanelWidth * 4 + kPanelWidth * kInvLd <= kTile * kXbLd, "potrf LDS fits in XB");
  using Acc = typename Mfma<T>::Acc;
  GP<T> data = pickData(dref);
  GP<const T> rawIn = (GP<const T>)rawInBase + blockIdx.y * rawStride;
  GP<const T> dinv = (GP<const T>)dinvInBase + (size_t)blockIdx.y * kDinvBatchStride;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15;
  const int nb = pd.nb, lda = pd.lda, rowsBelow = pd.rowsBelow, segEnd = sd.q0 + sd.m;
  GP<const T> Lkk = data + pd.diagOff;
  GP<T> P = data + pd.diagOff + (int64_t)nb * lda;  // the panel's rows below, in place

  if (fuse && blockIdx.x == 0) {
    // tile (0,0) = the next panel's diagonal block: update it inside the potrf and factor it
    __builtin_amdgcn_s_setprio(3);
    yieldPublish(yieldFlag, cuKey());  // bulk waves on this CU pause until the potrf is done
    T(*blk)[4] = reinterpret_cast<T(*)[4]>(XB);
    T(*sol)[4] = blk + 3 * kPanelWidth;
    T* Ld = XB + 4 * kPanelWidth * 4;
    ChainTile<T> ct;
    const int ri = 16 * w + n;
    if (kMem0 == 0) ct.load(rawIn, Lkk, dinv, lda, nb, ri, ri, rowsBelow, segEnd, true);
    auto pre = [&](Acc* acc) {
      Acc D[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      if (kMem0 > 0) {
        ChainTile<T>::multiplyMem(data + memOff + (kMem - kMem0), lda, kMem0, 0, 0, rowsBelow, segEnd,
                                  true, XB, D);
        ct.load(rawIn, Lkk, dinv, lda, nb, ri, ri, rowsBelow, segEnd, true);
      }
      ct.solve(nb, XB, P + (int64_t)ri * lda);
      ct.multiply(XB, D);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (t <= w) acc[t] -= D[t];
      }
      ldsBarrier();  // XB is about to become the potrf's blk / sol / Ld
    };
    BSP_STAMP(0);
    static_assert(4 * kPanelWidth * 4 + kPanelWidth * kInvLd + 4 * kPanelWidth * 4 <= kTile * kXbLd, "second potrf buffer fits in XB");
    potrfPanelTiles<T>(data + next.diagOff, next.nb, next.lda, blk, sol,
                       XB + 4 * kPanelWidth * 4 + kPanelWidth * kInvLd, pre, Ld,
                       (GP<T>)dinvOutBase + (size_t)blockIdx.y * kDinvBatchStride);
    yieldPublish(yieldFlag, 0u);
    BSP_STAMP(3);
    BSP_EXTENT_END(traceId, true);
    return;
  }

  __builtin_amdgcn_s_setprio(BSP_TILE_PRIO);
  BSP_STAMP_TILE(4);
  const bool extra = extraDiag && (int)blockIdx.x == nTasks;  // (grid = nTasks + 1 then)
  int idx = fuse ? 1 + xcdContiguous(blockIdx.x - 1, nTasks - 1) : xcdContiguous(blockIdx.x, nTasks);
  int colTile = sd.q0, rowTile;
  if (extra) {
    rowTile = colTile = segEnd;
  } else {
    for (;;) {
      const int cnt = (rowsBelow - colTile + kTile - 1) / kTile;
      if (idx < cnt) {
        rowTile = colTile + kTile * idx;
        break;
      }
      idx -= cnt;
      colTile += kTile;
    }
  }
  const int colEnd = extra ? segEnd + kTile : segEnd;  // columns this tile may write
  const int ri = rowTile + 16 * w + n, rj = colTile + 16 * w + n;
  Acc D[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  if (kMem > 0 && !extra) {
    // Tile (0,0) of the block-wide segment skips the panels that applied their update to it early
    // (extraDiag; kMem0 = what is left).  With a fused potrf that tile belongs to workgroup 0, above;
    // a block-last step whose update is this ONE tile (a last outer block of a single panel, no rows
    // below the lump) is not fused, and took all kMem columns again: lump widths of 256 k + 64, k >= 2,
    // got the updates of the block's first three panels twice (found by tools/stress.py seed 9426).
    const int skip = (rowTile == sd.q0 && colTile == sd.q0) ? kMem - kMem0 : 0;
    if (kMem - skip > 0) {
      ChainTile<T>::multiplyMem(data + memOff + skip, lda, kMem - skip, rowTile, colTile, rowsBelow,
                                segEnd, rowTile == colTile, XB, D);
    }
  }
  ChainTile<T> ct;
  ct.load(rawIn, Lkk, dinv, lda, nb, ri, rj, rowsBelow, colEnd, rowTile == colTile);
  // (column tile q0 covers every row tile once: its workgroups store X_i in place)
  ct.solve(nb, XB, colTile == sd.q0 ? P + (int64_t)(ct.actI ? ri : 0) * lda : nullptr);
  BSP_STAMP_TILE(5);
  // (the old target values are fetched after the solve: these tiles are not on the critical path
  //  -- the potrf workgroup is -- and the registers are needed for the trsm operands)
  GP<T> tgt = data + sd.tgtBase;
  T old[16];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int qc = min(colTile + 16 * t + n, colEnd - 1);
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int qr = min(rowTile + 16 * w + Mfma<T>::row(lane, reg), rowsBelow - 1);
      old[t * 4 + reg] = tgt[(int64_t)qr * sd.tgtStride + qc];
    }
  }
  ct.multiply(XB, D);
  BSP_STAMP_TILE(6);
  GP<T> rawOut = (GP<T>)rawOutBase + blockIdx.y * rawStride;
  const int nbNext = next.nb;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int qc = colTile + 16 * t + n;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int qr = rowTile + 16 * w + Mfma<T>::row(lane, reg);
      if (qc < colEnd && qr < rowsBelow && qr >= qc && qr >= sd.rowMin) {
        if (extra) {
          atomicSub(tgt + (int64_t)qr * sd.tgtStride + qc, D[t][reg]);
        } else {
          const T val = old[t * 4 + reg] - D[t][reg];
          tgt[(int64_t)qr * sd.tgtStride + qc] = val;
          if (rawOutBase && colTile == 0 && qc < nbNext && qr >= nbNext) {
            rawOut[(int64_t)(qr - nbNext) * kTile + qc] = val;
          }
        }
      }
    }
  }
  BSP_STAMP_TILE(7);
  BSP_EXTENT_END(traceId, false);
}

// (A 128x128-tile variant of K5 -- 4x4 MFMA tiles per wave, 70 KB LDS, 2 workgroups per CU -- was
// measured at 32-35 TF/s against 42 for the 64x64 tile at K = 256, and removed.)

// empty launch: loads this translation unit's code object (HipSymbolicCtx::prepareDevice)
__global__ void warmupKernel() {}

// ------------------------------------------------------------------------------------------
// Measurement helper: sustained rate of back-to-back independent v_mfma_f64_16x16x4_f64 (4
// accumulators per wave).  bench.py reports it next to the datasheet peak.
// ------------------------------------------------------------------------------------------
__global__ void mfmaF64Probe(double* out, int iters) {
  // (no __launch_bounds__(256): with it hipcc puts the accumulators in AGPRs but keeps the
  //  loop-carried values in VGPRs and copies them over and back around the MFMAs of every
  //  iteration -- 64 moves per 4 MFMAs, which measured 44 TFLOP/s instead of 77)
  double4_t c[4];
#pragma unroll
  for (int k = 0; k < 4; k++) c[k] = double4_t{0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) s += c[k][k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------------
// K8  pseudo-factor of spans (NumericCtx::pseudoFactorSpans, MatOps.h:117; factor_spans_kernel,
// MatOpsCuda.cu:188-233; CPU: factorSpan, MatOpsCpuBase.h:185-210): for every span s of the range,
// in-place Cholesky of its diagonal block and  rows_below <- rows_below * L_ss^-T  on the span's
// columns (the rest of its lump's diagonal block and every chain row below).  One wave per span
// (spans are parameter blocks: a few columns), the block in LDS, lanes over the rows below.
// Not on the factor() path: block-Jacobi / Gauss-Seidel preconditioners use it.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pseudoFactorSpansKernel(SkelDev sk, DataRef<T> dref,
                                                               int64_t spanBegin, int64_t spanEnd) {
  constexpr int NMAX = kElimSmallMax, LD = NMAX + 1;
  __shared__ T diagS[4][NMAX * LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t s = spanBegin + (int64_t)blockIdx.x * 4 + wave;
  if (s >= spanEnd) return;
  const int64_t lump = sk.spanToLump[s];
  const int n = (int)(sk.spanStart[s + 1] - sk.spanStart[s]);
  const int64_t off = sk.spanOffsetInLump[s];
  int64_t first = s;  // first span of the lump
  while (sk.spanOffsetInLump[first] != 0) first--;
  const int64_t idxInLump = s - first;
  const int64_t lda = sk.lumpStart[lump + 1] - sk.lumpStart[lump];
  const int64_t c0 = sk.chainColPtr[lump], nCh = sk.chainColPtr[lump + 1] - c0;
  GP<T> data = pickData(dref);
  GP<T> D = data + sk.chainData[c0 + idxInLump] + off;
  GP<T> B = data + sk.chainData[c0 + idxInLump + 1] + off;
  const int64_t rowsBelow = sk.chainRowsTillEnd[c0 + nCh - 1] - sk.chainRowsTillEnd[c0 + idxInLump];
  T* S = diagS[wave];
  for (int e = lane; e < n * n; e += 64) {
    const int i = e / n, j = e - i * n;
    S[i * LD + j] = D[(int64_t)i * lda + j];
  }
  waveSync();
  for (int j = 0; j < n; j++) {
    const T d = sqrt(S[j * LD + j]);
    waveSync();
    if (lane == 0) S[j * LD + j] = d;
    if (lane > j && lane < n) S[lane * LD + j] /= d;
    waveSync();
    const int rem = n - j - 1;
    for (int e = lane; e < rem * rem; e += 64) {
      const int a = e / rem, b = e - a * rem;
      if (b <= a) S[(j + 1 + a) * LD + (j + 1 + b)] -= S[(j + 1 + a) * LD + j] * S[(j + 1 + b) * LD + j];
    }
    waveSync();
  }
  for (int e = lane; e < n * n; e += 64) {
    const int i = e / n, j = e - i * n;
    if (j <= i) D[(int64_t)i * lda + j] = S[i * LD + j];
  }
  for (int64_t r = lane; r < rowsBelow; r += 64) {
    GP<T> row = B + r * lda;
    T x[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; j++) x[j] = j < n ? row[j] : T(0);
#pragma unroll
    for (int j = 0; j < NMAX; j++) {
      if (j < n) {
        T sres = x[j];
#pragma unroll
        for (int i = 0; i < j; i++) sres -= x[i] * S[j * LD + i];
        x[j] = sres / S[j * LD + j];
      }
    }
#pragma unroll
    for (int j = 0; j < NMAX; j++) {
      if (j < n) row[j] = x[j];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Per-op boundary (NumericCtx::prepareAssemble / assemble, MatOps.h:132-135).  The fused path never
// uses these; they let the reference's own driver loop (Solver.cpp:198-218) run on this backend.
// ------------------------------------------------------------------------------------------
__global__ void prepareAssembleKernel(SkelDev sk, int64_t* spanToChainOffset, int64_t targetLump) {
  const int64_t i = sk.chainColPtr[targetLump] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < sk.chainColPtr[targetLump + 1]) spanToChainOffset[sk.chainRowSpan[i]] = sk.chainData[i];
}

// target column -= frontal product; `negTemp` holds MINUS the product (see saveSyrkGemm), hence +=.
// One workgroup per block row r; block columns c <= r (MatOpsRef.cpp:144-175).
template <typename T>
__global__ __launch_bounds__(256) void assembleKernel(SkelDev sk, const int64_t* spanToChainOffset,
                                                      const T* negTemp, int64_t tempStride,
                                                      DataRef<T> dref, int64_t rectRowBegin,
                                                      int64_t dstStride, int64_t srcColDataOffset,
                                                      int64_t srcRectWidth, int64_t numBlockRows,
                                                      int64_t numBlockCols) {
  const int64_t r = blockIdx.x;
  GP<T> data = pickData(dref);
  const T* temp = negTemp + (int64_t)blockIdx.y * tempStride;
  const int64_t* cre = sk.chainRowsTillEnd + srcColDataOffset;
  const int64_t* toSpan = sk.chainRowSpan + srcColDataOffset;
  const int64_t rBegin = cre[r - 1] - rectRowBegin;
  const int64_t rSize = cre[r] - rBegin - rectRowBegin;
  const int64_t rOffset = spanToChainOffset[toSpan[r]];
  const int64_t cEnd = numBlockCols < r + 1 ? numBlockCols : r + 1;
  // all block columns 0..cEnd-1 are contiguous columns [0, colsTotal) of the frontal rectangle
  const int64_t colsTotal = cre[cEnd - 1] - rectRowBegin;
  for (int64_t e = threadIdx.x; e < rSize * colsTotal; e += blockDim.x) {
    const int64_t j = e / colsTotal, col = e - j * colsTotal;
    // block column of `col`
    int64_t c = 0;
    while (cre[c] - rectRowBegin <= col) c++;
    const int64_t cStart = cre[c - 1] - rectRowBegin;
    GP<T> dst = data + rOffset + sk.spanOffsetInLump[toSpan[c]] + j * dstStride + (col - cStart);
    *dst += temp[(rBegin + j) * srcRectWidth + col];
  }
}

}  // namespace hipk
}  // namespace BaSpaCho
