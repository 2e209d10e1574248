// Structure study of the rank-256 tile update (the bulk kernel): one 64x64 tile per workgroup,
// operands from a row-major panel P[rows][K] in global memory, staged through LDS.
//   V0: the library's loop -- per chunk of 32: loads, barrier, LDS stores, barrier, 32 MFMAs/wave
//   V1: V0 + register prefetch of the next chunk
//   V2: chunks of 16, two LDS buffers, ONE barrier per chunk; the next chunk's LDS stores are
//       issued between the MFMAs of the current one
// hipcc -O3 --offload-arch=gfx950 tools/tile_update_probe.hip -o /tmp/tp && /tmp/tp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const double* GPc;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

__global__ void fillRandom(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}
template <int V>
__global__ __launch_bounds__(256) void tile(const double* Pg, double* out, int K, int lda, int tilesPerRow) {
  extern __shared__ double lds[];
  GPc P = (GPc)Pg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int rowTile = 64 * (blockIdx.x % tilesPerRow), colTile = 64 * ((blockIdx.x / tilesPerRow) % tilesPerRow);
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  d4 c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00;
  if (V <= 1 || V == 6 || V == 7) {
    constexpr int KC = 32, LD = KC + 2;
    double* As = lds; double* Bs = lds + 64 * LD;
    const int sk = tid % KC, sr = tid / KC;
    double va[8], vb[8];
    auto fetch = [&](int kBase) {
#pragma unroll
      for (int it = 0; it < 8; it++) va[it] = P[(size_t)(rowTile + sr + 8 * it) * lda + kBase + sk];
#pragma unroll
      for (int it = 0; it < 8; it++) vb[it] = P[(size_t)(colTile + sr + 8 * it) * lda + kBase + sk];
    };
    if (V == 1) fetch(0);
    if (V == 6) { for (int i = tid; i < 2 * 64 * LD; i += 256) lds[i] = i * 1e-4; __syncthreads(); }
    for (int kBase = 0; kBase < K; kBase += KC) {
      if (V == 0 || V == 7) fetch(kBase);
      if (kBase > 0 && V != 7) __syncthreads();
      if (V != 6) {
#pragma unroll
        for (int it = 0; it < 8; it++) As[(sr + 8 * it) * LD + sk] = va[it];
#pragma unroll
        for (int it = 0; it < 8; it++) Bs[(sr + 8 * it) * LD + sk] = vb[it];
      }
      if (V != 7) __syncthreads();
      if (V == 1 && kBase + KC < K) fetch(kBase + KC);
      for (int k0 = 0; k0 < KC; k0 += 4) {
        const double a0 = As[(wr + li) * LD + k0 + lk], a1 = As[(wr + 16 + li) * LD + k0 + lk];
        const double b0 = Bs[(wc + li) * LD + k0 + lk], b1 = Bs[(wc + 16 + li) * LD + k0 + lk];
        c00 = MFMA(a0, b0, c00); c01 = MFMA(a0, b1, c01); c10 = MFMA(a1, b0, c10); c11 = MFMA(a1, b1, c11);
      }
    }
  } else if (V == 8 || V == 9) {
    // operands go straight from global memory to LDS (global_load_lds_dwordx4: lane t of a wave
    // lands at base + 16 t bytes, so a wave instruction fills 4 rows of 32 doubles); no padding,
    // the pair index is XOR-swizzled with the row instead, which keeps the MFMA operand reads at
    // two lanes per 8-byte slot.  V8: one buffer, two barriers per chunk.  V9: two buffers, the
    // next chunk is requested before the current one is multiplied, one barrier per chunk.
    typedef __attribute__((address_space(1))) const void* GV;
    typedef __attribute__((address_space(3))) void* LV;
    constexpr int KC = 32, NBUF = V == 9 ? 2 : 1, OP = 64 * KC;
    const int w = wave, t = lane;
    GPc srcA[4], srcB[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int r = 16 * it + 4 * w + (t >> 4);
      const int pair = (t & 15) ^ (r & 15);
      srcA[it] = P + (size_t)(rowTile + r) * lda + 2 * pair;
      srcB[it] = P + (size_t)(colTile + r) * lda + 2 * pair;
    }
    auto request = [&](int kBase, int buf) {
      double* As = lds + buf * 2 * OP; double* Bs = As + OP;
#pragma unroll
      for (int it = 0; it < 4; it++) {
        __builtin_amdgcn_global_load_lds((GV)(srcA[it] + kBase), (LV)(As + (16 * it + 4 * w) * KC), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((GV)(srcB[it] + kBase), (LV)(Bs + (16 * it + 4 * w) * KC), 16, 0, 0);
      }
    };
    const int ra0 = wr + li, ra1 = wr + 16 + li, rb0 = wc + li, rb1 = wc + 16 + li;
    auto multiply = [&](int buf) {
      const double* As = lds + buf * 2 * OP; const double* Bs = As + OP;
#pragma unroll
      for (int k0 = 0; k0 < KC; k0 += 4) {
        const int k = k0 + lk, pr = k >> 1, lo = k & 1;
        const double a0 = As[ra0 * KC + 2 * (pr ^ (ra0 & 15)) + lo], a1 = As[ra1 * KC + 2 * (pr ^ (ra1 & 15)) + lo];
        const double b0 = Bs[rb0 * KC + 2 * (pr ^ (rb0 & 15)) + lo], b1 = Bs[rb1 * KC + 2 * (pr ^ (rb1 & 15)) + lo];
        c00 = MFMA(a0, b0, c00); c01 = MFMA(a0, b1, c01); c10 = MFMA(a1, b0, c10); c11 = MFMA(a1, b1, c11);
      }
    };
    if (V == 8) {
      for (int kBase = 0; kBase < K; kBase += KC) {
        if (kBase > 0) __syncthreads();
        request(kBase, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        multiply(0);
      }
    } else {
      request(0, 0);
      int buf = 0;
      for (int kBase = 0; kBase < K; kBase += KC, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk kBase has landed for everybody, the other buffer is free
        if (kBase + KC < K) request(kBase + KC, buf ^ 1);
        multiply(buf);
      }
    }
  } else if (V == 3 || V == 4) {
    constexpr int KC = V == 3 ? 64 : 32, LD = KC + 2;
    typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
    typedef __attribute__((address_space(1))) const d2* GP2;
    double* As = lds; double* Bs = lds + 64 * LD;
    constexpr int TPR = KC / 2, RPP = 256 / TPR, NP = 64 / RPP;   // threads per row, rows per pass, passes
    const int sk = 2 * (tid % TPR), sr = tid / TPR;
    d2 va[NP], vb[NP];
    for (int kBase = 0; kBase < K; kBase += KC) {
#pragma unroll
      for (int it = 0; it < NP; it++) va[it] = *(GP2)(P + (size_t)(rowTile + sr + RPP * it) * lda + kBase + sk);
#pragma unroll
      for (int it = 0; it < NP; it++) vb[it] = *(GP2)(P + (size_t)(colTile + sr + RPP * it) * lda + kBase + sk);
      if (kBase > 0) __syncthreads();
#pragma unroll
      for (int it = 0; it < NP; it++) *(double2*)&As[(sr + RPP * it) * LD + sk] = double2{va[it][0], va[it][1]};
#pragma unroll
      for (int it = 0; it < NP; it++) *(double2*)&Bs[(sr + RPP * it) * LD + sk] = double2{vb[it][0], vb[it][1]};
      __syncthreads();
      for (int k0 = 0; k0 < KC; k0 += 4) {
        const double a0 = As[(wr + li) * LD + k0 + lk], a1 = As[(wr + 16 + li) * LD + k0 + lk];
        const double b0 = Bs[(wc + li) * LD + k0 + lk], b1 = Bs[(wc + 16 + li) * LD + k0 + lk];
        c00 = MFMA(a0, b0, c00); c01 = MFMA(a0, b1, c01); c10 = MFMA(a1, b0, c10); c11 = MFMA(a1, b1, c11);
      }
    }
  } else {
    constexpr int KC = 16, LD = KC + 2, BUF = 2 * 64 * LD;
    const int sk = tid % KC, sr = tid / KC;  // 16 rows per pass, 4 passes per operand
    double va[4], vb[4];
    auto fetch = [&](int kBase) {
#pragma unroll
      for (int it = 0; it < 4; it++) va[it] = P[(size_t)(rowTile + sr + 16 * it) * lda + kBase + sk];
#pragma unroll
      for (int it = 0; it < 4; it++) vb[it] = P[(size_t)(colTile + sr + 16 * it) * lda + kBase + sk];
    };
    auto stage = [&](int buf) {
      double* As = lds + buf * BUF; double* Bs = As + 64 * LD;
#pragma unroll
      for (int it = 0; it < 4; it++) As[(sr + 16 * it) * LD + sk] = va[it];
#pragma unroll
      for (int it = 0; it < 4; it++) Bs[(sr + 16 * it) * LD + sk] = vb[it];
    };
    fetch(0);
    stage(0);
    if (KC < K) fetch(KC);
    __syncthreads();
    int buf = 0;
    for (int kBase = 0; kBase < K; kBase += KC, buf ^= 1) {
      const double* As = lds + buf * BUF; const double* Bs = As + 64 * LD;
      // first half of the chunk's MFMAs
#pragma unroll
      for (int k0 = 0; k0 < 8; k0 += 4) {
        const double a0 = As[(wr + li) * LD + k0 + lk], a1 = As[(wr + 16 + li) * LD + k0 + lk];
        const double b0 = Bs[(wc + li) * LD + k0 + lk], b1 = Bs[(wc + 16 + li) * LD + k0 + lk];
        c00 = MFMA(a0, b0, c00); c01 = MFMA(a0, b1, c01); c10 = MFMA(a1, b0, c10); c11 = MFMA(a1, b1, c11);
      }
      // the next chunk goes into the other buffer (free since the last barrier), the one after
      // that is requested
      if (kBase + KC < K) stage(buf ^ 1);
      if (kBase + 2 * KC < K) fetch(kBase + 2 * KC);
#pragma unroll
      for (int k0 = 8; k0 < 16; k0 += 4) {
        const double a0 = As[(wr + li) * LD + k0 + lk], a1 = As[(wr + 16 + li) * LD + k0 + lk];
        const double b0 = Bs[(wc + li) * LD + k0 + lk], b1 = Bs[(wc + 16 + li) * LD + k0 + lk];
        c00 = MFMA(a0, b0, c00); c01 = MFMA(a0, b1, c01); c10 = MFMA(a1, b0, c10); c11 = MFMA(a1, b1, c11);
      }
      __syncthreads();
    }
  }
  if (out == (double*)2) {
    // epilogue through LDS: the accumulators are laid out as the 64x64 tile in LDS (free now),
    // then every wave read-modify-writes whole rows: 64 lanes x 8 B = 512 contiguous bytes per
    // instruction instead of four 128-byte pieces of four rows
    double* T = const_cast<double*>(Pg) + (size_t)rowTile * lda + 256 + colTile;
    const d4* accs[4] = {&c00, &c01, &c10, &c11};
    double old[16];
#pragma unroll
    for (int j = 0; j < 16; j++) old[j] = T[(size_t)(16 * wave + j) * lda + lane];   // issued first
    __syncthreads();  // everybody is done with the operands in LDS
    constexpr int LDC = 65;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        lds[(wr + (t >> 1) * 16 + lk + 4 * r) * LDC + wc + (t & 1) * 16 + li] = (*accs[t])[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++)
      T[(size_t)(16 * wave + j) * lda + lane] = old[j] - lds[(16 * wave + j) * LDC + lane] * 1e-9;
  } else if (out == (double*)1) {  // epilogue with no-return atomics (no read of the target)
    double* T = const_cast<double*>(Pg) + (size_t)rowTile * lda + 256 + colTile;
    const d4* accs[4] = {&c00, &c01, &c10, &c11};
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        unsafeAtomicAdd(&T[(size_t)(wr + (t >> 1) * 16 + lk + 4 * r) * lda + wc + (t & 1) * 16 + li], -(*accs[t])[r] * 1e-9);
  } else if (out) {
    out[(size_t)blockIdx.x * 256 + tid] = c00[0] + c01[1] + c10[2] + c11[3];
  } else {  // epilogue of the real kernel: read-modify-write of the 64x64 target tile (in P itself)
    double* T = const_cast<double*>(Pg) + (size_t)rowTile * lda + 256 + colTile;
    const d4* accs[4] = {&c00, &c01, &c10, &c11};
    double old[16];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        old[4 * t + r] = T[(size_t)(wr + (t >> 1) * 16 + lk + 4 * r) * lda + wc + (t & 1) * 16 + li];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        T[(size_t)(wr + (t >> 1) * 16 + lk + 4 * r) * lda + wc + (t & 1) * 16 + li] = old[4 * t + r] - (*accs[t])[r] * 1e-9;
  }
}

template <int V>
void run(const char* name, int wgPerCu, const double* P, double* out, int K, int lda, int tilesPerRow, int nTiles) {
  const size_t base = V == 3 ? 2 * 64 * 66 * 8 : V == 9 ? 2 * 2 * 64 * 32 * 8 : V == 2 ? 2 * 2 * 64 * 18 * 8 : 2 * 64 * 34 * 8;
  const size_t want = wgPerCu == 4 ? 40000 : wgPerCu == 3 ? 41000 : 80000;  // >= 64 x 65 doubles
  const size_t smem = want > base ? want : base;
  hipFuncSetAttribute((const void*)tile<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  tile<V><<<nTiles, 256, smem>>>(P, out, K, lda, tilesPerRow);
  hipEventRecord(e0);
  tile<V><<<nTiles, 256, smem>>>(P, out, K, lda, tilesPerRow);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s WG/CU %d : %6.1f TF/s  (%.3f ms)\n", name, wgPerCu, double(nTiles) * 64 * 64 * K * 2 / (ms * 1e-3) / 1e12, ms);
}
int main() {
  const int rows = 7680, K = 256, lda = 7839, tilesPerRow = rows / 64, nTiles = 7200;
  double *P, *out;
  hipMalloc(&P, (size_t)rows * lda * 8); hipMemset(P, 0, (size_t)rows * lda * 8);
  hipMalloc(&out, (size_t)72000 * 256 * 8);
  fillRandom<<<1024, 256>>>(P, (size_t)rows * lda);
  const int big = 36000;
  // correctness of the direct-to-LDS variants: same products in the same order as V0
  {
    double* h0 = new double[7200 * 256]; double* h1 = new double[7200 * 256];
    run<0>("V0", 3, P, out, K, lda, tilesPerRow, 7200); hipMemcpy(h0, out, 7200 * 256 * 8, hipMemcpyDeviceToHost);
    run<8>("V8", 3, P, out, K, lda, tilesPerRow, 7200); hipMemcpy(h1, out, 7200 * 256 * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < 7200 * 256; i++) bad += h0[i] != h1[i];
    printf("V8 vs V0 mismatches: %zu\n", bad);
    run<9>("V9", 2, P, out, K, lda, tilesPerRow, 7200); hipMemcpy(h1, out, 7200 * 256 * 8, hipMemcpyDeviceToHost);
    bad = 0; for (size_t i = 0; i < 7200 * 256; i++) bad += h0[i] != h1[i];
    printf("V9 vs V0 mismatches: %zu\n", bad);
  }
  for (int rep = 0; rep < 2; rep++) {
    run<8>("V8 no epilogue", 3, P, out, 256, lda, tilesPerRow, big);
    run<8>("V8 RMW from accumulator layout", 3, P, nullptr, 256, lda, tilesPerRow, big);
    run<8>("V8 RMW of whole rows via LDS", 3, P, (double*)2, 256, lda, tilesPerRow, big);
    run<8>("V8 atomics", 3, P, (double*)1, 256, lda, tilesPerRow, big);
  }
  return 0;
}
