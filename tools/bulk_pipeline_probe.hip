// Round-3 study of the bulk rank-256 tile update at the launch sizes the library actually uses
// (2-4 rounds of workgroups per launch, not the 47 rounds of tools/tile_update_probe.hip).
//
// Hypothesis: in a launch of ~2 rounds all 768 resident workgroups start together, so their load
// and multiply phases stay aligned -- the matrix pipe idles while every workgroup of a CU waits for
// its chunk, and the memory system idles while they all multiply.  A long launch de-synchronises by
// itself (which is why the long-launch harness saw no gain from double buffering); a short one does
// not.  Variants, all with the read-modify-write epilogue of the library kernel:
//   S1  one LDS buffer, chunk of 32 (the library's updateTileBulk loop)
//   D32 two buffers, chunks of 32 (64 KB: 2 workgroups per CU)
//   D16 two buffers, chunks of 16 (32 KB: 3-4 workgroups per CU)
//   T16 three buffers, chunks of 16 (48 KB: 3 workgroups per CU), vmcnt(N) waits
//   P*  the same, PERSISTENT: grid = resident workgroups, tiles pulled through an atomic ticket,
//       the first chunks of the next tile requested before the epilogue of the current one
// hipcc -O3 --offload-arch=gfx950 tools/bulk_pipeline_probe.hip -o tools/bulk_pipeline_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const double* GPc;
typedef __attribute__((address_space(1))) double* GPm;
typedef __attribute__((address_space(1))) const void* GV;
typedef __attribute__((address_space(3))) void* LV;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

struct Task {
  long long srcOff, tgtOff;
  int rowTile, colTile, K, pad;
};

__global__ void fillRandom(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

// LDS layout of one operand chunk [64 rows][KC]: 256-byte lines (32 doubles = 32/KC rows), the
// 16-byte slot inside a line XOR-ed with the line index
template <int KC>
struct Lay {
  static constexpr int RPL = 32 / KC;  // rows per line
  static __device__ __forceinline__ int at(int r, int k) {
    const int line = r / RPL, e = (r % RPL) * KC + k;
    return line * 32 + 2 * ((e >> 1) ^ (line & 15)) + (e & 1);
  }
};

// EPI: 0 read-modify-write after the K loop (the library), 1 store only (no old values: bound),
//      2 no-return atomic adds, 3 old values requested BEFORE the last chunk's multiplies
template <int KC, int NBUF, bool PERSIST, int EPI = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void bulk(const Task* tasks, int nTasks, double* data, int lda,
                                            int* ticket) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int OP = 64 * KC;             // doubles per operand chunk
  constexpr int LINES = OP / 32;          // 256-byte lines per operand chunk
  constexpr int IPW = LINES / 16;         // wave instructions per operand, chunk and wave
  constexpr int RPL = 32 / KC;
  __shared__ int nextTask;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  GPm D = (GPm)data;

  int cur = PERSIST ? -1 : (int)blockIdx.x;
  if (PERSIST) {
    if (tid == 0) nextTask = atomicAdd(ticket, 1);
    __syncthreads();
    cur = __builtin_amdgcn_readfirstlane(nextTask);
    __syncthreads();
  }
  // lane's source of wave instruction `it`: row r0 + 16 RPL it (one pointer + a uniform stride)
  GPc srcA, srcB;
  const long long itStride = (long long)16 * RPL * lda;
  auto setup = [&](const Task& t) {
    const int line = 4 * wave + (lane >> 4);
    const int slot = (lane & 15) ^ (line & 15);
    const int e = 2 * slot, r = line * RPL + e / KC, k = e % KC;
    srcA = (GPc)D + t.srcOff + (long long)(t.rowTile + r) * lda + k;
    srcB = (GPc)D + t.srcOff + (long long)(t.colTile + r) * lda + k;
  };
  auto request = [&](int kBase, int buf) {
    double* As = lds + buf * 2 * OP;
    double* Bs = As + OP;
#pragma unroll
    for (int it = 0; it < IPW; it++) {
      __builtin_amdgcn_global_load_lds((GV)(srcA + it * itStride + kBase), (LV)(As + 128 * (4 * it + wave)), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((GV)(srcB + it * itStride + kBase), (LV)(Bs + 128 * (4 * it + wave)), 16, 0, 0);
    }
  };

  if (cur >= nTasks) return;
  Task t = tasks[cur];
  setup(t);
#pragma unroll
  for (int p = 0; p < (NBUF > 1 ? NBUF - 1 : 0); p++) request(p * KC, p);
  for (;;) {
    d4 c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00;
    const int K = t.K, nChunks = K / KC;
    int nxt = -1;
    if (PERSIST && tid == 0) nextTask = atomicAdd(ticket, 1);
    bool first = true;
    double old[16];
    for (int c = 0; c < nChunks; c++) {
      const int buf = NBUF > 1 ? c % NBUF : 0;
      if (NBUF == 1) {
        if (c > 0 && EPI != 6) __syncthreads();
        if ((EPI != 4 && EPI != 6) || c == 0) request(c * KC, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (EPI != 6 || c == 0) __syncthreads();
        if (EPI == 3 && c == nChunks - 1) {
          GPm tg = D + t.tgtOff;
#pragma unroll
          for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
              const int qr = t.rowTile + wr + (q >> 1) * 16 + lk + 4 * reg, qc = t.colTile + wc + (q & 1) * 16 + li;
              old[4 * q + reg] = tg[(long long)qr * lda + qc];
            }
          }
        }
      } else {
        // chunk c has landed when at most (NBUF - 2) chunks' worth of loads are still in flight;
        // the first wait of a tile (stores of the previous epilogue may be in flight) and the
        // last chunks wait for everything
        if (NBUF == 2 || first || c + NBUF - 2 >= nChunks) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * 2 * IPW) : "memory");
        }
        first = false;
        __syncthreads();
        if (c + NBUF - 1 < nChunks) request((c + NBUF - 1) * KC, (c + NBUF - 1) % NBUF);
      }
      const double* As = lds + buf * 2 * OP;
      const double* Bs = As + OP;
#pragma unroll
      for (int k0 = 0; k0 < KC; k0 += 4) {
        // (addresses recomputed per step from two opaque row indices: hoisted, the 32 swizzled
        //  offsets of a chunk would cost 32 registers)
        int ra = wr + li, rb = wc + li;
        asm volatile("" : "+v"(ra), "+v"(rb));
        const double a0 = As[Lay<KC>::at(ra, k0 + lk)], a1 = As[Lay<KC>::at(ra + 16, k0 + lk)];
        const double b0 = Bs[Lay<KC>::at(rb, k0 + lk)], b1 = Bs[Lay<KC>::at(rb + 16, k0 + lk)];
        if (EPI == 5) {
          c00[0] += a0 + b0; c01[0] += a1 + b1;
        } else {
          c00 = MFMA(a0, b0, c00); c01 = MFMA(a0, b1, c01); c10 = MFMA(a1, b0, c10); c11 = MFMA(a1, b1, c11);
        }
      }
    }
    // epilogue: read-modify-write of the 64x64 target from accumulator layout
    GPm tgt = D + t.tgtOff;
    Task tn;
    bool more = false;
    if (PERSIST) {
      __syncthreads();  // every wave is out of the last chunk: buffers free, nextTask visible
      nxt = __builtin_amdgcn_readfirstlane(nextTask);
      more = nxt < nTasks;
      if (more) {
        tn = tasks[nxt];
        setup(tn);
        if (NBUF > 1) {
#pragma unroll
          for (int p = 0; p < NBUF - 1; p++) request(p * KC, p);
        }
      }
    }
    const d4* accs[4] = {&c00, &c01, &c10, &c11};
    if (EPI == 2) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int qr = t.rowTile + wr + (q >> 1) * 16 + lk + 4 * reg, qc = t.colTile + wc + (q & 1) * 16 + li;
          unsafeAtomicAdd(data + t.tgtOff + (long long)qr * lda + qc, -(*accs[q])[reg] * 1e-9);
        }
      }
    } else {
      if (EPI == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int qr = t.rowTile + wr + (q >> 1) * 16 + lk + 4 * reg, qc = t.colTile + wc + (q & 1) * 16 + li;
            old[4 * q + reg] = tgt[(long long)qr * lda + qc];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int qr = t.rowTile + wr + (q >> 1) * 16 + lk + 4 * reg, qc = t.colTile + wc + (q & 1) * 16 + li;
          tgt[(long long)qr * lda + qc] = ((EPI == 1 || EPI >= 4) ? 0.0 : old[4 * q + reg]) - (*accs[q])[reg] * 1e-9;
        }
      }
    }
    if (!PERSIST || !more) break;
    t = tn;
    cur = nxt;
    __syncthreads();  // nextTask may be rewritten
  }
}

struct Result { double tf; float ms; };
// W2: one workgroup = 64 rows x 128 columns (two column tiles of one row tile): the row operand is
// loaded once for both, 48 KB of LDS (2 workgroups per CU beside a chain workgroup, or 3 without)
template <int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void bulkWide(
    const Task* tasks, int nTasks, double* data, int lda) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int OPA = 64 * KC, OPB = 128 * KC;
  constexpr int RPL = 32 / KC;
  constexpr int IPWA = (OPA / 32) / 16, IPWB = (OPB / 32) / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
  GPm D = (GPm)data;
  const int cur = blockIdx.x;
  if (2 * cur >= nTasks) return;
  const Task t = tasks[2 * cur];  // (column tiles 2 cur and 2 cur + 1 of the row-major list: neighbours)
  const long long itStride = (long long)16 * RPL * lda;
  const int line = 4 * wave + (lane >> 4);
  const int slot = (lane & 15) ^ (line & 15);
  const int e = 2 * slot, r = line * RPL + e / KC, k = e % KC;
  GPc srcA = (GPc)D + t.srcOff + (long long)(t.rowTile + r) * lda + k;
  GPc srcB = (GPc)D + t.srcOff + (long long)(t.colTile + r) * lda + k;
  double* As = lds;
  double* Bs = lds + OPA;
  d4 c[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) c[i][j] = d4{0, 0, 0, 0};
  const int nChunks = t.K / KC;
  for (int ch = 0; ch < nChunks; ch++) {
    if (ch > 0) __syncthreads();
#pragma unroll
    for (int it = 0; it < IPWA; it++)
      __builtin_amdgcn_global_load_lds((GV)(srcA + it * itStride + ch * KC), (LV)(As + 128 * (4 * it + wave)), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < IPWB; it++)
      __builtin_amdgcn_global_load_lds((GV)(srcB + it * itStride + ch * KC), (LV)(Bs + 128 * (4 * it + wave)), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
      int ra = wr + li, rb = wc + li;
      asm volatile("" : "+v"(ra), "+v"(rb));
      const double a0 = As[Lay<KC>::at(ra, k0 + lk)], a1 = As[Lay<KC>::at(ra + 16, k0 + lk)];
      double b[4];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = Bs[Lay<KC>::at(rb + 16 * j, k0 + lk)];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c[0][j] = MFMA(a0, b[j], c[0][j]);
        c[1][j] = MFMA(a1, b[j], c[1][j]);
      }
    }
  }
  GPm tgt = D + t.tgtOff;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    double old[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int qr = t.rowTile + wr + half * 16 + lk + 4 * reg, qc = t.colTile + wc + q * 16 + li;
        old[4 * q + reg] = tgt[(long long)qr * lda + qc];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int qr = t.rowTile + wr + half * 16 + lk + 4 * reg, qc = t.colTile + wc + q * 16 + li;
        tgt[(long long)qr * lda + qc] = old[4 * q + reg] - c[half][q][reg] * 1e-9;
      }
    }
  }
}

template <int KC>
Result runWide(const Task* dTasks, int nTasks, double* data, int lda, int wgPerCu, int K) {
  const size_t need = (size_t)(64 + 128) * KC * 8;
  size_t smem = (160 * 1024) / wgPerCu - 1024;
  if (smem < need) smem = need;
  hipFuncSetAttribute((const void*)bulkWide<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    hipEventRecord(e0);
    bulkWide<KC><<<nTasks / 2, 256, smem>>>(dTasks, nTasks, data, lda);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return {double(nTasks / 2 * 2) * 64 * 64 * K * 2 / (best * 1e-3) / 1e12, best};
}


// Q: one workgroup of 512 threads = 128 rows x 128 columns (2 x 2 tiles): 16 flops per operand byte
// instead of 8; wave (wy, wx) owns 32 rows x 64 columns
template <int KC, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void bulkQuad(
    const Task* tasks, int nQuads, double* data, int lda) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int OP = 128 * KC;
  constexpr int RPL = 32 / KC;
  constexpr int IPW = (OP / 32) / 32;  // wave instructions per operand, chunk and wave (8 waves x 4 lines)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
  GPm D = (GPm)data;
  const Task t = tasks[blockIdx.x];
  const long long itStride = (long long)32 * RPL * lda;
  const int line = 4 * wave + (lane >> 4);
  const int slot = (lane & 15) ^ (line & 15);
  const int e = 2 * slot, r = line * RPL + e / KC, k = e % KC;
  GPc srcA = (GPc)D + t.srcOff + (long long)(t.rowTile + r) * lda + k;
  GPc srcB = (GPc)D + t.srcOff + (long long)(t.colTile + r) * lda + k;
  double* As = lds;
  double* Bs = lds + OP;
  d4 c[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) c[i][j] = d4{0, 0, 0, 0};
  const int nChunks = t.K / KC;
  GPm tgt = D + t.tgtOff;
  for (int ch = 0; ch < nChunks; ch++) {
    if (ch > 0) __syncthreads();
    if (EPI != 4 || ch == 0)
#pragma unroll
    for (int it = 0; it < IPW; it++) {
      __builtin_amdgcn_global_load_lds((GV)(srcA + it * itStride + ch * KC), (LV)(As + 128 * (8 * it + wave)), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((GV)(srcB + it * itStride + ch * KC), (LV)(Bs + 128 * (8 * it + wave)), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
      int ra = wr + li, rb = wc + li;
      asm volatile("" : "+v"(ra), "+v"(rb));
      const double a0 = As[Lay<KC>::at(ra, k0 + lk)], a1 = As[Lay<KC>::at(ra + 16, k0 + lk)];
      double b[4];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = Bs[Lay<KC>::at(rb + 16 * j, k0 + lk)];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (EPI == 5) {
          c[0][j][0] += a0 + b[j]; c[1][j][0] += a1;
        } else {
          c[0][j] = MFMA(a0, b[j], c[0][j]);
          c[1][j] = MFMA(a1, b[j], c[1][j]);
        }
      }
    }
  }
#pragma unroll
  for (int half = 0; half < 2; half++) {
    double old[16];
    if (EPI == 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int qr = t.rowTile + wr + half * 16 + lk + 4 * reg, qc = t.colTile + wc + q * 16 + li;
          old[4 * q + reg] = tgt[(long long)qr * lda + qc];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int qr = t.rowTile + wr + half * 16 + lk + 4 * reg, qc = t.colTile + wc + q * 16 + li;
        if (EPI == 2) {
          unsafeAtomicAdd(data + t.tgtOff + (long long)qr * lda + qc, -c[half][q][reg] * 1e-9);
        } else {
          tgt[(long long)qr * lda + qc] = ((EPI == 1 || EPI >= 4) ? 0.0 : old[4 * q + reg]) - c[half][q][reg] * 1e-9;
        }
      }
    }
  }
}

template <int KC, int EPI>
Result runQuad(const Task* dQuads, int nQuads, double* data, int lda, int wgPerCu, int K) {
  const size_t need = (size_t)(128 + 128) * KC * 8;
  size_t smem = (160 * 1024) / wgPerCu - 1024;
  if (smem < need) smem = need;
  hipFuncSetAttribute((const void*)bulkQuad<KC, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    hipEventRecord(e0);
    bulkQuad<KC, EPI><<<nQuads, 512, smem>>>(dQuads, nQuads, data, lda);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return {double(nQuads) * 128 * 128 * K * 2 / (best * 1e-3) / 1e12, best};
}


template <int KC, int NBUF, bool PERSIST, int EPI = 0>
Result run(const Task* dTasks, int nTasks, double* data, int lda, int* ticket, int wgPerCu, int K) {
  const size_t need = (size_t)NBUF * 2 * 64 * KC * 8;
  // pad dynamic LDS so that exactly wgPerCu workgroups fit the 160 KB of a CU
  size_t smem = (160 * 1024) / wgPerCu - 1024;
  if (smem < need) smem = need;
  hipFuncSetAttribute((const void*)bulk<KC, NBUF, PERSIST, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = PERSIST ? (256 * wgPerCu < nTasks ? 256 * wgPerCu : nTasks) : nTasks;
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    hipMemsetAsync(ticket, 0, 4);
    hipEventRecord(e0);
    bulk<KC, NBUF, PERSIST, EPI><<<grid, 256, smem>>>(dTasks, nTasks, data, lda, ticket);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return {double(nTasks) * 64 * 64 * K * 2 / (best * 1e-3) / 1e12, best};
}

int main(int argc, char** argv) {
  const int n = 7839, lda = n, K = 256;
  double* data;
  hipMalloc(&data, (size_t)n * lda * 8);
  fillRandom<<<1024, 256>>>(data, (size_t)n * lda);
  int* ticket; hipMalloc(&ticket, 4);
  // a lookahead unit as the library builds it: source block b (256 columns), target column block
  // c > b: row tiles below, 4 column tiles each, row-tile-major; the launch = units of one source
  // block to the next column blocks, cut to nTiles; XCD-contiguous permutation
  auto makeTasks = [&](int nTiles) {
    std::vector<Task> v;
    const int b = 2;
    for (int c = b + 2; (int)v.size() < nTiles && c < 30; c++) {
      const int col0 = 256 * c;
      for (int rt = col0; rt + 64 <= n - 31 && (int)v.size() < nTiles; rt += 64) {
        for (int ct = 0; ct < 4 && (int)v.size() < nTiles; ct++) {
          if (col0 + 64 * ct > rt) continue;
          Task t;
          // rows are indexed from the first row below the source block
          const int base = 256 * (b + 1);
          t.srcOff = (long long)base * lda + 256 * b;
          t.tgtOff = (long long)base * lda + base;
          t.rowTile = rt - base; t.colTile = col0 + 64 * ct - base; t.K = K; t.pad = 0;
          v.push_back(t);
        }
      }
    }
    // XCD-contiguous permutation (workgroup i lands on XCD i % 8)
    std::vector<Task> p(v.size());
    const int nn = (int)v.size(), bs = nn >> 3, ex = nn & 7;
    for (int i = 0; i < nn; i++) {
      const int x = i & 7;
      int idx = x * bs + (x < ex ? x : ex) + (i >> 3);
      if (idx >= nn) idx = i;
      p[i] = v[idx];
    }
    return p;
  };
  // pairs of neighbouring column tiles for the 64 x 128 variant: full rows only, XCD-contiguous by pair
  auto makeTasksWide = [&](int nTiles) {
    std::vector<Task> v;
    const int b = 2;
    for (int c = b + 2; (int)v.size() < nTiles && c < 30; c++) {
      const int col0 = 256 * c;
      for (int rt = col0 + 192; rt + 64 <= n - 31 && (int)v.size() + 4 <= nTiles; rt += 64) {
        for (int ct = 0; ct < 4; ct++) {
          Task t;
          const int base = 256 * (b + 1);
          t.srcOff = (long long)base * lda + 256 * b;
          t.tgtOff = (long long)base * lda + base;
          t.rowTile = rt - base; t.colTile = col0 + 64 * ct - base; t.K = K; t.pad = 0;
          v.push_back(t);
        }
      }
    }
    const int np = (int)v.size() / 2, bs = np >> 3, ex = np & 7;
    std::vector<Task> p(v.size());
    for (int i = 0; i < np; i++) {
      const int x = i & 7;
      int idx = x * bs + (x < ex ? x : ex) + (i >> 3);
      if (idx >= np) idx = i;
      p[2 * i] = v[2 * idx];
      p[2 * i + 1] = v[2 * idx + 1];
    }
    return p;
  };
  if (argc > 1 && argv[1][0] == 'q') {
    // quads (2 x 2 tiles) of full rows, XCD-contiguous; the same tiles as a plain list for S1
    printf("%-10s%9s%9s%9s%9s%9s%9s%9s%9s\n", "tiles", "S1@3", "S1st@3", "S1nold@3", "S1nomm@3", "Q32@2", "Q32st@2", "S1nobar@3", "S1nobar@4");
    for (int nt : {768, 1536, 3072, 6144}) {
      std::vector<Task> quads, tiles;
      const int b = 2, base = 256 * (b + 1);
      for (int c = b + 2; (int)tiles.size() < nt && c < 30; c++) {
        const int col0 = 256 * c;
        for (int rt = col0 + 256; rt + 128 <= n - 31 && (int)tiles.size() + 8 <= nt; rt += 128) {
          for (int cq = 0; cq < 2; cq++) {
            Task t;
            t.srcOff = (long long)base * lda + 256 * b;
            t.tgtOff = (long long)base * lda + base;
            t.rowTile = rt - base; t.colTile = col0 + 128 * cq - base; t.K = K; t.pad = 0;
            quads.push_back(t);
            for (int dr = 0; dr < 2; dr++) for (int dc = 0; dc < 2; dc++) {
              Task u = t; u.rowTile += 64 * dr; u.colTile += 64 * dc; tiles.push_back(u);
            }
          }
        }
      }
      auto xcd = [](std::vector<Task>& v) {
        std::vector<Task> p(v.size());
        const int nn = (int)v.size(), bs = nn >> 3, ex = nn & 7;
        for (int i = 0; i < nn; i++) {
          const int x = i & 7;
          int idx = x * bs + (x < ex ? x : ex) + (i >> 3);
          if (idx >= nn) idx = i;
          p[i] = v[idx];
        }
        v = p;
      };
      xcd(quads); xcd(tiles);
      Task *dQ, *dT;
      hipMalloc(&dQ, quads.size() * sizeof(Task)); hipMalloc(&dT, tiles.size() * sizeof(Task));
      hipMemcpy(dQ, quads.data(), quads.size() * sizeof(Task), hipMemcpyHostToDevice);
      hipMemcpy(dT, tiles.data(), tiles.size() * sizeof(Task), hipMemcpyHostToDevice);
      const int nT = (int)tiles.size(), nQ = (int)quads.size();
      Result r[8];
      r[0] = run<32, 1, false, 0>(dT, nT, data, lda, ticket, 3, K);
      r[1] = run<32, 1, false, 1>(dT, nT, data, lda, ticket, 3, K);
      r[2] = run<32, 1, false, 4>(dT, nT, data, lda, ticket, 3, K);
      r[3] = run<32, 1, false, 5>(dT, nT, data, lda, ticket, 3, K);
      r[4] = runQuad<32, 0>(dQ, nQ, data, lda, 2, K);
      r[5] = runQuad<32, 1>(dQ, nQ, data, lda, 2, K);
      r[6] = run<32, 1, false, 6>(dT, nT, data, lda, ticket, 3, K);
      r[7] = run<32, 1, false, 6>(dT, nT, data, lda, ticket, 4, K);
      printf("%-10d", nT);
      for (auto& x : r) printf("%9.1f", x.tf);
      printf("   TF/s\n%-10s", "");
      for (auto& x : r) printf("%9.1f", x.ms * 1e3);
      printf("   us\n");
      hipFree(dQ); hipFree(dT);
    }
    return 0;
  }
  printf("%-10s%9s%9s%9s%9s\n", "tiles", "S1@3", "W2@3", "W2@2", "W2k16@3");
  for (int nt : {768, 1536, 2304, 3072, 6144}) {
    auto host = makeTasksWide(nt);
    const int nTasks = (int)host.size();
    Task* dT; hipMalloc(&dT, nTasks * sizeof(Task));
    hipMemcpy(dT, host.data(), nTasks * sizeof(Task), hipMemcpyHostToDevice);
    Result a = run<32, 1, false>(dT, nTasks, data, lda, ticket, 3, K);
    Result w3 = runWide<32>(dT, nTasks, data, lda, 3, K);
    Result w2 = runWide<32>(dT, nTasks, data, lda, 2, K);
    Result w16 = runWide<16>(dT, nTasks, data, lda, 3, K);
    printf("%-10d%9.1f%9.1f%9.1f%9.1f   TF/s\n%-10s%9.1f%9.1f%9.1f%9.1f   us\n", nTasks, a.tf, w3.tf, w2.tf, w16.tf, "",
           a.ms * 1e3, w3.ms * 1e3, w2.ms * 1e3, w16.ms * 1e3);
    hipFree(dT);
  }
  if (argc > 1) return 0;   // (any argument: the wide-tile table only)
  const int sizes[] = {768, 1536, 1900, 2304, 3072, 6144};
  printf("%-10s", "tiles");
  const char* names[] = {"S1@3", "S1@4", "D32@2", "D16@3", "D16@4", "T16@3", "PS1@3", "PD32@2", "PD16@3", "PD16@4", "PT16@3"};
  for (auto nm : names) printf("%9s", nm);
  printf("\n");
  for (int nt : sizes) {
    auto host = makeTasks(nt);
    const int nTasks = (int)host.size();
    Task* dT; hipMalloc(&dT, nTasks * sizeof(Task));
    hipMemcpy(dT, host.data(), nTasks * sizeof(Task), hipMemcpyHostToDevice);
    Result r[11];
    r[0] = run<32, 1, false>(dT, nTasks, data, lda, ticket, 3, K);
    r[1] = run<32, 1, false>(dT, nTasks, data, lda, ticket, 4, K);
    r[2] = run<32, 2, false>(dT, nTasks, data, lda, ticket, 2, K);
    r[3] = run<16, 2, false>(dT, nTasks, data, lda, ticket, 3, K);
    r[4] = run<16, 2, false>(dT, nTasks, data, lda, ticket, 4, K);
    r[5] = run<16, 3, false>(dT, nTasks, data, lda, ticket, 3, K);
    r[6] = run<32, 1, true>(dT, nTasks, data, lda, ticket, 3, K);
    r[7] = run<32, 2, true>(dT, nTasks, data, lda, ticket, 2, K);
    r[8] = run<16, 2, true>(dT, nTasks, data, lda, ticket, 3, K);
    r[9] = run<16, 2, true>(dT, nTasks, data, lda, ticket, 4, K);
    r[10] = run<16, 3, true>(dT, nTasks, data, lda, ticket, 3, K);
    printf("%-10d", nTasks);
    for (int i = 0; i < 11; i++) printf("%9.1f", r[i].tf);
    printf("   TF/s\n%-10s", "");
    for (int i = 0; i < 11; i++) printf("%9.1f", r[i].ms * 1e3);
    printf("   us\n");
    hipFree(dT);
  }
  return 0;
}
