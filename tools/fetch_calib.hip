// Calibration of rocprofv3's FETCH_SIZE (and the TCC_EA0_RDREQ request counters behind it) on gfx950
// for the access pattern of the sparse-elimination update (hip_kernels.h, elimGatherMfma): 216-byte
// source blocks (9 x 3 doubles) at 72-byte granularity, read 8 bytes per lane, two blocks per wave
// load (lanes 0-26 one block, lanes 32-58 another).  MI355X_MICROARCH.md calibrates the x2 correction
// for 16-byte-per-lane coalesced streams only.  Every kernel reads a KNOWN number of distinct bytes,
// none twice, from a buffer several times the 256 MB Infinity Cache:
//   stream16   coalesced, 16 B / lane      (the guide's reference: FETCH_SIZE reads half the bytes)
//   stream8    coalesced, 8 B / lane
//   blocks216  random 216-byte blocks at 72-byte granularity (2.69 cache lines each on average)
//   blocks256  random 256-byte-aligned 216-byte blocks (exactly 2 lines each)
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/fetch_calib   (see profiles/r4_fetch_calib.sh)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void stream16(const d2* src, size_t n2, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const d2 v = __builtin_nontemporal_load(src + i);
    s += v.x + v.y;
  }
  if (s == 1.2345) out[0] = s;
}
__global__ __launch_bounds__(256) void stream8(const double* src, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += src[i];
  if (s == 1.2345) out[0] = s;
}
// one wave reads two blocks per load instruction, 4 loads in flight, as elimGatherMfma does
__global__ __launch_bounds__(256) void blocks(const double* src, const uint32_t* offs, int nPairs, double* out) {
  const int lane = threadIdx.x & 63, half = lane >> 5, e = lane & 31;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nWaves = gridDim.x * 4;
  const uint32_t eOff = e < 27 ? e : 0;
  double s = 0;
  for (int p = wave * 4; p < nPairs; p += nWaves * 4) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = min(p + u, nPairs - 1);
      v[u] = src[(size_t)offs[2 * q + half] + eOff];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) s += v[u];
  }
  if (s == 1.2345) out[0] = s;
}

int main() {
  const size_t bufBytes = (size_t)5 << 30;  // 5 GB: four disjoint regions (stream16, stream8, blocks216, blocks256)
  const size_t nD = bufBytes / 8;
  double *buf, *out;
  CK(hipMalloc(&buf, bufBytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, bufBytes));
  // stream kernels: the first 1 GB
  const size_t streamBytes = (size_t)1 << 30;
  hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const d2*)buf, streamBytes / 16, out);
  hipLaunchKernelGGL(stream8, dim3(4096), dim3(256), 0, 0, (const double*)(buf + ((size_t)1 << 27)), streamBytes / 8, out);
  CK(hipDeviceSynchronize());
  // block kernels: 4 M distinct blocks each (864 MB of useful bytes) from the rest of the buffer
  const int nBlocks = 4 << 20;
  std::mt19937_64 rng(12345);
  for (int variant = 0; variant < 2; variant++) {
    // slots of 9 doubles (72 bytes); a block = 3 consecutive slots.  variant 0: block b at slot 3 * perm(b)
    // + (random 0..2 shift kept inside its own 4-slot cell), i.e. arbitrary 72-byte alignment, distinct
    // bytes; variant 1: 256-byte aligned
    std::vector<uint32_t> cell(nBlocks);
    std::iota(cell.begin(), cell.end(), 0u);
    std::shuffle(cell.begin(), cell.end(), rng);
    std::vector<uint32_t> offs(nBlocks);
    const size_t base = variant == 0 ? ((size_t)2 << 27) : ((size_t)2 << 27) + (size_t)nBlocks * 36 + 1024;  // (disjoint regions: nothing cached)
    double lines = 0;
    for (int b = 0; b < nBlocks; b++) {
      size_t off;  // in doubles
      if (variant == 0) {
        off = (size_t)cell[b] * 36 + (rng() % 2) * 9;  // 36 doubles = 288-byte cell, block of 27 at +0 or +9
      } else {
        off = (size_t)cell[b] * 32;                    // 256-byte aligned
      }
      const size_t byte0 = off * 8, byte1 = byte0 + 216 - 1;
      lines += double(byte1 / 128 - byte0 / 128 + 1);
      offs[b] = (uint32_t)(base + off);
    }
    if (base + (size_t)nBlocks * 36 >= ((size_t)1 << 32) || base + (size_t)nBlocks * 36 > nD) {
      fprintf(stderr, "offsets do not fit 32 bits\n");
      return 1;
    }
    uint32_t* dOffs;
    CK(hipMalloc(&dOffs, (size_t)nBlocks * 4));
    CK(hipMemcpy(dOffs, offs.data(), (size_t)nBlocks * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(blocks, dim3(8192), dim3(256), 0, 0, (const double*)buf, dOffs, nBlocks / 2, out);
    CK(hipDeviceSynchronize());
    printf("variant %d (%s): %d blocks, useful bytes %.1f MB, cache lines touched %.0f = %.1f MB, offsets %.1f MB\n",
           variant, variant == 0 ? "72-byte granularity" : "256-byte aligned", nBlocks, nBlocks * 216.0 / 1e6, lines,
           lines * 128 / 1e6, nBlocks * 4.0 / 1e6);
    CK(hipFree(dOffs));
  }
  printf("stream16 / stream8: %.1f MB each\n", streamBytes / 1e6);
  return 0;
}
