// Does a saturating fp64-MFMA kernel slow an unrelated latency-bound workgroup chip-wide
// (power management) or only when both share a CU?  A one-workgroup latency loop is timed with
// the shader clock (clock64) and the constant 100 MHz clock (wall_clock64) alone, beside a hog
// on every CU and beside a hog on a fraction of the CUs; both kernels record where they ran.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned whereAmI() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // cu_id[11:8] sh_id[12] se_id[15:13]
  return ((xcc & 0xf) << 8) | (((hw >> 13) & 0x7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
}

__global__ __launch_bounds__(256) void hog(double* out, int iters, unsigned* where) {
  if (threadIdx.x == 0) where[blockIdx.x] = whereAmI();
  double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double x = threadIdx.x * 1e-3, y = 1.0 + blockIdx.x * 1e-6;
  for (int i = 0; i < iters; i++) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

__global__ __launch_bounds__(256) void latencyLoop(double* out, long long* stamps, int steps, int prio = 0) {
  __shared__ double sh[256];
  if (prio) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  double v = 1.0 + tid * 1e-3;
  sh[tid] = v;
  __syncthreads();
  long long c0 = clock64(), w0 = wall_clock64();
  for (int s = 0; s < steps; s++) {
    double u = sh[(tid + 17) & 255];
    v = v * 0.999 + u * 1e-3;        // dependent fp64 chain
    v = v * 0.999 + 1e-4;
    v = v * 0.999 + 1e-4;
    v = v * 0.999 + 1e-4;
    __syncthreads();
    sh[tid] = v;
    __syncthreads();
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (tid == 0) {
    stamps[0] = c1 - c0;
    stamps[1] = w1 - w0;
    stamps[2] = whereAmI();
  }
  out[tid] = v;
}

int main() {
  double* out; unsigned* where; long long* stamps; double* lout;
  CK(hipMalloc(&out, sizeof(double) * 256 * 4096));
  CK(hipMalloc(&where, sizeof(unsigned) * 4096));
  CK(hipMalloc(&stamps, sizeof(long long) * 4));
  CK(hipMalloc(&lout, sizeof(double) * 256));
  hipStream_t sA, sB;
  CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
  auto runLatency = [&](const char* tag, std::set<unsigned>* hogCus) -> int {
    latencyLoop<<<1, 256, 0, sA>>>(lout, stamps, 2000);
    CK(hipStreamSynchronize(sA));
    long long h[3];
    CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
    const double wallUs = h[1] / 100.0;
    printf("%-28s shader clocks %8lld  wall %8.1f us  => %.0f MHz, %.1f clk/step, on xcc%llu se%llu sh%llu cu%llu%s\n", tag,
           h[0], wallUs, h[0] / wallUs, h[0] / 2000.0, (unsigned long long)(h[2] >> 8), (unsigned long long)((h[2] >> 5) & 7),
           (unsigned long long)((h[2] >> 4) & 1), (unsigned long long)(h[2] & 15),
           hogCus ? (hogCus->count((unsigned)h[2]) ? "  [CU shared with hog]" : "  [CU free of hog]") : "");
    return 0;
  };
  if (runLatency("alone (cold)", nullptr)) return 1;
  if (runLatency("alone", nullptr)) return 1;
  const int grids[] = {1024, 256, 128, 64, 16};
  for (int g : grids) {
    const int iters = 400000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sB));
    hog<<<g, 256, 0, sB>>>(out, iters, where);
    CK(hipEventRecord(e1, sB));
    // wait until the hog is surely running, then time the latency loop a few times
    for (volatile int spin = 0; spin < 20000000; spin++) {}
    std::vector<unsigned> hw(g);
    char tag[64];
    // `where` is written at kernel start; read it after the hog ends, so classify afterwards
    std::vector<long long> rec;
    for (int rep = 0; rep < 3; rep++) {
      latencyLoop<<<1, 256, 0, sA>>>(lout, stamps, 2000);
      CK(hipStreamSynchronize(sA));
      long long h[3];
      CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
      rec.insert(rec.end(), h, h + 3);
    }
    const bool stillRunning = hipEventQuery(e1) == hipErrorNotReady;
    CK(hipStreamSynchronize(sB));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hw.data(), where, sizeof(unsigned) * g, hipMemcpyDeviceToHost));
    std::set<unsigned> cus(hw.begin(), hw.end());
    const double tf = 4.0 * 2 * 16 * 16 * 4 * 4 * (double)iters * g / (ms * 1e-3) / 1e12;
    printf("hog grid %4d: %.1f ms, %.2f TF/s on %zu distinct CUs, overlap %s\n", g, ms, tf, cus.size(),
           stillRunning ? "yes" : "NO (hog ended early)");
    for (int rep = 0; rep < 3; rep++) {
      const long long* h = &rec[rep * 3];
      const double wallUs = h[1] / 100.0;
      snprintf(tag, sizeof tag, "  beside hog grid %d", g);
      printf("%-28s shader clocks %8lld  wall %8.1f us  => %.0f MHz, %.1f clk/step %s\n", tag, h[0], wallUs,
             h[0] / wallUs, h[0] / 2000.0, cus.count((unsigned)h[2]) ? "[CU shared with hog]" : "[CU free of hog]");
    }
  }
  // ---- does s_setprio(3) protect the latency loop?
  for (int g : {1024, 768, 512}) {
    const int iters = 100000;
    hipEvent_t e1;
    CK(hipEventCreate(&e1));
    hog<<<g, 256, 0, sB>>>(out, iters, where);
    CK(hipEventRecord(e1, sB));
    for (volatile int spin = 0; spin < 20000000; spin++) {}
    for (int prio = 0; prio < 2; prio++) {
      for (int rep = 0; rep < 2; rep++) {
        latencyLoop<<<1, 256, 0, sA>>>(lout, stamps, 200, prio);
        CK(hipStreamSynchronize(sA));
        long long h[3];
        CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
        printf("hog grid %d, latency loop prio %d: %.1f clk/step (overlap %s)\n", g, prio * 3, h[0] / 200.0,
               hipEventQuery(e1) == hipErrorNotReady ? "yes" : "NO");
      }
    }
    CK(hipStreamSynchronize(sB));
  }
  // ---- CU masks: does hipExtStreamCreateWithCUMask confine the hog, and to which CUs?
  for (int k : {2, 4, 8}) {
    std::vector<uint32_t> mHog(8, 0u), mLat(8, 0u);
    for (int cu = 0; cu < 256; cu++) {
      if (cu % k == k - 1) mLat[cu / 32] |= 1u << (cu % 32); else mHog[cu / 32] |= 1u << (cu % 32);
    }
    hipStream_t sH, sL;
    CK(hipExtStreamCreateWithCUMask(&sH, 8, mHog.data()));
    CK(hipExtStreamCreateWithCUMask(&sL, 8, mLat.data()));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int g = 1024, iters = 400000;
    CK(hipEventRecord(e0, sH));
    hog<<<g, 256, 0, sH>>>(out, iters, where);
    CK(hipEventRecord(e1, sH));
    for (volatile int spin = 0; spin < 20000000; spin++) {}
    std::vector<long long> rec;
    for (int rep = 0; rep < 3; rep++) {
      latencyLoop<<<1, 256, 0, rep == 2 ? sA : sL>>>(lout, stamps, 2000);
      CK(hipStreamSynchronize(rep == 2 ? sA : sL));
      long long h[3];
      CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
      rec.insert(rec.end(), h, h + 3);
    }
    const bool stillRunning = hipEventQuery(e1) == hipErrorNotReady;
    CK(hipStreamSynchronize(sH));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> hw(g);
    CK(hipMemcpy(hw.data(), where, sizeof(unsigned) * g, hipMemcpyDeviceToHost));
    std::set<unsigned> cus(hw.begin(), hw.end());
    std::set<unsigned> xccs;
    for (unsigned c : cus) xccs.insert(c >> 8);
    const double tf = 4.0 * 2 * 16 * 16 * 4 * 4 * (double)iters * g / (ms * 1e-3) / 1e12;
    printf("masked hog (reserve every %d-th CU): %.1f ms, %.2f TF/s on %zu distinct CUs in %zu XCCs, overlap %s\n", k, ms,
           tf, cus.size(), xccs.size(), stillRunning ? "yes" : "NO");
    for (int rep = 0; rep < 3; rep++) {
      const long long* h = &rec[rep * 3];
      printf("   latency loop on %s stream: %.1f us, %.1f clk/step, xcc%lld se%lld sh%lld cu%lld %s\n",
             rep == 2 ? "UNMASKED" : "complement-masked", h[1] / 100.0, h[0] / 2000.0, h[2] >> 8, (h[2] >> 5) & 7,
             (h[2] >> 4) & 1, h[2] & 15, cus.count((unsigned)h[2]) ? "[CU shared with hog]" : "[CU free of hog]");
    }
  }
  return 0;
}
