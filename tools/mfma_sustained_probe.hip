// What the fp64 matrix pipe sustains on this GPU, as a function of how long it is kept busy:
// every SIMD of the chip runs W waves of back-to-back v_mfma_f64_16x16x4 (4 independent
// accumulators), for launch lengths from ~50 us to ~50 ms; the shader clock is read from the
// ratio of clock64() (shader cycles) to wall_clock64() (100 MHz) inside the kernel.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_sustained_probe.hip -o tools/mfma_sustained_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

__global__ __launch_bounds__(256) void mfmaLoop(double* out, long long* clk, int iters) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-6;
  const long long s0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      c0 = MFMA(a, b, c0); c1 = MFMA(a, b, c1); c2 = MFMA(a, b, c2); c3 = MFMA(a, b, c3);
    }
  }
  const long long s1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = s1 - s0; clk[1] = w1 - w0; }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  double* out; long long* clk;
  hipMalloc(&out, 256 * 8 * 4096); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%8s %8s %10s %10s %12s %10s\n", "wg/CU", "iters", "ms", "TF/s", "shader MHz", "clk/mfma");
  for (int wg : {1, 2, 3, 4, 8}) {
    for (int iters : {100, 1000, 10000}) {
      float best = 1e30f; long long h[2] = {0, 0};
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        mfmaLoop<<<256 * wg, 256>>>(out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
      }
      const double flops = 256.0 * wg * 4 * iters * 16 * 2048;
      printf("%8d %8d %10.3f %10.1f %12.0f %10.1f\n", wg, iters, best, flops / (best * 1e-3) / 1e12,
             h[0] / (h[1] / 100.0), (double)h[0] / (iters * 16.0 * wg));
    }
  }
  return 0;
}
