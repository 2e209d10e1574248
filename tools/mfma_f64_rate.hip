// What limits back-to-back v_mfma_f64_16x16x4_f64?  Sweeps independent accumulators per wave and
// waves per SIMD.  Build + run:  hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate.hip -o /tmp/mr && /tmp/mr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void probe(double* out, int iters) {
  d4 c[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) c[k] = d4{0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < NACC; k++) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < NACC; k++) s += c[k][k & 3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wavesPerSimd, double* out, int mfmasPerWave = 8000) {
  const int threads = 256;                       // 4 waves = 1 per SIMD
  const int blocks = 256 * wavesPerSimd;          // 256 CUs
  const int iters = mfmasPerWave / NACC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC><<<blocks, threads>>>(out, iters);
  hipEventRecord(e0);
  probe<NACC><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(blocks) * 4 * iters * NACC * 2048.0;
  const double clkPerMfma = ms * 1e-3 * 2.4e9 / (double(iters) * NACC * wavesPerSimd);
  printf("%7.2f ms  acc %d  waves/SIMD %d : %6.1f TF/s   %5.1f clk per MFMA per SIMD (at 2.4 GHz)\n", ms, NACC,
         wavesPerSimd, flops / (ms * 1e-3) / 1e12, clkPerMfma);
}
int main() {
  double* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  for (int w : {1, 2, 4, 8}) { run<1>(w, out); run<2>(w, out); run<4>(w, out); run<8>(w, out); }
  // sustained: the same stream of MFMAs for longer and longer (power management)
  for (int n : {8000, 32000, 128000, 512000, 2048000}) run<4>(4, out, n);
  for (int rep = 0; rep < 5; rep++) run<4>(4, out, 32000);
  return 0;
}
