// Inner loop of the rank-k tile update in isolation: per k-step 4 ds_read_b64 (2 A, 2 B operands)
// feed 4 v_mfma_f64_16x16x4_f64; operands already in LDS, no global traffic, no barriers.
// PIPE = 0: read, wait, multiply (the compiler's order for the rolled loop); PIPE = 1: the next
// step's operands are read before the current step's MFMAs are issued.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int PIPE, int UNROLL>
__global__ __launch_bounds__(256) void probe(double* out, int iters, int extraLds, int randomData) {
  constexpr int LD = 34;
  __shared__ double As[64 * LD];
  __shared__ double Bs[64 * LD];
  extern __shared__ double dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  for (int i = tid; i < 64 * LD; i += 256) {
    unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const double r = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    As[i] = randomData ? r : i * 1e-4;
    Bs[i] = randomData ? -r * 0.7 : i * 2e-4;
  }
  if (extraLds && tid == 0) dyn[0] = 0;
  __syncthreads();
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  d4 c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00;
  const double* pa0 = As + (wr + li) * LD + lk;
  const double* pa1 = As + (wr + 16 + li) * LD + lk;
  const double* pb0 = Bs + (wc + li) * LD + lk;
  const double* pb1 = Bs + (wc + 16 + li) * LD + lk;
  if (PIPE == 0) {
    for (int it = 0; it < iters; it++) {
#pragma unroll UNROLL
      for (int k0 = 0; k0 < 32; k0 += 4) {
        const double a0 = pa0[k0], a1 = pa1[k0], b0 = pb0[k0], b1 = pb1[k0];
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
      }
    }
  } else {
    double a0 = pa0[0], a1 = pa1[0], b0 = pb0[0], b1 = pb1[0];
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k0 = 0; k0 < 32; k0 += 4) {
        const int kn = (k0 + 4) & 31;
        const double na0 = pa0[kn], na1 = pa1[kn], nb0 = pb0[kn], nb1 = pb1[kn];
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        asm volatile("" ::: "memory");
      }
    }
  }
  out[blockIdx.x * 256 + tid] = c00[0] + c01[1] + c10[2] + c11[3];
}
template <int PIPE, int UNROLL>
void run(const char* name, int wgPerCu, double* out, int randomData = 0, int iters = 500) {
  // LDS per WG = 34.8 KB static; pad with dynamic LDS so that exactly wgPerCu fit (160 KB per CU)
  const int extra = wgPerCu == 4 ? 0 : wgPerCu == 3 ? 6144 : wgPerCu == 2 ? 30000 : 100000;
  hipFuncSetAttribute((const void*)probe<PIPE, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
  const int blocks = 256 * wgPerCu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<PIPE, UNROLL><<<blocks, 256, extra>>>(out, iters, extra, randomData);
  hipEventRecord(e0);
  probe<PIPE, UNROLL><<<blocks, 256, extra>>>(out, iters, extra, randomData);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(blocks) * 4 * iters * 32 * 2048.0;
  printf("%-28s WG/CU %d random %d : %6.1f TF/s (%.2f ms)\n", name, wgPerCu, randomData, flops / (ms * 1e-3) / 1e12, ms);
}
int main() {
  double* out; hipMalloc(&out, 256 * 4 * 256 * sizeof(double));
  for (int w : {1, 2, 3, 4}) {
    run<0, 1>("rolled read-wait-mfma", w, out);
    run<0, 8>("unrolled x8", w, out);
    run<1, 8>("pipelined operands", w, out);
  }
  for (int it : {500, 5000, 50000}) { run<0, 1>("rolled", 3, out, 0, it); run<0, 1>("rolled", 3, out, 1, it); }
  return 0;
}
