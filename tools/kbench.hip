// Developer micro-benchmark of the panel kernels on a dense n x n lump (not part of the product).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Ibaspacho_amd/csrc tools/kbench.hip -o /tmp/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define BSP_KDEBUG 1
#include "hip_kernels.h"
__global__ void emptyKernel() {}
using namespace BaSpaCho;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void clockProbe(long long* out, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  double x = threadIdx.x;
  for (int i = 0; i < iters; i++) x = x * 1.0000001 + 0.5;
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}

__global__ __launch_bounds__(256) void mfmaPeak(double* out, int iters) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void fmaPeak(double* out, int iters) {
  double c[8]; for (int j = 0; j < 8; j++) c[j] = j;
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) c[j] = __builtin_fma(c[j], b, a);
  }
  double s = 0; for (int j = 0; j < 8; j++) s += c[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float timeIt(F&& f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1000.f;  // us
}

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 7839;
  int c0 = argc > 2 ? atoi(argv[2]) : 0;  // panel start column
  size_t elems = (size_t)n * n;
  std::vector<double> h(elems);
  for (size_t i = 0; i < elems; i++) h[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  for (int i = 0; i < n; i++) h[(size_t)i * n + i] = n * 1.2;
  double* d; CK(hipMalloc(&d, elems * 8)); CK(hipMemcpy(d, h.data(), elems * 8, hipMemcpyHostToDevice));

  long long* dc; CK(hipMalloc(&dc, 64));
  for (int blocks : {1, 256}) {
    clockProbe<<<blocks, 64>>>(dc, 100000); long long hc[3]; CK(hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost));
    printf("clockProbe blocks=%d: shader cycles %lld, wall ticks(100MHz) %lld -> %.2f GHz\n", blocks, hc[0], hc[1], hc[0] / (hc[1] * 10.0));
  }

  {
    double* dout; CK(hipMalloc(&dout, 8 * 256 * 2048));
    for (int blocks : {256, 512, 1024}) {
      int iters = 20000;
      float us2 = timeIt([&] { mfmaPeak<<<blocks, 256>>>(dout, iters); }, 3);
      double fl = (double)blocks * 4 * iters * 4 * 2048.0;
      printf("mfma f64 16x16x4 peak probe, %d blocks x 4 waves: %.1f TF/s\n", blocks, fl / us2 / 1e6);
      us2 = timeIt([&] { fmaPeak<<<blocks, 256>>>(dout, iters); }, 3);
      fl = (double)blocks * 256 * iters * 8 * 2.0;
      printf("v_fma_f64 peak probe, %d blocks: %.1f TF/s\n", blocks, fl / us2 / 1e6);
    }
  }
  const int nb = 64;
  PanelDesc pd{}; pd.diagOff = (int64_t)c0 * n + c0; pd.lda = n; pd.nb = nb; pd.nRest = n - c0 - nb; pd.rowsBelow = pd.nRest; pd.lumpRowBase = 0; pd.lump = 0;
  int K = argc > 3 ? atoi(argv[3]) : 64;  // source width of the update
  SrcDesc sr{}; sr.off = pd.diagOff + (int64_t)nb * n - (K - nb); sr.lda = n; sr.K = K; sr.rowsBelow = pd.rowsBelow; sr.nRest = pd.nRest; sr.lumpRowBase = 0;
  SegDesc sd{}; sd.src = 0; sd.kind = kSegIntra; sd.q0 = 0; sd.m = pd.nRest; sd.tgtBase = (int64_t)(c0 + nb) * n + (c0 + nb); sd.tgtStride = n;
  // self-contained records, as DevPlan::upload builds them (hip_plan.h: TrsmTaskFat, UpdTaskWide)
  std::vector<TrsmTaskFat> tt;
  for (int r = 0; r < pd.rowsBelow; r += kTile) tt.push_back(TrsmTaskFat{pd.diagOff, pd.lda, pd.nb, pd.rowsBelow, r, 0, 0});
  std::vector<UpdTaskWide> ut;
  for (int cT = 0; cT < sd.m; cT += kTile) {
    for (int rT = cT; rT < pd.rowsBelow; rT += kTile) {
      UpdTaskWide w{};
      w.srcOff = sr.off; w.tgtBase = sd.tgtBase; w.chainTabPtr = 0;
      w.lda = sr.lda; w.K = sr.K; w.rowsBelow = sr.rowsBelow; w.nRest = sr.nRest;
      w.lumpRowBase = 0; w.kind = kSegIntra; w.segEnd = sd.q0 + sd.m; w.tgtStride = sd.tgtStride;
      w.firstChainOrd = 0; w.rowMin = 0; w.rowTile = rT; w.colTile = cT; w.atomic = 0;
      ut.push_back(w);
    }
  }
  std::vector<UpdTaskWide> utA = ut; for (auto& t : utA) t.atomic = 1;
  PanelDesc* dpd; TrsmTaskFat* dtt; UpdTaskWide* dut; UpdTaskWide* dutA;
  CK(hipMalloc(&dpd, sizeof pd)); CK(hipMemcpy(dpd, &pd, sizeof pd, hipMemcpyHostToDevice));
  CK(hipMalloc(&dtt, tt.size() * sizeof(TrsmTaskFat))); CK(hipMemcpy(dtt, tt.data(), tt.size() * sizeof(TrsmTaskFat), hipMemcpyHostToDevice));
  CK(hipMalloc(&dut, ut.size() * sizeof(UpdTaskWide))); CK(hipMemcpy(dut, ut.data(), ut.size() * sizeof(UpdTaskWide), hipMemcpyHostToDevice));
  CK(hipMalloc(&dutA, ut.size() * sizeof(UpdTaskWide))); CK(hipMemcpy(dutA, utA.data(), ut.size() * sizeof(UpdTaskWide), hipMemcpyHostToDevice));
  hipk::DataRef<double> ref{d, nullptr};

  printf("n=%d panel at %d: rowsBelow=%d trsmTasks=%zu updTasks=%zu\n", n, c0, pd.rowsBelow, tt.size(), ut.size());
  float us;
  us = timeIt([&] { hipk::potrfPanel<double><<<1, 256>>>(dpd, ref); }, 50);
  printf("potrfPanel        : %8.1f us\n", us);
  { long long st[16]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(hipk::bspDebugStamps), sizeof st));
    printf("   potrf cycles: load %lld, loop %lld, store %lld (total %lld = %.1f us @2.4GHz)\n", st[1]-st[0], st[2]-st[1], st[3]-st[2], st[3]-st[0], (st[3]-st[0])/2400.0); }
  us = timeIt([&] { emptyKernel<<<1, 256>>>(); }, 200);
  printf("empty kernel      : %8.1f us\n", us);
  us = timeIt([&] { hipk::trsmPanel<double><<<(unsigned)tt.size(), 256>>>(dtt, ref); }, 50);
  printf("trsmPanel         : %8.1f us  (%zu tasks)\n", us, tt.size());
  { long long st[16]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(hipk::bspDebugStamps), sizeof st));
    printf("   trsm cycles (block 0): stage L %lld, rows %lld (total %.1f us)\n", st[5]-st[4], st[6]-st[5], (st[6]-st[4])/2400.0); }
  double updFlops = 0; { double R = pd.rowsBelow, m = sd.m; updFlops = 2.0 * K * (m * R - m * (m - 1) / 2); }
  us = timeIt([&] { hipk::updateTile<double, false><<<(unsigned)ut.size(), 256>>>(dut, nullptr, nullptr, nullptr, nullptr, ref); }, 20);
  printf("updateTile plain  : %8.1f us  -> %.2f TF/s\n", us, updFlops / us / 1e6);
  us = timeIt([&] { hipk::updateTile<double, false><<<(unsigned)ut.size(), 256>>>(dutA, nullptr, nullptr, nullptr, nullptr, ref); }, 20);
  printf("updateTile atomic : %8.1f us  -> %.2f TF/s\n", us, updFlops / us / 1e6);
  CK(hipDeviceSynchronize());
  return 0;
}
