// accuracy of v_rsq_f64 and of one / two Newton steps on it, against the host's long double 1/sqrt
// hipcc -O3 --offload-arch=gfx950 tools/rsq_accuracy_probe.hip -o tools/rsq_accuracy_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i], y = __builtin_amdgcn_rsq(d);
  y0[i] = y;
  double r = fma(-0.5 * d * y, y, 0.5);
  y = fma(y, r, y);
  y1[i] = y;
  r = fma(-0.5 * d * y, y, 0.5);
  y = fma(y, r, y);
  y2[i] = y;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> hx(n), h0(n), h1(n), h2(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (double)(s >> 11) / 9007199254740992.0;
    hx[i] = std::exp((u - 0.5) * 60.0);  // 1e-13 .. 1e13
  }
  double *x, *y0, *y1, *y2;
  hipMalloc(&x, n * 8); hipMalloc(&y0, n * 8); hipMalloc(&y1, n * 8); hipMalloc(&y2, n * 8);
  hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(x, y0, y1, y2, n);
  hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(h1.data(), y1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(h2.data(), y2, n * 8, hipMemcpyDeviceToHost);
  long double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) {
    long double t = 1.0L / sqrtl((long double)hx[i]);
    e0 = fmaxl(e0, fabsl(h0[i] - t) / t);
    e1 = fmaxl(e1, fabsl(h1[i] - t) / t);
    e2 = fmaxl(e2, fabsl(h2[i] - t) / t);
  }
  printf("max relative error: estimate %.3Le (2^%.1Lf)  one step %.3Le (%.2Lf ulp)  two steps %.3Le (%.2Lf ulp)\n", e0, log2l(e0), e1,
         e1 / 1.1102230246251565e-16L, e2, e2 / 1.1102230246251565e-16L);
  return 0;
}
