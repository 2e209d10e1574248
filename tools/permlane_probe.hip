// semantics of gfx950's v_permlane16_swap_b32 / v_permlane32_swap_b32 (row = 16 lanes), printed lane by lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + threadIdx.x] = q[0];
  out[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap16 r0 (from a=lane)", "swap16 r1 (from b=lane+100)", "swap32 r0", "swap32 r1"};
  for (int v = 0; v < 4; v++) {
    printf("%s:\n", names[v]);
    for (int row = 0; row < 4; row++) { printf("  row %d:", row); for (int i = 0; i < 16; i += 5) printf(" %3u", h[64 * v + 16 * row + i]); printf("\n"); }
  }
  return 0;
}
