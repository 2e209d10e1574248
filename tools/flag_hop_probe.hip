// What does a dependency between two workgroups of ONE launch cost on gfx950, against a kernel
// boundary?  (Sizing of a level-walking kernel for small / batched fronts, DESIGN.md.)
// A chain of N workgroups: workgroup k waits for flag[k-1], reads the 64x64 fp64 tile that k-1 wrote,
// adds 1, writes its own tile, releases flag[k].  Consecutive workgroups sit on different XCDs
// (round-robin dispatch), so every hop crosses two non-coherent L2s: the release is an agent-scope
// write-back, the acquire an invalidate.
//   chain       the N hops inside one launch (in-order dispatch makes it deadlock-free)
//   launches    the same chain as N launches of one workgroup (what the level schedule pays today)
//   fan         levels of 1 -> F -> 1: a producer, F consumers that count in, a joiner (potrf -> trsm
//               tiles -> next stage), per-level latency
//   +load       the same beside a streaming kernel that keeps every L2 full of dirty lines
// build: hipcc -O3 --offload-arch=gfx950 tools/flag_hop_probe.hip -o tools/flag_hop_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

constexpr int kTileElems = 64 * 64;
constexpr unsigned kSpinMax = 1u << 22;  // bounded: a protocol bug ends as a wrong answer, not a hang

// POLL_ACQUIRE: every poll is an acquire load (an L2 invalidate per poll: with a thousand waiting
// workgroups nothing stays cached); otherwise relaxed polls and ONE acquire fence after the last
template <bool POLL_ACQUIRE>
__device__ __forceinline__ bool waitGE(const unsigned* f, unsigned want) {
  unsigned n = 0;
  if (POLL_ACQUIRE) {
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++n > kSpinMax) return false;
    }
  } else {
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(8);
      if (++n > kSpinMax) return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  return true;
}

__device__ __forceinline__ void tileStep(const double* in, double* out) {
#pragma unroll
  for (int e = 0; e < 16; e++) out[threadIdx.x + 256 * e] = in[threadIdx.x + 256 * e] + 1.0;
}

template <bool PA>
__global__ __launch_bounds__(256) void chain(double* tiles, unsigned* flags, unsigned epoch, unsigned* err) {
  const int k = blockIdx.x;
  if (k > 0) {
    if (threadIdx.x == 0 && !waitGE<PA>(flags + k - 1, epoch)) atomicAdd(err, 1u);
    __syncthreads();
  }
  tileStep(tiles + (size_t)k * kTileElems, tiles + (size_t)(k + 1) * kTileElems);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + k, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void oneStep(double* tiles, int k) {
  tileStep(tiles + (size_t)k * kTileElems, tiles + (size_t)(k + 1) * kTileElems);
}

// level l: workgroup 0 = producer (waits for join[l-1] == F, writes tile P_l, sets prod[l]); workgroups
// 1..F = consumers (wait for prod[l], read P_l, write their own tile C_{l,f}, count into join[l]).
// The producer of level l + 1 reads C_{l,0}.  blockIdx.x = l * (F + 1) + role.
template <bool PA>
__global__ __launch_bounds__(256) void fan(double* tiles, const double* seed, unsigned* prod,
                                           unsigned* join, int F, unsigned epoch, unsigned* err) {
  const int l = blockIdx.x / (F + 1), role = blockIdx.x % (F + 1);
  double* P = tiles + (size_t)l * (F + 1) * kTileElems;
  if (role == 0) {
    const double* src = seed;  // level 0 reads the seed tile (zeros)
    if (l > 0) {
      if (threadIdx.x == 0 && !waitGE<PA>(join + l - 1, epoch * F)) atomicAdd(err, 1u);
      __syncthreads();
      src = tiles + ((size_t)(l - 1) * (F + 1) + 1) * kTileElems;
    }
    tileStep(src, P);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(prod + l, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (threadIdx.x == 0 && !waitGE<PA>(prod + l, epoch)) atomicAdd(err, 1u);
    __syncthreads();
    tileStep(P, P + (size_t)role * kTileElems);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(join + l, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(256) void fanProd(double* tiles, const double* seed, int l, int F) {
  const double* src = l > 0 ? tiles + ((size_t)(l - 1) * (F + 1) + 1) * kTileElems : seed;
  tileStep(src, tiles + (size_t)l * (F + 1) * kTileElems);
}
__global__ __launch_bounds__(256) void fanCons(double* tiles, int l, int F) {
  double* P = tiles + (size_t)l * (F + 1) * kTileElems;
  tileStep(P, P + (size_t)(blockIdx.x + 1) * kTileElems);
}

// background: read-modify-write stream over a buffer far larger than the L2s
__global__ __launch_bounds__(256) void dirty(double* buf, size_t n, int reps) {
  for (int r = 0; r < reps; r++) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] += 1.0;
  }
}

int main() {
  const int N = 64, F = 32, L = 32, reps = 20;
  double* tiles;
  unsigned *flags, *err;
  const size_t nTiles = (size_t)L * (F + 1) + N + 2;
  CK(hipMalloc(&tiles, nTiles * kTileElems * sizeof(double)));
  CK(hipMalloc(&flags, 4096 * sizeof(unsigned)));
  CK(hipMalloc(&err, sizeof(unsigned)));
  CK(hipMemset(tiles, 0, nTiles * kTileElems * sizeof(double)));
  CK(hipMemset(flags, 0, 4096 * sizeof(unsigned)));
  CK(hipMemset(err, 0, sizeof(unsigned)));
  double* big;
  const size_t nBig = (size_t)1 << 27;  // 1 GB
  CK(hipMalloc(&big, nBig * sizeof(double)));
  CK(hipMemset(big, 0, nBig * sizeof(double)));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned epoch = 0;
  auto check = [&](size_t tileIdx, double want, const char* what) {
    double v = 0;
    hipMemcpy(&v, tiles + tileIdx * kTileElems + 77, sizeof(double), hipMemcpyDeviceToHost);
    unsigned e = 0;
    hipMemcpy(&e, err, sizeof(unsigned), hipMemcpyDeviceToHost);
    if (v != want || e) printf("  !! %s: value %.1f (want %.1f), spin time-outs %u\n", what, v, want, e);
  };
  for (int load = 0; load < 2; load++) {
    printf("%s\n", load ? "beside a streaming read-modify-write kernel (every L2 full of dirty lines):" : "idle GPU:");
    float ms;
    for (int pa = 0; pa < 2; pa++) {
    // ---- chain, one launch
    for (int it = 0; it < 3; it++) {
      if (load) hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, s1, big, nBig, 4);
      CK(hipMemsetAsync(tiles, 0, kTileElems * sizeof(double), s0));
      CK(hipEventRecord(e0, s0));
      for (int r = 0; r < reps; r++) {
        epoch++;
        if (pa) hipLaunchKernelGGL(chain<true>, dim3(N), dim3(256), 0, s0, tiles, flags, epoch, err);
        else hipLaunchKernelGGL(chain<false>, dim3(N), dim3(256), 0, s0, tiles, flags, epoch, err);
      }
      CK(hipEventRecord(e1, s0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it == 2) printf("  chain of %d hops in one launch (%s): %7.2f us per launch, %5.2f us per hop\n", N, pa ? "acquire polls" : "relaxed polls", 1e3 * ms / reps, 1e3 * ms / reps / N);
    }
    check(N, (double)N, "chain");
    }
    // ---- chain, N launches
    for (int it = 0; it < 3; it++) {
      if (load) hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, s1, big, nBig, 4);
      CK(hipEventRecord(e0, s0));
      for (int r = 0; r < reps; r++) {
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(oneStep, dim3(1), dim3(256), 0, s0, tiles, k);
      }
      CK(hipEventRecord(e1, s0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it == 2) printf("  the same as %d launches:             %7.2f us,            %5.2f us per hop\n", N, 1e3 * ms / reps, 1e3 * ms / reps / N);
    }
    // ---- fan, one launch
    for (int pa = 0; pa < 2; pa++) {
    CK(hipMemset(flags, 0, 4096 * sizeof(unsigned)));
    epoch = 0;
    for (int it = 0; it < 3; it++) {
      if (load) hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, s1, big, nBig, 4);
      CK(hipEventRecord(e0, s0));
      for (int r = 0; r < reps; r++) {
        epoch++;
        if (pa) hipLaunchKernelGGL(fan<true>, dim3(L * (F + 1)), dim3(256), 0, s0, tiles, tiles + (nTiles - 1) * kTileElems, flags, flags + 2048, F, epoch, err);
        else hipLaunchKernelGGL(fan<false>, dim3(L * (F + 1)), dim3(256), 0, s0, tiles, tiles + (nTiles - 1) * kTileElems, flags, flags + 2048, F, epoch, err);
      }
      CK(hipEventRecord(e1, s0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it == 2) printf("  %d levels of 1 -> %d -> 1, one launch (%s): %7.2f us per launch, %5.2f us per level\n", L, F, pa ? "acquire polls" : "relaxed polls", 1e3 * ms / reps, 1e3 * ms / reps / L);
    }
    check((size_t)(L - 1) * (F + 1) + 1, 2.0 * L, "fan");
    }
    // ---- fan, 2 L launches
    for (int it = 0; it < 3; it++) {
      if (load) hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, s1, big, nBig, 4);
      CK(hipEventRecord(e0, s0));
      for (int r = 0; r < reps; r++) {
        for (int l = 0; l < L; l++) {
          hipLaunchKernelGGL(fanProd, dim3(1), dim3(256), 0, s0, tiles, tiles + (nTiles - 1) * kTileElems, l, F);
          hipLaunchKernelGGL(fanCons, dim3(F), dim3(256), 0, s0, tiles, l, F);
        }
      }
      CK(hipEventRecord(e1, s0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it == 2) printf("  the same as %d launches:              %7.2f us,            %5.2f us per level\n", 2 * L, 1e3 * ms / reps, 1e3 * ms / reps / L);
    }
  }
  return 0;
}
