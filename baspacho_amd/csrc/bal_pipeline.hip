// Caller pipeline of the bundle-adjustment-at-large benchmark on the device: linearisation of the
// reprojection residuals and assembly of the Gauss-Newton Hessian / gradient straight into the
// solver's numeric buffer through Solver::deviceAccessor() -- what the reference's optimizer does on
// the host in its computeStep (benchmarking/BaAtLargeOptimizer.cpp:100-131: accessor.diagBlock /
// accessor.block per observation, then the Levenberg-Marquardt damping of every diagonal) and what
// its CUDA backend offers the accessor for (Accessor.h:110-200 is host/device, MatOpsCuda.cu:85-92
// hands out the device arrays).  Cameras carry the 9 parameters of a BAL file (Rodrigues rotation,
// translation, f, k1, k2 -- the block size BAL_bench uses, BaAtLargeBench.cpp:50-65); the
// derivatives come from forward-mode dual numbers, so there is no hand-derived Jacobian to get wrong.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>

#include "accessor.h"
#include "bal_pipeline.h"

namespace BaSpaCho {

namespace {

#define balCHECK(expr)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr); \
    }                                                                                       \
  } while (0)

constexpr int kND = 12;  // 9 camera + 3 point partials

struct Dual {
  double v;
  double d[kND];
};
__device__ __forceinline__ Dual constant(double v) {
  Dual r;
  r.v = v;
#pragma unroll
  for (int i = 0; i < kND; i++) r.d[i] = 0.0;
  return r;
}
__device__ __forceinline__ Dual variable(double v, int idx) {
  Dual r = constant(v);
  r.d[idx] = 1.0;
  return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < kND; i++) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < kND; i++) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < kND; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
  Dual r;
  const double inv = 1.0 / b.v;
  r.v = a.v * inv;
#pragma unroll
  for (int i = 0; i < kND; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ Dual scale(const Dual& a, double s, double ds_dv) {
  // f(a) with f(a.v) = s and f'(a.v) = ds_dv
  Dual r;
  r.v = s;
#pragma unroll
  for (int i = 0; i < kND; i++) r.d[i] = ds_dv * a.d[i];
  return r;
}

// residual (2) and its derivatives for one observation; camera = [r(3), t(3), f, k1, k2]
__device__ __forceinline__ void reproject(const double* cam, const double* pt, const double* xy,
                                          Dual (&res)[2]) {
  Dual w[3], t[3], X[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    w[i] = variable(cam[i], i);
    t[i] = variable(cam[3 + i], 3 + i);
    X[i] = variable(pt[i], 9 + i);
  }
  const Dual f = variable(cam[6], 6), k1 = variable(cam[7], 7), k2 = variable(cam[8], 8);
  // Rodrigues rotation of X by the axis-angle vector w
  const Dual th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  Dual P[3];
  if (th2.v > 1e-20) {
    const double th = sqrt(th2.v);
    const Dual theta = scale(th2, th, 0.5 / th);
    const Dual c = scale(theta, cos(th), -sin(th)), s = scale(theta, sin(th), cos(th));
    Dual k[3];
#pragma unroll
    for (int i = 0; i < 3; i++) k[i] = w[i] / theta;
    const Dual kx[3] = {k[1] * X[2] - k[2] * X[1], k[2] * X[0] - k[0] * X[2], k[0] * X[1] - k[1] * X[0]};
    const Dual kdot = (k[0] * X[0] + k[1] * X[1] + k[2] * X[2]) * (constant(1.0) - c);
#pragma unroll
    for (int i = 0; i < 3; i++) P[i] = X[i] * c + kx[i] * s + k[i] * kdot;
  } else {  // first-order: R X = X + w x X
    P[0] = X[0] + (w[1] * X[2] - w[2] * X[1]);
    P[1] = X[1] + (w[2] * X[0] - w[0] * X[2]);
    P[2] = X[2] + (w[0] * X[1] - w[1] * X[0]);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) P[i] = P[i] + t[i];
  const Dual zero = constant(0.0);
  const Dual px = (zero - P[0]) / P[2], py = (zero - P[1]) / P[2];
  const Dual r2 = px * px + py * py;
  const Dual dist = constant(1.0) + r2 * (k1 + k2 * r2);
  res[0] = f * dist * px - constant(xy[0]);
  res[1] = f * dist * py - constant(xy[1]);
}

__global__ __launch_bounds__(64) void balLinearizeKernel(int64_t numObs, const int64_t* obsCam,
                                                         const int64_t* obsPt, const double* obsXy,
                                                         const double* cams, const double* pts,
                                                         double* res, double* Jc, double* Jp) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= numObs) return;
  Dual r[2];
  reproject(cams + 9 * obsCam[o], pts + 3 * obsPt[o], obsXy + 2 * o, r);
#pragma unroll
  for (int e = 0; e < 2; e++) {
    res[2 * o + e] = r[e].v;
#pragma unroll
    for (int i = 0; i < 9; i++) Jc[18 * o + 9 * e + i] = r[e].d[i];
#pragma unroll
    for (int i = 0; i < 3; i++) Jp[6 * o + 3 * e + i] = r[e].d[9 + i];
  }
}

// One observation per 64 threads (a wave): H(cam,cam) += Jc^T Jc, H(pt,pt) += Jp^T Jp,
// H(cam,pt) += Jc^T Jp, g += J^T r -- every block located with the device accessor.
// Parameter numbering of the caller: points 0..numPts-1, cameras numPts.. (BaAtLargeBench.cpp:50-57).
template <typename T>
__global__ __launch_bounds__(256) void balFillKernel(PermutedCoalescedAccessor acc, int64_t numPts,
                                                     int64_t numObs, const int64_t* obsCam,
                                                     const int64_t* obsPt, const double* Jc,
                                                     const double* Jp, const double* res, T* data,
                                                     T* grad, int64_t* dbg) {
  const int lane = threadIdx.x & 63;
  const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= numObs) return;
  const int64_t ptId = obsPt[o], camId = numPts + obsCam[o];
  const double* jc = Jc + 18 * o;
  const double* jp = Jp + 6 * o;
  // off-diagonal block (cam, pt): 9 x 3, stored as it is or transposed (flipped)
  const auto off = acc.blockOffset(camId, ptId);
  const int64_t bo = std::get<0>(off), bs = std::get<1>(off);
  const bool flip = std::get<2>(off);
  if (lane < 27) {
    const int i = lane / 3, j = lane % 3;  // (camera row, point column)
    const double v = jc[i] * jp[j] + jc[9 + i] * jp[3 + j];
    T* p = data + bo + (flip ? (int64_t)j * bs + i : (int64_t)i * bs + j);
    unsafeAtomicAdd(p, (T)v);
  }
  // diagonal blocks (lower AND upper triangle written, as Eigen's += on the full block does)
  const auto dc = acc.diagBlockOffset(camId);
  for (int e = lane; e < 81; e += 64) {
    const int i = e / 9, j = e % 9;
    unsafeAtomicAdd(data + dc.first + (int64_t)i * dc.second + j,
                    (T)(jc[i] * jc[j] + jc[9 + i] * jc[9 + j]));
  }
  const auto dp = acc.diagBlockOffset(ptId);
  if (lane < 9) {
    const int i = lane / 3, j = lane % 3;
    unsafeAtomicAdd(data + dp.first + (int64_t)i * dp.second + j,
                    (T)(jp[i] * jp[j] + jp[3 + i] * jp[3 + j]));
  }
  if (grad) {
    const double r0 = res[2 * o], r1 = res[2 * o + 1];
    if (lane < 9) unsafeAtomicAdd(grad + acc.paramStart(camId) + lane, (T)(jc[lane] * r0 + jc[9 + lane] * r1));
    if (lane >= 16 && lane < 19) {
      const int i = lane - 16;
      unsafeAtomicAdd(grad + acc.paramStart(ptId) + i, (T)(jp[i] * r0 + jp[3 + i] * r1));
    }
  }
  if (dbg && lane == 0) {  // what the accessor answered, for the integer parity test
    int64_t* d = dbg + 7 * o;
    d[0] = bo;
    d[1] = bs;
    d[2] = flip ? 1 : 0;
    d[3] = dc.first;
    d[4] = dc.second;
    d[5] = dp.first;
    d[6] = dp.second;
  }
}

// Levenberg-Marquardt damping of every diagonal entry (BaAtLargeOptimizer.cpp:126-130)
template <typename T>
__global__ __launch_bounds__(256) void balDampKernel(PermutedCoalescedAccessor acc, int64_t numParams,
                                                     T lambda, T* data) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= numParams) return;
  const auto d = acc.diagBlockOffset(p);
  const int64_t n = acc.paramSize(p);
  for (int64_t i = 0; i < n; i++) {
    T* e = data + d.first + i * (d.second + 1);
    *e = *e * (T(1) + lambda) + lambda * T(1e-3);
  }
}

}  // namespace

void balLinearize(int64_t numObs, const int64_t* obsCam, const int64_t* obsPt, const double* obsXy,
                  const double* cams, const double* pts, double* res, double* Jc, double* Jp,
                  void* stream) {
  if (numObs <= 0) return;
  balLinearizeKernel<<<dim3((unsigned)((numObs + 63) / 64)), 64, 0, (hipStream_t)stream>>>(
      numObs, obsCam, obsPt, obsXy, cams, pts, res, Jc, Jp);
  balCHECK(hipGetLastError());
}

template <typename T>
void balFillHessian(const PermutedCoalescedAccessor& acc, int64_t numPts, int64_t numCams,
                    int64_t numObs, const int64_t* obsCam, const int64_t* obsPt, const double* Jc,
                    const double* Jp, const double* res, T lambda, T* data, T* grad, int64_t* dbg,
                    void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (numObs > 0) {
    balFillKernel<T><<<dim3((unsigned)((numObs + 3) / 4)), 256, 0, s>>>(
        acc, numPts, numObs, obsCam, obsPt, Jc, Jp, res, data, grad, dbg);
  }
  const int64_t numParams = numPts + numCams;
  if (lambda != T(0) && numParams > 0) {
    balDampKernel<T><<<dim3((unsigned)((numParams + 255) / 256)), 256, 0, s>>>(acc, numParams, lambda,
                                                                              data);
  }
  balCHECK(hipGetLastError());
}

template void balFillHessian<double>(const PermutedCoalescedAccessor&, int64_t, int64_t, int64_t,
                                     const int64_t*, const int64_t*, const double*, const double*,
                                     const double*, double, double*, double*, int64_t*, void*);
template void balFillHessian<float>(const PermutedCoalescedAccessor&, int64_t, int64_t, int64_t,
                                    const int64_t*, const int64_t*, const double*, const double*,
                                    const double*, float, float*, float*, int64_t*, void*);

}  // namespace BaSpaCho
