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

// The REFERENCE's parameterisation (benchmarking/BaAtLarge.h:56-150, Cost::compute_residual as the
// LM optimizer calls it, BaAtLargeOptimizer.cpp:108-124): the camera is T_W_C in SE(3) (camPt = R X +
// t) with FIXED calibration (f, k1, k2); its 6 parameters are the tangent of a LEFT perturbation,
// translation first, T <- exp(delta) T, and the optimizer applies T <- exp(-step) T
// (BaAtLargeOptimizer.cpp:176-183).  Jacobians in closed form, exactly the expressions of the
// reference: with p = -camPt.xy / camPt.z, s = |p|^2, r = 1 + (k1 + k2 s) s,
//   D = d p / d(.) ,   J = f r D + p (2 p^T D) f (k1 + 2 k2 s),
// D_point = [(z R_0 - x R_2); (z R_1 - y R_2)] * (-1 / z^2),  D_cam = the 2 x 6 matrix below.
// A point in front of the image plane (camPt.z > 0.01; BAL cameras look down -z) gives the fixed
// residual (25, 0) and zero Jacobians, as in the reference.
__device__ __forceinline__ void rodriguesMatrix(const double* w, double (&R)[3][3]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double a, b;  // R = I + a [w]x + b [w]x^2
  if (th2 > 1e-20) {
    const double th = sqrt(th2);
    a = sin(th) / th;
    b = (1.0 - cos(th)) / th2;
  } else {
    a = 1.0;
    b = 0.5;
  }
  const double K[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double k2 = 0;
#pragma unroll
      for (int l = 0; l < 3; l++) k2 += K[i][l] * K[l][j];
      R[i][j] = (i == j ? 1.0 : 0.0) + a * K[i][j] + b * k2;
    }
  }
}

__global__ __launch_bounds__(64) void balLinearizeSe3Kernel(int64_t numObs, const int64_t* obsCam,
                                                            const int64_t* obsPt, const double* obsXy,
                                                            const double* cams, const double* pts,
                                                            double* res, double* Jc, double* Jp) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= numObs) return;
  const double* cam = cams + 9 * obsCam[o];
  const double* X = pts + 3 * obsPt[o];
  double R[3][3];
  rodriguesMatrix(cam, R);
  double P[3];
#pragma unroll
  for (int i = 0; i < 3; i++) P[i] = R[i][0] * X[0] + R[i][1] * X[1] + R[i][2] * X[2] + cam[3 + i];
  double* jc = Jc + 12 * o;
  double* jp = Jp + 6 * o;
  if (P[2] > 0.01) {
    res[2 * o] = 25.0;
    res[2 * o + 1] = 0.0;
#pragma unroll
    for (int i = 0; i < 12; i++) jc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) jp[i] = 0.0;
    return;
  }
  const double f = cam[6], k1 = cam[7], k2 = cam[8];
  const double p0 = -P[0] / P[2], p1 = -P[1] / P[2];
  const double sq = p0 * p0 + p1 * p1;
  const double r = 1.0 + (k1 + k2 * sq) * sq;
  res[2 * o] = f * r * p0 - obsXy[2 * o];
  res[2 * o + 1] = f * r * p1 - obsXy[2 * o + 1];
  const double g = f * (k1 + k2 * 2.0 * sq);
  // point Jacobian
  {
    const double denum = -1.0 / (P[2] * P[2]);
    double D[2][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      D[0][c] = (P[2] * R[0][c] - P[0] * R[2][c]) * denum;
      D[1][c] = (P[2] * R[1][c] - P[1] * R[2][c]) * denum;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double dsq = 2.0 * (p0 * D[0][c] + p1 * D[1][c]);
      jp[c] = f * r * D[0][c] + p0 * dsq * g;
      jp[3 + c] = f * r * D[1][c] + p1 * dsq * g;
    }
  }
  // camera Jacobian (translation, rotation)
  {
    const double dz = 1.0 / P[2], xdz = P[0] * dz, ydz = P[1] * dz, xydz2 = xdz * ydz;
    const double D[2][6] = {{-dz, 0.0, xdz * dz, xydz2, -1.0 - xdz * xdz, ydz},
                            {0.0, -dz, ydz * dz, 1.0 + ydz * ydz, -xydz2, -xdz}};
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const double dsq = 2.0 * (p0 * D[0][c] + p1 * D[1][c]);
      jc[c] = f * r * D[0][c] + p0 * dsq * g;
      jc[6 + c] = f * r * D[1][c] + p1 * dsq * g;
    }
  }
}

// One observation per 64 threads (a wave): H(cam,cam) += Jc^T Jc, H(pt,pt) += Jp^T Jp,
// H(cam,pt) += Jc^T Jp, g += J^T r -- every block located with the device accessor.
// Parameter numbering of the caller: points 0..numPts-1, cameras numPts.. (BaAtLargeBench.cpp:50-57).
template <typename T>
__global__ __launch_bounds__(256) void balFillKernel(PermutedCoalescedAccessor acc, int camSize,
                                                     int64_t numPts, int64_t numCams,
                                                     int64_t numObs, const int64_t* obsCam,
                                                     const int64_t* obsPt, const double* Jc,
                                                     const double* Jp, const double* res, T* data,
                                                     T* grad, int64_t* dbg) {
  const int lane = threadIdx.x & 63;
  const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= numObs) return;
  // (an observation whose indices fall outside [0, numPts) x [0, numCams) is skipped: its atomics
  //  would land outside the numeric data; the host-side loader rejects such files, bal.py)
  if (obsPt[o] < 0 || obsPt[o] >= numPts || obsCam[o] < 0 || obsCam[o] >= numCams) return;
  const int64_t ptId = obsPt[o], camId = numPts + obsCam[o];
  const int CS = camSize;  // camera block: 9 (BAL file parameters) or 6 (SE3 tangent)
  const double* jc = Jc + 2 * CS * o;
  const double* jp = Jp + 6 * o;
  // off-diagonal block (cam, pt): 9 x 3, stored as it is or transposed (flipped)
  const auto off = acc.blockOffset(camId, ptId);
  const int64_t bo = std::get<0>(off), bs = std::get<1>(off);
  const bool flip = std::get<2>(off);
  if (lane < 3 * CS) {
    const int i = lane / 3, j = lane % 3;  // (camera row, point column)
    const double v = jc[i] * jp[j] + jc[CS + i] * jp[3 + j];
    T* p = data + bo + (flip ? (int64_t)j * bs + i : (int64_t)i * bs + j);
    unsafeAtomicAdd(p, (T)v);
  }
  // diagonal blocks (lower AND upper triangle written, as Eigen's += on the full block does)
  const auto dc = acc.diagBlockOffset(camId);
  for (int e = lane; e < CS * CS; e += 64) {
    const int i = e / CS, j = e % CS;
    unsafeAtomicAdd(data + dc.first + (int64_t)i * dc.second + j,
                    (T)(jc[i] * jc[j] + jc[CS + i] * jc[CS + j]));
  }
  const auto dp = acc.diagBlockOffset(ptId);
  if (lane < 9) {
    const int i = lane / 3, j = lane % 3;
    unsafeAtomicAdd(data + dp.first + (int64_t)i * dp.second + j,
                    (T)(jp[i] * jp[j] + jp[3 + i] * jp[3 + j]));
  }
  if (grad) {
    const double r0 = res[2 * o], r1 = res[2 * o + 1];
    if (lane < CS) unsafeAtomicAdd(grad + acc.paramStart(camId) + lane, (T)(jc[lane] * r0 + jc[CS + lane] * r1));
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

void balLinearizeSe3(int64_t numObs, const int64_t* obsCam, const int64_t* obsPt, const double* obsXy,
                     const double* cams, const double* pts, double* res, double* Jc, double* Jp,
                     void* stream) {
  if (numObs <= 0) return;
  balLinearizeSe3Kernel<<<dim3((unsigned)((numObs + 63) / 64)), 64, 0, (hipStream_t)stream>>>(
      numObs, obsCam, obsPt, obsXy, cams, pts, res, Jc, Jp);
  balCHECK(hipGetLastError());
}

template <typename T>
void balFillHessian(const PermutedCoalescedAccessor& acc, int camSize, int64_t numPts, int64_t numCams,
                    int64_t numObs, const int64_t* obsCam, const int64_t* obsPt, const double* Jc,
                    const double* Jp, const double* res, T lambda, T* data, T* grad, int64_t* dbg,
                    void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (numObs > 0) {
    balFillKernel<T><<<dim3((unsigned)((numObs + 3) / 4)), 256, 0, s>>>(
        acc, camSize, numPts, numCams, numObs, obsCam, obsPt, Jc, Jp, res, data, grad, dbg);
  }
  const int64_t numParams = numPts + numCams;
  if (lambda != T(0) && numParams > 0) {
    balDampKernel<T><<<dim3((unsigned)((numParams + 255) / 256)), 256, 0, s>>>(acc, numParams, lambda,
                                                                              data);
  }
  balCHECK(hipGetLastError());
}

template void balFillHessian<double>(const PermutedCoalescedAccessor&, int, int64_t, int64_t, int64_t,
                                     const int64_t*, const int64_t*, const double*, const double*,
                                     const double*, double, double*, double*, int64_t*, void*);
template void balFillHessian<float>(const PermutedCoalescedAccessor&, int, int64_t, int64_t, int64_t,
                                    const int64_t*, const int64_t*, const double*, const double*,
                                    const double*, float, float*, float*, int64_t*, void*);

}  // namespace BaSpaCho
