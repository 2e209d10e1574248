// Device-side caller pipeline of the BAL benchmark (see bal_pipeline.hip): residuals + Jacobians of
// every observation, and Hessian / gradient assembly through the device accessor.
#pragma once

#include <cstdint>

#include "accessor.h"

namespace BaSpaCho {

// res[2 nObs], Jc[nObs][2][9], Jp[nObs][2][3] (all device, fp64); cams[nCams][9], pts[nPts][3]
void balLinearize(int64_t numObs, const int64_t* obsCam, const int64_t* obsPt, const double* obsXy,
                  const double* cams, const double* pts, double* res, double* Jc, double* Jp,
                  void* stream);

// the same in the reference's parameterisation (BaAtLarge.h:56-150): camera = SE(3) tangent of a left
// perturbation (translation, rotation), calibration fixed; Jc[nObs][2][6]
void balLinearizeSe3(int64_t numObs, const int64_t* obsCam, const int64_t* obsPt, const double* obsXy,
                     const double* cams, const double* pts, double* res, double* Jc, double* Jp,
                     void* stream);

// data += J^T J block by block (data must have been zeroed by the caller), grad += J^T r (optional),
// then every diagonal entry d <- d (1 + lambda) + 1e-3 lambda.  acc = Solver::deviceAccessor();
// caller's parameter numbering: points first, cameras after.  dbg (optional, 7 int64 per
// observation): the offsets / strides / flip the accessor returned.
// camSize: 9 (BAL file parameters, Jc[.][2][9]) or 6 (SE3 tangent, Jc[.][2][6])
template <typename T>
void balFillHessian(const PermutedCoalescedAccessor& acc, int camSize, int64_t numPts, int64_t numCams,
                    int64_t numObs, const int64_t* obsCam, const int64_t* obsPt, const double* Jc,
                    const double* Jp, const double* res, T lambda, T* data, T* grad, int64_t* dbg,
                    void* stream);

}  // namespace BaSpaCho
