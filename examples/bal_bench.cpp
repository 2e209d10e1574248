// A C++ caller of the reference, compiled against THIS library's headers: the BAL_bench flow of
// benchmarking/BaAtLargeBench.cpp:44-97 (testSolvers: points first, cameras after, one block per
// observation, sparse elimination range {0, numPts}; mock data = uniform(-1,1) + damp(0, 1.2 order);
// heat-up factor, timed factor) with the GPU backend, followed by what the LM optimizer does with
// the accessor (BaAtLargeOptimizer.cpp:100-131): blocks are written through accessor().block<9,3>()
// / diagBlock<3>() views on the host copy, the system is factored and solved on the device, and the
// residual b - A x is evaluated block by block through the same views.
//
// Only the include path and the backend enum differ from a caller of the reference.  The BAL data
// file is replaced by a synthetic structure (no dataset offline).  Exit code 0 = residual OK.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I. examples/bal_bench.cpp \
//         -Lbaspacho_amd -lbaspacho_amd -Wl,-rpath,'$ORIGIN/../baspacho_amd' -o examples/bal_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "baspacho_amd/csrc/solver.h"  // was: baspacho/baspacho/Solver.h

using namespace BaSpaCho;
using hrc = std::chrono::high_resolution_clock;

#define HIP_OK(x)                                                          \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
      return 2;                                                            \
    }                                                                      \
  } while (0)

struct Obs {
  int64_t camIdx, ptIdx;
};

// splitmix64: uniform doubles in [0, 1)
static double unit(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return double((z ^ (z >> 31)) >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
  const int64_t numCams = argc > 1 ? std::atoll(argv[1]) : 120;
  const int64_t numPts = argc > 2 ? std::atoll(argv[2]) : 40000;
  const int nPointParams = 3, nCameraParams = 9;
  uint64_t rng = 37;

  // observations: every point is seen by 2..6 cameras around a centre camera
  std::vector<Obs> observations;
  for (int64_t p = 0; p < numPts; p++) {
    const int64_t centre = (int64_t)(unit(rng) * numCams);
    const int track = 2 + (int)(unit(rng) * 5);
    for (int t = 0; t < track; t++) {
      int64_t c = centre + (int64_t)((unit(rng) * 2 - 1) * 12);
      c = c < 0 ? 0 : (c >= numCams ? numCams - 1 : c);
      observations.push_back({c, p});
    }
  }

  // ---- testSolvers (BaAtLargeBench.cpp:44-73), verbatim in structure
  const int64_t totNumParams = numPts + numCams;
  std::vector<int64_t> paramSize(totNumParams);
  std::vector<std::set<int64_t>> colBlocks(totNumParams);
  for (int64_t i = 0; i < numPts; i++) {  // points go first
    paramSize[i] = nPointParams;
    colBlocks[i].insert(i);
  }
  for (int64_t i = numPts; i < totNumParams; i++) {  // then cams
    paramSize[i] = nCameraParams;
    colBlocks[i].insert(i);
  }
  for (auto& obs : observations) colBlocks[obs.ptIdx].insert(numPts + obs.camIdx);
  // columnsToCscStruct(colBlocks).transpose(): csr of the lower triangle
  SparseStructure csc;
  csc.ptrs.push_back(0);
  for (auto& col : colBlocks) {
    csc.inds.insert(csc.inds.end(), col.begin(), col.end());
    csc.ptrs.push_back((int64_t)csc.inds.size());
  }
  SparseStructure origSs = csc.transpose();

  auto startAnalysis = hrc::now();
  Settings settings;
  settings.backend = BackendHip;  // was: {.numThreads = 16} / BackendCuda
  auto solver = createSolver(settings, paramSize, origSs, {0, numPts});
  const double analysisTime = std::chrono::duration<double>(hrc::now() - startAnalysis).count();

  // ---- numeric data through the block views, as BaAtLargeOptimizer.cpp:119-129 fills the Hessian:
  // camera-point blocks 9x3, point diagonal 3x3, camera diagonal 9x9; then damp like the bench
  std::vector<double> matData(solver->dataSize(), 0.0);
  auto acc = solver->accessor();
  for (auto& obs : observations) {
    auto blk = acc.block<9, 3>(matData.data(), numPts + obs.camIdx, obs.ptIdx);
    blk += [&](int64_t, int64_t) { return unit(rng) * 2 - 1; };
    auto dp = acc.diagBlock<3>(matData.data(), obs.ptIdx);
    auto dc = acc.diagBlock<9>(matData.data(), numPts + obs.camIdx);
    for (int i = 0; i < 3; i++) dp(i, i) += 1.0;
    for (int i = 0; i < 9; i++) dc(i, i) += 1.0;
  }
  solver->skel().damp(matData, double(0), double(solver->order() * 1.2));
  const std::vector<double> A = matData;  // the un-factored matrix, for the residual

  double* dev = nullptr;
  HIP_OK(hipMalloc((void**)&dev, matData.size() * sizeof(double)));
  HIP_OK(hipMemcpy(dev, matData.data(), matData.size() * sizeof(double), hipMemcpyHostToDevice));
  solver->factor(dev);  // heat up
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(dev, matData.data(), matData.size() * sizeof(double), hipMemcpyHostToDevice));
  auto startFactor = hrc::now();
  solver->factor(dev);
  HIP_OK(hipDeviceSynchronize());
  const double factorTime = std::chrono::duration<double>(hrc::now() - startFactor).count();

  // ---- solve A x = b on the device, residual through the views on the host
  const int64_t order = solver->order();
  std::vector<double> b(order), x(order);
  for (auto& v : b) v = unit(rng) * 2 - 1;
  double* dvec = nullptr;
  HIP_OK(hipMalloc((void**)&dvec, order * sizeof(double)));
  HIP_OK(hipMemcpy(dvec, b.data(), order * sizeof(double), hipMemcpyHostToDevice));
  solver->solve(dev, dvec, order, 1);
  HIP_OK(hipMemcpy(x.data(), dvec, order * sizeof(double), hipMemcpyDeviceToHost));
  // vectors are in the solver's internal order: parameter i starts at acc.paramStart(i)
  std::vector<double> r = b;
  auto applyBlock = [&](int64_t rowP, int64_t colP) {
    auto blk = acc.block(const_cast<double*>(A.data()), rowP, colP);
    const int64_t r0 = acc.paramStart(rowP), c0 = acc.paramStart(colP);
    for (int64_t i = 0; i < blk.rows(); i++) {
      for (int64_t j = 0; j < blk.cols(); j++) {
        if (rowP == colP && j > i) continue;  // diagonal blocks: lower triangle, mirrored below
        r[r0 + i] -= blk(i, j) * x[c0 + j];
        if (rowP != colP || i != j) r[c0 + j] -= blk(i, j) * x[r0 + i];
      }
    }
  };
  for (int64_t p = 0; p < totNumParams; p++) applyBlock(p, p);
  for (int64_t p = 0; p < numPts; p++) {
    for (int64_t rowP : colBlocks[p]) {
      if (rowP != p) applyBlock(rowP, p);
    }
  }
  double rn = 0, bn = 0;
  for (int64_t i = 0; i < order; i++) {
    rn += r[i] * r[i];
    bn += b[i] * b[i];
  }
  const double rel = std::sqrt(rn / bn);
  std::printf("cams %lld pts %lld obs %zu order %lld\n", (long long)numCams, (long long)numPts,
              observations.size(), (long long)order);
  std::printf("Total Analysis Time..: %.3f ms\n", analysisTime * 1e3);
  std::printf("Total Factor Time....: %.3f ms (%.1f GF/s)\n", factorTime * 1e3,
              solver->factorFlops() / factorTime / 1e9);
  std::printf("relative residual |b - A x| / |b| = %.3e\n", rel);
  (void)hipFree(dev);
  (void)hipFree(dvec);
  if (!(rel < 1e-10)) {
    std::printf("FAILED\n");
    return 1;
  }
  std::printf("BAL_BENCH_OK\n");
  return 0;
}
