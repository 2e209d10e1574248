// Polynomial timing models of the primitive factor operations, used only to decide supernode
// merges in the elimination tree (behaviour: baspacho/baspacho/ComputationModel.h:20-112,
// consumer EliminationTree.cpp:231-240).  No Eigen: plain arrays.
#pragma once

#include <array>

namespace BaSpaCho {

struct LinCost {  // cost as a linear function  c0 + c1 * x
  double c0 = 0, c1 = 0;
  LinCost& operator+=(const LinCost& o) {
    c0 += o.c0;
    c1 += o.c1;
    return *this;
  }
  LinCost& operator-=(const LinCost& o) {
    c0 -= o.c0;
    c1 -= o.c1;
    return *this;
  }
};

struct ComputationModel {
  ComputationModel() {}
  ComputationModel(const std::array<double, 4>& potrfParams_,
                   const std::array<double, 6>& trsmParams_,
                   const std::array<double, 6>& sygeParams_,
                   const std::array<double, 4>& asmblParams_)
      : potrfParams(potrfParams_),
        trsmParams(trsmParams_),
        sygeParams(sygeParams_),
        asmblParams(asmblParams_) {}

  // t(potrf n)        ~ a + b n + c n^2 + d n^3
  double potrfEst(double n) const {
    const auto& p = potrfParams;
    return p[0] + n * (p[1] + n * (p[2] + n * p[3]));
  }
  // t(trsm n,k)       ~ a + b n + c n^2 + (d + e n + f n^2) k
  double trsmEst(double n, double k) const {
    const auto& p = trsmParams;
    return p[0] + n * (p[1] + n * p[2]) + k * (p[3] + n * (p[4] + n * p[5]));
  }
  // t(syrk/gemm m,n,k) symmetric in m,n: a + b u + c v + k (d + e u + f v), u=m+n, v=mn
  double sygeEst(double m, double n, double k) const {
    LinCost l = sygeLinEst(m, n);
    return l.c0 + l.c1 * k;
  }
  // t(assemble br,bc) ~ a + b br + c bc + d br bc
  double asmblEst(double br, double bc) const {
    LinCost l = asmblLinEst(br);
    return l.c0 + l.c1 * bc;
  }
  // syrk/gemm cost as a linear function of the node size k
  LinCost sygeLinEst(double m, double n) const {
    const auto& p = sygeParams;
    double u = m + n, v = m * n;
    return {p[0] + u * p[1] + v * p[2], p[3] + u * p[4] + v * p[5]};
  }
  // assemble cost as a linear function of the number of column blocks
  LinCost asmblLinEst(double br) const {
    const auto& p = asmblParams;
    return {p[0] + br * p[1], p[2] + br * p[3]};
  }

  std::array<double, 4> potrfParams{};
  std::array<double, 6> trsmParams{};
  std::array<double, 6> sygeParams{};
  std::array<double, 4> asmblParams{};

  // fitted-constant sets published by the reference (ComputationModel.cpp:12-31), kept for
  // callers that name them; neither describes an MI355X.
  static const ComputationModel model_OpenBlas_i7_1185g7;
  static const ComputationModel model_Cuda117_2080Ti;
  // model for the level-scheduled HIP backend on MI355X ("merge model")
  static const ComputationModel model_Hip_MI355X;
};

}  // namespace BaSpaCho
