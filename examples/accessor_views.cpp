// Host-only check of the accessor's block()/diagBlock() views (csrc/accessor.h; the reference's
// Accessor.h:69-107,165-200): a random block-sparse symmetric matrix is written into the numeric
// data THROUGH the views of the permuted accessor (flipped blocks included), and the result of
// CoalescedBlockMatrixSkel::densify must be exactly the matrix assembled directly, under the
// solver's permutation.  No device call: createSolver is host code.  Exit code 0 = identical.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I. examples/accessor_views.cpp \
//         -Lbaspacho_amd -lbaspacho_amd -Wl,-rpath,'$ORIGIN/../baspacho_amd' -o examples/accessor_views
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "baspacho_amd/csrc/solver.h"

using namespace BaSpaCho;

static double unit(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return double((z ^ (z >> 31)) >> 11) * (1.0 / 9007199254740992.0);
}

int main() {
  uint64_t rng = 11;
  const int64_t n = 60;
  std::vector<int64_t> paramSize(n);
  std::vector<int64_t> start(n + 1, 0);
  for (int64_t i = 0; i < n; i++) {
    paramSize[i] = 1 + (int64_t)(unit(rng) * 4);
    start[i + 1] = start[i] + paramSize[i];
  }
  std::vector<std::set<int64_t>> colBlocks(n);
  for (int64_t i = 0; i < n; i++) {
    colBlocks[i].insert(i);
    for (int64_t j = i + 1; j < n; j++) {
      if (unit(rng) < 0.08) colBlocks[i].insert(j);
    }
  }
  SparseStructure csc;
  csc.ptrs.push_back(0);
  for (auto& col : colBlocks) {
    csc.inds.insert(csc.inds.end(), col.begin(), col.end());
    csc.ptrs.push_back((int64_t)csc.inds.size());
  }
  SparseStructure ss = csc.transpose();
  Settings settings;
  auto solver = createSolver(settings, paramSize, ss);
  const int64_t order = solver->order();
  auto acc = solver->accessor();
  const auto& perm = solver->paramToSpan();

  std::vector<double> data(solver->dataSize(), 0.0);
  std::vector<double> dense((size_t)order * order, 0.0);  // in the solver's INTERNAL order
  int flips = 0;
  for (int64_t c = 0; c < n; c++) {
    for (int64_t r : colBlocks[c]) {
      const int64_t r0 = acc.paramStart(r), c0 = acc.paramStart(c);
      if (r == c) {
        auto d = acc.diagBlock(data.data(), c);
        for (int64_t i = 0; i < d.rows(); i++) {
          for (int64_t j = 0; j <= i; j++) {
            const double v = unit(rng);
            d(i, j) = v;
            dense[(r0 + i) * order + c0 + j] = v;
            dense[(c0 + j) * order + r0 + i] = v;
          }
        }
      } else {
        auto b = acc.block(data.data(), r, c);
        if (std::get<2>(acc.blockOffset(r, c))) flips++;
        if (b.rows() != paramSize[r] || b.cols() != paramSize[c]) return 3;
        for (int64_t i = 0; i < b.rows(); i++) {
          for (int64_t j = 0; j < b.cols(); j++) {
            const double v = unit(rng);
            b(i, j) = v;
            dense[(r0 + i) * order + c0 + j] = v;
            dense[(c0 + j) * order + r0 + i] = v;
          }
        }
        // transpose() is the view of block (c, r)
        auto bt = acc.block(data.data(), c, r);
        if (bt.rows() != b.cols() || &bt(0, 0) != &b(0, 0) || (b.rows() > 1 && &bt(0, 1) != &b(1, 0))) return 4;
      }
    }
  }
  std::vector<double> got;
  int64_t denseOrder = 0;
  solver->skel().densify(got, denseOrder, data.data(), /*fillUpperHalf=*/true);
  if (denseOrder != order) return 5;
  // densify is column-major (like the reference's Eigen default); the matrix is symmetric
  size_t bad = 0;
  for (size_t k = 0; k < dense.size(); k++) bad += dense[k] != got[k];
  // a compile-time size that does not match must throw, as BASPACHO_CHECK_EQ does
  bool threw = false;
  try {
    int64_t p = 0;
    while (paramSize[p] == 7) p++;
    (void)acc.diagBlock<7>(data.data(), p);
  } catch (const std::exception&) {
    threw = true;
  }
  std::printf("params %lld order %lld flipped blocks %d mismatches %zu size-check %s perm[0]=%lld\n",
              (long long)n, (long long)order, flips, bad, threw ? "throws" : "MISSING", (long long)perm[0]);
  if (bad || !threw || flips == 0) return 1;
  std::printf("ACCESSOR_VIEWS_OK\n");
  return 0;
}
