// Block sparsity pattern in compressed form (ptrs/inds); each entry stands for a block.
// Host-side, dependency-free restatement of the reference's symbolic helper
// (behaviour: baspacho/baspacho/SparseStructure.h:19-56, SparseStructure.cpp:24-375).
#pragma once

#include <cstdint>
#include <vector>

namespace BaSpaCho {

struct SparseStructure {
  std::vector<int64_t> ptrs;
  std::vector<int64_t> inds;

  SparseStructure() {}
  SparseStructure(std::vector<int64_t>&& ptrs_, std::vector<int64_t>&& inds_)
      : ptrs(std::move(ptrs_)), inds(std::move(inds_)) {}
  SparseStructure(const std::vector<int64_t>& ptrs_, const std::vector<int64_t>& inds_)
      : ptrs(ptrs_), inds(inds_) {}

  int64_t order() const { return (int64_t)ptrs.size() - 1; }

  // sort the indices inside every row
  void sortIndices();

  // transpose (square matrix); output rows come out sorted
  SparseStructure transpose() const;

  // drop the strictly lower (clearLower=true) or strictly upper half
  SparseStructure clear(bool clearLower = true) const;

  // symmetric permutation of a half-stored pattern: row i moves to mapPerm[i];
  // result is the lower (lowerHalf) / upper half in csc (= upper / lower half in csr)
  SparseStructure symmetricPermutation(const std::vector<int64_t>& mapPerm, bool lowerHalf = true,
                                       bool sortIndices = true) const;

  // csr lower-half input: add the fill produced by eliminating the mutually independent
  // parameters [start,end)
  SparseStructure addIndependentEliminationFill(int64_t start, int64_t end,
                                                bool sortIdx = true) const;

  // csr lower-half input: pattern of the complete Cholesky factor
  SparseStructure addFullEliminationFill() const;

  // perm[i] = old index that should move to position i (approximate minimum degree)
  // contractChains: createSolver's variant, odd-even rounds on chain-like parts first (min_degree.h)
  std::vector<int64_t> fillReducingPermutation(bool contractChains = false) const;

  // pattern of the trailing principal sub-matrix from `start`
  SparseStructure extractRightBottom(int64_t start) const;
};

}  // namespace BaSpaCho
