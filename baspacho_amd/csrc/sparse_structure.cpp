#include "sparse_structure.h"

#include <algorithm>
#include <cstdlib>

#include "bsp_utils.h"
#include "min_degree.h"

namespace BaSpaCho {

namespace {

// Generic two-pass "bucket" builder: emit(f) must call f(bucket, value) for every entry and
// produce the same sequence on both passes.
template <typename Emit>
SparseStructure bucketize(int64_t numBuckets, Emit&& emit) {
  SparseStructure out;
  out.ptrs.assign(numBuckets + 1, 0);
  emit([&](int64_t b, int64_t) { out.ptrs[b]++; });
  int64_t total = cumSumVec(out.ptrs);
  out.inds.resize(total);
  std::vector<int64_t> cursor(out.ptrs.begin(), out.ptrs.end() - 1);
  emit([&](int64_t b, int64_t v) { out.inds[cursor[b]++] = v; });
  return out;
}

}  // namespace

void SparseStructure::sortIndices() {
  for (int64_t i = 0, n = order(); i < n; i++) {
    std::sort(inds.begin() + ptrs[i], inds.begin() + ptrs[i + 1]);
  }
}

SparseStructure SparseStructure::transpose() const {
  const int64_t n = order();
  return bucketize(n, [&](auto&& put) {
    for (int64_t i = 0; i < n; i++) {
      for (int64_t k = ptrs[i]; k < ptrs[i + 1]; k++) {
        BASPACHO_CHECK_LT(inds[k], n);
        put(inds[k], i);
      }
    }
  });
}

SparseStructure SparseStructure::clear(bool clearLower) const {
  const int64_t n = order();
  return bucketize(n, [&](auto&& put) {
    for (int64_t i = 0; i < n; i++) {
      for (int64_t k = ptrs[i]; k < ptrs[i + 1]; k++) {
        int64_t j = inds[k];
        BASPACHO_CHECK_LT(j, n);
        bool dropped = (j != i) && ((j > i) == clearLower);
        if (!dropped) put(i, j);
      }
    }
  });
}

SparseStructure SparseStructure::symmetricPermutation(const std::vector<int64_t>& mapPerm,
                                                      bool lowerHalf, bool sortIdx) const {
  const int64_t n = order();
  BASPACHO_CHECK_EQ(n, (int64_t)mapPerm.size());
  SparseStructure out = bucketize(n, [&](auto&& put) {
    for (int64_t i = 0; i < n; i++) {
      int64_t pi = mapPerm[i];
      BASPACHO_CHECK_LT(pi, n);
      for (int64_t k = ptrs[i]; k < ptrs[i + 1]; k++) {
        BASPACHO_CHECK_LT(inds[k], n);
        int64_t pj = mapPerm[inds[k]];
        BASPACHO_CHECK_LT(pj, n);
        int64_t lo = std::min(pi, pj), hi = std::max(pi, pj);
        if (lowerHalf) {
          put(lo, hi);  // column = smaller index, row = larger
        } else {
          put(hi, lo);
        }
      }
    }
  });
  if (sortIdx) out.sortIndices();
  return out;
}

SparseStructure SparseStructure::addIndependentEliminationFill(int64_t elimStart, int64_t elimEnd,
                                                               bool sortIdx) const {
  const int64_t n = order();
  if (elimEnd == n) return *this;  // nothing below the eliminated set

  // columns of the eliminated parameters (rows sorted by construction of transpose())
  SparseStructure cols = transpose();

  SparseStructure out;
  out.ptrs.assign(ptrs.begin(), ptrs.begin() + elimEnd + 1);
  out.inds.assign(inds.begin(), inds.begin() + ptrs[elimEnd]);
  out.ptrs.reserve(ptrs.size());

  std::vector<int64_t> seenInRow(n, -1);
  for (int64_t row = elimEnd; row < n; row++) {
    auto mark = [&](int64_t c) {
      if (seenInRow[c] != row) {
        seenInRow[c] = row;
        out.inds.push_back(c);
      }
    };
    mark(row);
    for (int64_t q = ptrs[row]; q < ptrs[row + 1]; q++) {
      int64_t c = inds[q];
      if (c >= row) continue;
      mark(c);
      if (c >= elimStart && c < elimEnd) {
        // eliminating c connects `row` to every earlier row of column c
        for (int64_t t = cols.ptrs[c]; t < cols.ptrs[c + 1]; t++) {
          int64_t w = cols.inds[t];
          if (w >= row) break;
          mark(w);
        }
      }
    }
    out.ptrs.push_back((int64_t)out.inds.size());
  }
  if (sortIdx) out.sortIndices();
  return out;
}

SparseStructure SparseStructure::addFullEliminationFill() const {
  const int64_t n = order();
  // Row-by-row symbolic factorisation with an on-the-fly elimination tree (Liu): the pattern
  // of row k of L is the union of the tree paths from each A(k,i), i<k, up to k.
  std::vector<int64_t> parent(n, -1), stamp(n, -1);
  std::vector<std::vector<int64_t>> rows(n);
  for (int64_t k = 0; k < n; k++) {
    stamp[k] = k;
    rows[k].push_back(k);
    for (int64_t q = ptrs[k]; q < ptrs[k + 1]; q++) {
      int64_t i = inds[q];
      if (i >= k) continue;
      while (stamp[i] != k) {
        if (parent[i] < 0) parent[i] = k;
        stamp[i] = k;
        rows[k].push_back(i);
        i = parent[i];
      }
    }
    std::sort(rows[k].begin(), rows[k].end());
  }
  SparseStructure out;
  out.ptrs.resize(n + 1);
  int64_t total = 0;
  for (int64_t k = 0; k < n; k++) {
    out.ptrs[k] = total;
    total += (int64_t)rows[k].size();
  }
  out.ptrs[n] = total;
  out.inds.reserve(total);
  for (int64_t k = 0; k < n; k++) out.inds.insert(out.inds.end(), rows[k].begin(), rows[k].end());
  return out;
}

std::vector<int64_t> SparseStructure::fillReducingPermutation(bool contractChains) const {
  // 50 = the smallest set elimination_tree.cpp turns into a sparse-elimination range
  return minimumDegreeOrdering(ptrs, inds, contractChains ? 50 : 0);
}

SparseStructure SparseStructure::extractRightBottom(int64_t start) const {
  const int64_t n = order();
  BASPACHO_CHECK_GE(start, 0);
  BASPACHO_CHECK_LE(start, n);
  return bucketize(n - start, [&](auto&& put) {
    for (int64_t i = start; i < n; i++) {
      for (int64_t k = ptrs[i]; k < ptrs[i + 1]; k++) {
        BASPACHO_CHECK_LT(inds[k], n);
        if (inds[k] >= start) put(i - start, inds[k] - start);
      }
    }
  });
}

}  // namespace BaSpaCho
