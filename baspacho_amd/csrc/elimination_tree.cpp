#include "elimination_tree.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <queue>

#include "bsp_utils.h"

namespace BaSpaCho {

using std::vector;

// thresholds of the sparse-elimination detector (reference EliminationTree.cpp:133-134)
static constexpr int64_t kMaxSparseElimNodeSize = 12;
static constexpr int64_t kMinNumSparseElimNodes = 50;

EliminationTree::EliminationTree(const vector<int64_t>& paramSize_, const SparseStructure& ss_,
                                 const ComputationModel* compMod_)
    : paramSize(paramSize_),
      ss(ss_),
      compMod(compMod_ ? *compMod_ : ComputationModel::model_Hip_MI355X) {
  BASPACHO_CHECK_EQ(paramSize.size() + 1, ss.ptrs.size());
}

void EliminationTree::buildTree() {
  const int64_t n = ss.order();
  parent.assign(n, -1);
  nodeSize = paramSize;
  nodeRows.assign(n, 0);
  nodeRowBlocks.assign(n, 0);
  perColNodes.assign(n, {});

  // Liu's row-subtree walk: row k of L = union of etree paths from each A(k,i), i<k.
  vector<int64_t> visitedBy(n, -1);
  for (int64_t k = 0; k < n; k++) {
    visitedBy[k] = k;
    for (int64_t q = ss.ptrs[k]; q < ss.ptrs[k + 1]; q++) {
      int64_t i = ss.inds[q];
      if (i >= k) continue;
      while (visitedBy[i] != k) {
        if (parent[i] < 0) parent[i] = k;
        visitedBy[i] = k;
        nodeRows[i] += paramSize[k];  // L(k,i) is a nonzero block
        nodeRowBlocks[i]++;
        perColNodes[i].push_back(k);
        i = parent[i];
      }
    }
  }

  // per-column modelled cost of the syrk/gemm + assemble calls, and the per-row view of the
  // same blocks (needed to update those costs incrementally when nodes merge)
  sygeCosts.assign(n, LinCost{});
  asmblCosts.assign(n, LinCost{});
  perRowNodeStats.assign(n, {});
  for (int64_t col = 0; col < n; col++) {
    auto& rowsOfCol = perColNodes[col];
    rowsOfCol.push_back(col);
    std::sort(rowsOfCol.begin(), rowsOfCol.end());

    int64_t rowsBelow = 0, blocksBelow = 0;
    LinCost syge, asmbl;
    for (auto it = rowsOfCol.rbegin(); it != rowsOfCol.rend(); ++it) {
      int64_t row = *it, sz = paramSize[row];
      syge += compMod.sygeLinEst(double(rowsBelow + sz), double(sz));
      asmbl += compMod.asmblLinEst(double(blocksBelow + 1));
      perRowNodeStats[row].push_back(NodeStats{col, 1, sz, blocksBelow, rowsBelow});
      rowsBelow += sz;
      blocksBelow++;
    }
    sygeCosts[col] = syge;
    asmblCosts[col] = asmbl;
  }
}

void EliminationTree::computeNodeHeights(const vector<int64_t>& noCrossPoints) {
  const int64_t n = ss.order();
  unmergedHeightNode.resize(n);
  forbidMerge.assign(n, false);

  vector<int64_t> height(n, 0);
  const size_t numRanges = noCrossPoints.size() + 1;
  for (size_t r = 0; r < numRanges; r++) {
    int64_t begin = r == 0 ? 0 : noCrossPoints[r - 1];
    int64_t end = r < noCrossPoints.size() ? noCrossPoints[r] : n;
    for (int64_t k = begin; k < end; k++) {
      unmergedHeightNode[k] = std::make_tuple(height[k], nodeSize[k], k);
      int64_t par = parent[k];
      if (par < 0) continue;
      if (par >= end) forbidMerge[k] = true;  // merging would cross the barrier
      height[par] = std::max(height[par], height[k] + 1);
    }
    std::sort(unmergedHeightNode.begin() + begin, unmergedHeightNode.begin() + end);
  }
}

void EliminationTree::computeSparseElimRanges(const vector<int64_t>& noCrossPoints) {
  const int64_t n = ss.order();
  sparseElimRanges.push_back(0);

  const size_t numRanges = noCrossPoints.size() + 1;
  for (size_t r = 0; r < numRanges; r++) {
    int64_t begin = r == 0 ? 0 : noCrossPoints[r - 1];
    int64_t end = r < noCrossPoints.size() ? noCrossPoints[r] : n;

    int64_t k0 = begin;
    while (k0 < end) {
      // candidate set: nodes of one height class, in (size, id) order, while small enough
      const int64_t classHeight = std::get<0>(unmergedHeightNode[k0]);
      int64_t k1 = k0, numEasyMerge = 0;
      while (k1 < end && std::get<0>(unmergedHeightNode[k1]) == classHeight &&
             std::get<1>(unmergedHeightNode[k1]) <= kMaxSparseElimNodeSize) {
        int64_t p = parent[k1];
        if (p >= 0) {
          double fillAfterMerge = double(nodeRows[k1]) / double(nodeRows[p] + nodeSize[p]);
          if (fillAfterMerge > 0.8) numEasyMerge++;
        }
        k1++;
      }
      // give up when the set is small, or when most of it would merge cheaply anyway
      if (k1 - k0 < kMinNumSparseElimNodes || k1 - k0 < numEasyMerge * 3) break;

      for (int64_t k = k0; k < k1; k++) forbidMerge[std::get<2>(unmergedHeightNode[k])] = true;
      sparseElimRanges.push_back(k1);
      k0 = k1;
    }
    if (k0 < end) break;
  }
  if (sparseElimRanges.size() == 1) sparseElimRanges.clear();
}

void EliminationTree::computeMerges() {
  const int64_t n = ss.order();
  numMergedNodes.assign(n, 1);
  mergeWith.assign(n, -1);
  numMerges = 0;

  auto score = [&](int64_t k, int64_t p) {
    return double(nodeRows[k]) / double(nodeRows[p] + nodeSize[p]);
  };
  auto nodeTime = [&](int64_t node, double size, double rows, double merged) {
    return compMod.potrfEst(size) + compMod.trsmEst(size, rows) + sygeCosts[node].c0 +
           sygeCosts[node].c1 * size + asmblCosts[node].c0 + asmblCosts[node].c1 * merged;
  };

  // Extension for the level-scheduled backend (no counterpart in EliminationTree.cpp): a node
  // that is the ONLY child of its parent runs strictly before it, one level (potrf -> trsm ->
  // update, ~20 us of launches and dependent round trips) per 64-column panel, with nothing else
  // of that subtree to overlap; merging such a pair saves the levels the panel count drops by.
  // Siblings share their levels, so the term is not charged for them.
  const double kChainLevelCost = levelCost;
  constexpr double kPanel = 64.0;
  vector<int64_t> childCount(n, 0);
  for (int64_t k = 0; k < n; k++) {
    if (parent[k] >= 0) childCount[parent[k]]++;
  }
  auto levelsOf = [&](double size) { return std::ceil(size / kPanel); };
  // Round 6: the same credit for the child that is its parent's DEEPEST subtree (in panels = levels):
  // the parent's first panel comes one level after the last panel of that child, whatever the
  // siblings do, so a merge that drops a panel there drops a level of the whole factor.  Depths are
  // those of the un-merged tree, refreshed for a parent when a child is merged into it.
  vector<double> depth(n, 0.0), deepestChild(n, 0.0);  // depth[k]: panels from the leaves up to and incl. k
  for (int64_t k = 0; k < n; k++) {  // (parents have larger indices)
    depth[k] = deepestChild[k] + levelsOf(double(nodeSize[k]));
    if (parent[k] >= 0) deepestChild[parent[k]] = std::max(deepestChild[parent[k]], depth[k]);
  }

  using Cand = std::tuple<double, int64_t, int64_t>;  // (score, child, parent)
  std::priority_queue<Cand> queue;
  for (int64_t k = n - 1; k >= 0; k--) {
    if (forbidMerge[k] || parent[k] < 0) continue;
    queue.emplace(score(k, parent[k]), k, parent[k]);
  }

  vector<NodeStats> mergedStats;
  const bool denseMergeOn = denseMergeRule;
  const bool dbg = std::getenv("BSP_MERGE_DEBUG") != nullptr;  // developer aid
  long dbgCand = 0;
  while (!queue.empty()) {
    Cand top = queue.top();
    queue.pop();
    const int64_t k = std::get<1>(top);
    int64_t p = std::get<2>(top);

    // the recorded parent may have been merged upwards since: re-key on its current root
    const int64_t recorded = p;
    while (mergeWith[p] >= 0) p = mergeWith[p];
    if (p != recorded) {
      queue.emplace(score(k, p), k, p);
      continue;
    }

    const double sk = double(nodeSize[k]), rk = double(nodeRows[k]);
    const double sp = double(nodeSize[p]), rp = double(nodeRows[p]);
    const double tSeparate = nodeTime(k, sk, rk, double(numMergedNodes[k])) +
                             nodeTime(p, sp, rp, double(numMergedNodes[p]));
    double tMerged = nodeTime(p, sp + sk, rp, double(numMergedNodes[k] + numMergedNodes[p]));
    if (childCount[p] == 1 || depth[k] >= deepestChild[p]) {
      tMerged -= kChainLevelCost * (levelsOf(sk) + levelsOf(sp) - levelsOf(sk + sp));
    }
    // Extension: a child whose rows are (nearly) all of its parent's column -- the nodes of a dense
    // trailing block, e.g. the cameras of a Schur complement -- merges whatever the model says: the
    // merged lump adds < 10 % of explicit zeros and is one chain of panels with lookahead, the
    // split one pays a lump boundary (BAL-871 with a model of lower fixed costs: 8.5 against 7.7 ms)
    const bool denseMerge = denseMergeOn && score(k, p) >= 0.9;
    if (dbg) {
      dbgCand++;
      if (dbgCand <= 12) fprintf(stderr, "[merge] k %ld (size %g rows %g depth %g) p %ld (size %g rows %g deepest %g children %ld): separate %.3e merged %.3e\n", (long)k, sk, rk, depth[k], (long)p, sp, rp, deepestChild[p], (long)childCount[p], tSeparate, tMerged);
    }
    if (!(tMerged < tSeparate) && !denseMerge) continue;
    childCount[p] += childCount[k] - 1;
    deepestChild[p] = std::max(deepestChild[p] == depth[k] ? 0.0 : deepestChild[p], deepestChild[k]);
    depth[p] = deepestChild[p] + levelsOf(sk + sp);

    const int64_t oldSizeP = nodeSize[p], oldMergedP = numMergedNodes[p];
    mergeWith[k] = p;
    nodeSize[p] += nodeSize[k];
    numMergedNodes[p] += numMergedNodes[k];
    numMerges++;

    // merge the (column-sorted) row views of k and p; a column holding both rows now has one
    // taller block, and its modelled costs change accordingly
    const auto& viewK = perRowNodeStats[k];
    const auto& viewP = perRowNodeStats[p];
    mergedStats.clear();
    size_t ik = 0, ip = 0;
    while (ik < viewK.size() || ip < viewP.size()) {
      bool takeK = ip >= viewP.size() || (ik < viewK.size() && viewK[ik].colIdx < viewP[ip].colIdx);
      bool takeP = ik >= viewK.size() || (ip < viewP.size() && viewP[ip].colIdx < viewK[ik].colIdx);
      if (takeK) {
        if (viewK[ik].colIdx != k) mergedStats.push_back(viewK[ik]);
        ik++;
      } else if (takeP) {
        if (viewP[ip].colIdx != p) mergedStats.push_back(viewP[ip]);
        ip++;
      } else {
        const NodeStats& a = viewK[ik];
        const NodeStats& b = viewP[ip];
        const int64_t c = b.colIdx;
        sygeCosts[c] -= compMod.sygeLinEst(double(a.rowsDown + a.rows), double(a.rows));
        asmblCosts[c] -= compMod.asmblLinEst(double(a.rBlocksDown + a.rBlocks));
        sygeCosts[c] -= compMod.sygeLinEst(double(b.rowsDown + b.rows), double(b.rows));
        asmblCosts[c] -= compMod.asmblLinEst(double(b.rBlocksDown + b.rBlocks));
        const int64_t rows = a.rows + b.rows, blocks = a.rBlocks + b.rBlocks;
        sygeCosts[c] += compMod.sygeLinEst(double(b.rowsDown + rows), double(rows));
        asmblCosts[c] += compMod.asmblLinEst(double(b.rBlocksDown + blocks));
        mergedStats.push_back(NodeStats{c, blocks, rows, b.rBlocksDown, b.rowsDown});
        ik++;
        ip++;
      }
    }
    // the diagonal block of p grew
    sygeCosts[p] -= compMod.sygeLinEst(double(nodeRows[p] + oldSizeP), double(oldSizeP));
    asmblCosts[p] -= compMod.asmblLinEst(double(nodeRowBlocks[p] + oldMergedP));
    sygeCosts[p] += compMod.sygeLinEst(double(nodeRows[p] + nodeSize[p]), double(nodeSize[p]));
    asmblCosts[p] += compMod.asmblLinEst(double(nodeRowBlocks[p] + numMergedNodes[p]));
    mergedStats.push_back(
        NodeStats{p, numMergedNodes[p], nodeSize[p], nodeRowBlocks[p], nodeRows[p]});
    perRowNodeStats[p].swap(mergedStats);
  }
}

void EliminationTree::collapseMergePointers() {
  if (std::getenv("BSP_MERGE_DEBUG")) fprintf(stderr, "[merge] %ld merges of %ld nodes\n", (long)numMerges, (long)ss.order());
  // parents have larger indices, so a descending sweep sees final roots first
  for (int64_t k = ss.order() - 1; k >= 0; k--) {
    int64_t p = mergeWith[k];
    if (p >= 0 && mergeWith[p] >= 0) mergeWith[k] = mergeWith[p];
  }
}

void EliminationTree::processTree(bool detectSparseElimRanges,
                                  const vector<int64_t>& noCrossPoints, bool findOnlyElims) {
  const int64_t n = ss.order();

  computeNodeHeights(noCrossPoints);
  if (detectSparseElimRanges) computeSparseElimRanges(noCrossPoints);

  if (findOnlyElims) {
    mergeWith.assign(n, -1);
    numMergedNodes.assign(n, 1);
    numMerges = 0;
  } else {
    computeMerges();
    collapseMergePointers();
  }

  // lumps are numbered in (height,size,id) order of their root node
  const int64_t numLumps = n - numMerges;
  lumpStart.assign(numLumps + 1, 0);
  lumpToSpan.assign(numLumps + 1, 0);
  vector<int64_t> rootToLump(n, -1);
  int64_t lump = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t node = std::get<2>(unmergedHeightNode[i]);
    if (mergeWith[node] >= 0) continue;
    rootToLump[node] = lump;
    lumpStart[lump] = nodeSize[node];
    lumpToSpan[lump] = numMergedNodes[node];
    lump++;
  }
  BASPACHO_CHECK_EQ(lump, numLumps);
  cumSumVec(lumpStart);
  cumSumVec(lumpToSpan);

  // spans of a lump keep their relative (original) order
  permInverse.resize(n);
  vector<int64_t> cursor(lumpToSpan.begin(), lumpToSpan.end() - 1);
  for (int64_t i = 0; i < n; i++) {
    int64_t root = mergeWith[i] >= 0 ? mergeWith[i] : i;
    permInverse[i] = cursor[rootToLump[root]]++;
  }
}

void EliminationTree::computeAggregateStruct(bool fillOnlyForElims) {
  const int64_t n = ss.order();
  const int64_t numLumps = n - numMerges;

  SparseStructure filled = ss.symmetricPermutation(permInverse, /*lowerHalf=*/false,
                                                   /*sortIndices=*/false);
  if (fillOnlyForElims) {
    for (size_t e = 0; e + 1 < sparseElimRanges.size(); e++) {
      filled = filled.addIndependentEliminationFill(sparseElimRanges[e], sparseElimRanges[e + 1]);
    }
  } else {
    filled = filled.addFullEliminationFill();
  }
  SparseStructure cols = filled.transpose();  // csc of the lower half

  colStart.assign(1, 0);
  rowParam.clear();
  vector<int64_t> seenInLump(n, -1);
  for (int64_t a = 0; a < numLumps; a++) {
    for (int64_t q = cols.ptrs[lumpToSpan[a]]; q < cols.ptrs[lumpToSpan[a + 1]]; q++) {
      int64_t row = cols.inds[q];
      if (seenInLump[row] != a) {
        seenInLump[row] = a;
        rowParam.push_back(row);
      }
    }
    std::sort(rowParam.begin() + colStart.back(), rowParam.end());
    colStart.push_back((int64_t)rowParam.size());
  }
}

vector<int64_t> EliminationTree::computeSpanStart() {
  vector<int64_t> spanStart(paramSize.size() + 1, 0);
  leftPermute(spanStart.begin(), permInverse, paramSize);
  cumSumVec(spanStart);
  return spanStart;
}

}  // namespace BaSpaCho
