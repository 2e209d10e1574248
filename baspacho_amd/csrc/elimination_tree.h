// Elimination tree of a (permuted) block pattern + the two heuristics layered on it:
// detection of "sparse elimination" ranges (large sets of small independent leaves) and
// cost-model driven merging of children into parents (supernodes == lumps).
// Host-only.  Behaviour follows baspacho/baspacho/EliminationTree.{h,cpp}
// (buildTree :28-95, heights :106-131, elim ranges :133-180, merges :182-293,
//  processTree :311-366, aggregate structure :368-402).
#pragma once

#include <cstdint>
#include <tuple>
#include <vector>

#include "computation_model.h"
#include "sparse_structure.h"

namespace BaSpaCho {

struct EliminationTree {
  EliminationTree(const std::vector<int64_t>& paramSize, const SparseStructure& ss,
                  const ComputationModel* compMod = nullptr);

  void buildTree();

  void processTree(bool detectSparseElimRanges, const std::vector<int64_t>& noCrossPoints = {},
                   bool findOnlyElims = false);

  void computeAggregateStruct(bool fillOnlyForElims = false);

  std::vector<int64_t> computeSpanStart();

  // steps of processTree
  void computeNodeHeights(const std::vector<int64_t>& noCrossPoints);
  void computeSparseElimRanges(const std::vector<int64_t>& noCrossPoints);
  void computeMerges();
  void collapseMergePointers();

  // inputs
  bool denseMergeRule = true;  // HipBackendOptions::denseMerge ("rows >= 90 % of the parent" merges)
  int expectedBatch = 1;       // matrices per factor() call (createSolver scales the model's throughput terms by it)
  double levelCost = 2.8e-5;   // seconds one level on the critical path costs (HipBackendOptions::levelCostUs)
  std::vector<int64_t> paramSize;
  const SparseStructure& ss;  // csr, lower half, already fill-reducing ordered
  const ComputationModel& compMod;

  // buildTree
  std::vector<int64_t> parent;
  std::vector<int64_t> nodeSize;
  std::vector<int64_t> nodeRows;
  std::vector<int64_t> nodeRowBlocks;
  std::vector<std::vector<int64_t>> perColNodes;
  struct NodeStats {
    int64_t colIdx, rBlocks, rows, rBlocksDown, rowsDown;
  };
  std::vector<std::vector<NodeStats>> perRowNodeStats;
  std::vector<LinCost> sygeCosts;   // per column, linear in the column's node size
  std::vector<LinCost> asmblCosts;  // per column, linear in the number of merged nodes

  // processTree
  std::vector<int64_t> sparseElimRanges;
  std::vector<std::tuple<int64_t, int64_t, int64_t>> unmergedHeightNode;  // (height,size,node)
  std::vector<bool> forbidMerge;
  std::vector<int64_t> numMergedNodes;
  std::vector<int64_t> mergeWith;
  int64_t numMerges = 0;

  // outputs
  std::vector<int64_t> permInverse;
  std::vector<int64_t> lumpStart;
  std::vector<int64_t> lumpToSpan;
  std::vector<int64_t> colStart;
  std::vector<int64_t> rowParam;
};

}  // namespace BaSpaCho
