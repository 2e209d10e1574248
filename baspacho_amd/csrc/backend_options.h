// Schedule switches of the MI355X backend in ONE struct (round 6): a caller sets them through
// Settings::hipOptions / bsp_hip_options (include/baspacho_amd.h), the library resolves defaults, and
// the environment -- read HERE and nowhere else, once per Solver -- only overrides them for A/B
// scripts.  No counterpart in the reference (its Settings has the four fields of Solver.h:212-218);
// every field: a negative value (NaN for the doubles) = the library's default.
#pragma once

#include <cmath>
#include <cstdint>

namespace BaSpaCho {

struct HipBackendOptions {
  // launch schedule of factor()
  int32_t lookahead = -1;         // 0: lookahead units in line (BSP_NO_LOOKAHEAD=1)
  int32_t dueStream = -1;         // 0: one auxiliary stream (BSP_DUE_STREAM)
  int32_t splitK = -1;            // 0: no split-K tile lists (BSP_SPLIT_K)
  int32_t gatherMaxPairs = -1;    // pairs per gather item (BSP_GATHER_MAX_PAIRS)
  int32_t gatherOverlap = -1;     // 1: gather chunks beside the dense chain (BSP_GATHER_OVERLAP)
  int32_t subBatchMin = -1;       // batches of at least this many as concurrent sub-batches; 0: never (BSP_SUB_BATCH_MIN)
  int32_t subBatches = -1;        // ... this many parts (BSP_SUB_BATCHES)
  int32_t tailBlocks = -1;        // outer blocks of a wide root lump factored by the persistent tail; 0: off (BSP_TAIL_BLOCKS)
  int32_t lazyPlan = -1;          // 1: no eager device plan at construction (BSP_LAZY_PLAN)
  // solve()
  int32_t blockSolve = -1;        // 0: wide lumps by panel (BSP_BLOCK_SOLVE)
  int32_t solveInv = -1;          // 0: substitution instead of inverted diagonal blocks (BSP_SOLVE_INV)
  int32_t solveSweep = -1;        // 0: no persistent sweeps (BSP_SOLVE_SWEEP)
  int32_t sweepMinWidth = -1;     // narrowest run a sweep takes (BSP_SWEEP_MIN_WIDTH)
  int32_t solveWide = -1;         // 0: no right-hand-sides-across-the-lanes backward elimination pass (BSP_SOLVE_WIDE)
  // symbolic analysis
  int32_t chainContraction = -1;  // 0: no contraction of pivot chains before the ordering (BSP_CHAIN_CONTRACTION)
  int32_t denseMerge = -1;        // 0: no "rows >= 90 % of the parent" merge rule (BSP_DENSE_MERGE_OFF=1)
  int32_t expectedBatch = -1;     // matrices per factor() call the merge model plans for (BSP_EXPECTED_BATCH); default 1
  double lookaheadMinGF = NAN;    // GF per fork below which lookahead units stay in line (BSP_LOOKAHEAD_MIN_GF)
  double bulkAhead = NAN;         // share of the next block's chain handed out as optional units (BSP_BULK_AHEAD)
  double levelCostUs = NAN;       // supernode merges: what one level on the critical path costs (BSP_LEVEL_COST_US); default 28

  // the environment on top (A/B scripts); called once per Solver
  void applyEnv();
  static bool on(int32_t v, bool dflt) { return v < 0 ? dflt : v != 0; }
};

}  // namespace BaSpaCho
