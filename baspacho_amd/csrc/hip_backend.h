// Extras of the MI355X backend that sit beside the reference boundary (mat_ops.h):
// per-kernel-class event timing and plan statistics, used by bench.py / the C ABI.
#pragma once

#include <cstdint>
#include <vector>

#include "mat_ops.h"

namespace BaSpaCho {

enum HipProfKind {
  kProfElimFactor = 0,
  kProfElimUpdate = 1,
  kProfPotrf = 2,
  kProfTrsm = 3,
  kProfUpdate = 4,       // updateTile launches (task lists; the bulk of the flops)
  kProfChainUpdate = 5,  // updateTileDirect / updateTileDirectPotrf launches of one-panel levels
  kProfNumKinds = 6
};

struct HipKernelProfile {
  double ms[kProfNumKinds] = {0, 0, 0, 0, 0, 0};
  int64_t launches[kProfNumKinds] = {0, 0, 0, 0, 0, 0};
  // time during which AT LEAST ONE launch of the class was running (launches of one class overlap
  // when they go to different streams: their durations then add up to more than the time the class
  // occupies the device)
  double busyMs[kProfNumKinds] = {0, 0, 0, 0, 0, 0};
};

struct HipPlanStats {
  double flops = 0, updElems = 0, updFlops = 0, elimPairElems = 0, elimPairFlops = 0,
         elimColElems = 0, updFlopsDirect = 0, elimPairOperandElems = 0, elimTargetElems = 0,
         trsmFlops = 0, potrfFlops = 0, trsmFlopsMerged = 0, potrfFlopsFused = 0;
  int64_t numLaunches = 0, numLevels = 0, numPanels = 0, numSegs = 0, numUpdTasks = 0,
          numTrsmTasks = 0, chainTabEntries = 0, maxPanelsInLevel = 0, numAtomicUpdTasks = 0,
          numGatherGroups = 0,  // (always 0 since round 4: the overlapped elimination was removed)
          numForkLevels = 0;
  double deferredFlops = 0;
  double tailUpdFlops = 0;  // update flops done inside persistent tail launches (not part of updFlops)
  int64_t numTailPanels = 0;
};

// while `prof` is non-null every kernel launch of the context is bracketed by HIP events and
// accumulated per kernel class (factor calls then synchronise).  inSitu = false: the launches run
// one after the other on the execution stream (lookahead off: isolated kernel times); inSitu =
// true: the real schedule, side-stream launches timed on the side stream (what a kernel takes
// beside the others -- the number a rocprofv3 kernel trace of the timed steps shows)
void hipBackendSetProfile(SymbolicCtx& sym, HipKernelProfile* prof, bool inSitu = false);

// sustained fp64 MFMA rate of this GPU measured with a register-only probe kernel (TFLOP/s)
double hipBackendMfmaF64ProbeTflops();
// in-situ kernel trace (BSP_KTRACE builds only; returns 0 records otherwise)
int hipBackendReadTrace(long long* out, int maxRecords);
// BSP_KTRACE + BSP_TRACE_TILE builds: per chain-step launch {first start, last start, last end,
// workgroup-0 end} over all its workgroups (wall clock, 100 MHz); resets the table
int hipBackendReadExtents(unsigned long long* out, int maxLaunches);

// TESTING: make factor() take the reference-style per-op loop (potrf/trsm/saveSyrkGemm/
// prepareAssemble/assemble/doElimination virtuals) instead of the fused path
void hipBackendForcePerOp(SymbolicCtx& sym, bool on);

// TESTING, fault injection for the full-size parity tests: kind 1 = factor() does not launch the
// sparse-elimination update (the factor is then wrong and the checks must say so), 0 = off.
// Never read from the environment: a leaked variable cannot corrupt a caller's factor.
void hipBackendSetFault(SymbolicCtx& sym, int kind);

// what the calls on this Solver actually ran (cumulative): lets a test see that the path it means
// to test was taken, and a caller poll for a watchdog report without waiting for the next solve
struct HipRunCounters {
  int64_t sweepLaunches = 0;       // persistent solve sweeps launched (hip_sweep_kernels.h)
  int64_t sweepTimeouts = 0;       // ... that ran into their watchdog (reported by a later call)
  int64_t splitListsUsed = 0;      // update launches that took a split-K tile list
  int64_t subBatchesEnqueued = 0;  // sub-batches enqueued on their own stream
  int64_t lookaheadForks = 0;      // lookahead launches handed to the auxiliary streams
  int64_t gatherChunksOverlapped = 0;  // sparse-elimination gather chunks launched beside the dense chain
  int64_t tailLaunches = 0;        // persistent tail launches (hip_tail_kernel.h)
  int64_t solveWideLaunches = 0;   // backward elimination passes of hip_solve_wide.h
  int64_t invReused = 0;           // backward passes that reused the inverses of their forward pass
  int64_t potrfFoldedLevels = 0;   // tree levels whose potrf launch was folded into their trsm launch
  int64_t sweepMfmaLaunches = 0;   // ... of sweepLaunches: the matrix-core form for several right-hand sides
  int64_t sweepsRetired = 0;       // 1: a time-out retired the sweeps of this Solver
  int64_t sweepErrorPending = 0;   // 1: a time-out has been raised and not been reported yet
};
HipRunCounters hipBackendRunCounters(SymbolicCtx& sym);
// developer aid (BSP_SWEEP_TRACE=1 when the Solver is created): clock stamps {start, operands on chip,
// inputs arrived, x published} (100 MHz) of every spine workgroup of the LAST persistent sweep
int hipBackendReadSweepTrace(SymbolicCtx& sym, long long* out, int maxBlocks);

// per level of the plan (host only): {elimination range or -1, panels, widest panel, most rows below a
// panel, trsm tasks, update tiles, lookahead tiles, rows below summed over the panels}
std::vector<int64_t> hipBackendPlanLevels(SymbolicCtx& sym, int64_t startLump, int64_t upToLump);

HipPlanStats hipBackendPlanStats(SymbolicCtx& sym, int64_t startLump, int64_t upToLump);

}  // namespace BaSpaCho
