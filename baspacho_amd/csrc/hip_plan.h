// Device plan of the MI355X backend: everything the numeric kernels need, derived ONCE per
// (skeleton, lump range) on the host and uploaded as a handful of flat arrays.
//
// Design: the reference drives factor() as a host-serial loop of per-lump /
// per-board library calls (Solver.cpp:198-218).  Here every dense lump is cut into column
// PANELS of at most kPanelWidth columns; a panel is factored (potrf of its diagonal block,
// trsm of all rows below) and then pushes its rank-nb update  C = B*B^T  straight into its
// targets (right-looking), one SEGMENT per target: the rest of its own lump (linear
// addressing) and every board of the lump column (scatter through a per-(lump,board) chain
// offset table, replacing prepareAssemble's per-target table rebuild, MatOpsCuda.cu:471-481).
// Panels are grouped into LEVELS of mutually independent panels (elimination-tree
// parallelism the reference never exploits); one level = three launches (potrf, trsm, update)
// whose grids cover all panels of the level.
#pragma once

#include <cstdint>
#include <vector>

#include "skeleton.h"

namespace BaSpaCho {

constexpr int kPanelWidth = 64;   // max panel width nb (potrf/trsm granularity)
constexpr int kOuterWidth = 256;  // outer block: panels update only their own outer block right
                                  // away; everything to its right gets ONE rank-256 update
constexpr int kTile = 64;         // update tile (rows x cols) and trsm row tile
constexpr int kElimSmallMax = 16; // widest lump handled by the small sparse-elim kernels

struct PanelDesc {
  int64_t diagOff;     // data offset of the nb x nb diagonal block of the panel
  int32_t lda;         // row stride (= lump width)
  int32_t nb;          // panel width
  int32_t rowsBelow;   // rows below the diagonal block: nRest + chain rows of the lump
  int32_t nRest;       // rows that still belong to the lump's own diagonal block
  int32_t lumpRowBase; // index of the lump's first chain row in the rowChain/rowLocal/rowColOff arrays
  int32_t lump;
  int32_t vecOff;      // row index (in the full matrix) of the panel's first column
  int32_t pad;
};

// source of a rank-K update: K consecutive columns of a lump and all the rows below them
struct SrcDesc {
  int64_t off;         // data offset of (first row below the source columns, first source column)
  int32_t lda;         // row stride (= lump width)
  int32_t K;           // number of source columns (<= kOuterWidth)
  int32_t rowsBelow;   // rows below the source columns: nRest + chain rows of the lump
  int32_t nRest;       // of which still inside the lump's own diagonal block
  int32_t lumpRowBase; // as in PanelDesc
  int32_t pad;
};

enum SegKind : int32_t { kSegIntra = 0, kSegBoard = 1 };

struct SegDesc {
  int32_t src;
  int32_t kind;
  int32_t q0;            // first below-row index covered by the segment's columns
  int32_t m;             // number of columns
  int64_t tgtBase;       // intra: data offset of element (row q=0, col q=0) of the target region
  int32_t tgtStride;     // row stride of the target lump
  int32_t firstChainOrd; // board: below-diagonal chain ordinal of the segment's first chain
  int64_t chainTabPtr;   // board: index into chainOffTab of that chain's entry
  int32_t outer;         // 1: source is a complete outer block (rank-kOuterWidth update)
  int32_t lump;          // lump owning the source columns
  int32_t rowMin;        // rows (below-row index) smaller than this are left untouched
  int32_t pad;           // bit 0: several writers in one launch, bit 1: writers of two side streams may meet
};

struct UpdTask {
  int32_t seg;
  int32_t rowTile;  // first below-row index of the tile rows
  int32_t colTile;  // first below-row index of the tile cols
  int32_t atomic;   // bit 0: several writers in one launch (panels of a level, units of a launch);
                    // bit 1: writers of two side streams may meet (kernels mask it off otherwise)
};

// Self-contained form of an UpdTask of an intra-lump segment whose source width is a multiple of
// the K chunk (every lookahead unit is): what updateTileBulk needs in ONE uniform 64-byte load
// instead of the task -> segment -> source chain of three dependent ones.  Built at upload time.
struct UpdTaskFat {
  int64_t srcOff;   // SrcDesc::off
  int64_t tgtBase;  // SegDesc::tgtBase
  int32_t lda, K, rowsBelow, segEnd;
  int32_t rowTile, colTile, tgtStride, rowMin;
  int32_t atomic, fast, pad0, pad1;  // fast = 0: not eligible (board segment / ragged K)
};
static_assert(sizeof(UpdTaskFat) == 64, "one scalar load");

// Self-contained form of ANY UpdTask (board segments, ragged K): the fields of the task, its
// segment and its source that updateTile reads, in one uniform 96-byte load -- the task -> segment
// -> source chain was three dependent round trips at the head of every tile of a small front.
// Built at upload time.
struct UpdTaskWide {
  int64_t srcOff;       // SrcDesc::off
  int64_t tgtBase;      // SegDesc::tgtBase
  int64_t chainTabPtr;  // SegDesc::chainTabPtr
  int32_t lda, K, rowsBelow, nRest;                 // SrcDesc
  int32_t lumpRowBase, kind, segEnd, tgtStride;     // SrcDesc / SegDesc (segEnd = q0 + m)
  int32_t firstChainOrd, rowMin, rowTile, colTile;  // SegDesc / UpdTask
  int32_t atomic, pad0, pad1, pad2, pad3, pad4;
};
static_assert(sizeof(UpdTaskWide) == 96, "two scalar loads");

struct TrsmTask {
  int32_t panel;
  int32_t rowTile;
};

// Self-contained form of a TrsmTask (the panel fields the kernel reads beside the row tile): one
// uniform load instead of task -> panel.  Built at upload time.
struct TrsmTaskFat {
  int64_t diagOff;
  int32_t lda, nb, rowsBelow, rowTile;
  int32_t pad0, pad1;
};
static_assert(sizeof(TrsmTaskFat) == 32, "one scalar load");

struct LevelRange {
  int64_t panelBegin, panelEnd;  // into levelPanels
  int64_t trsmBegin, trsmEnd;    // into trsmTasks
  int64_t updBegin, updEnd;      // into updTasks: tiles that must run before the next level
  // LOOKAHEAD.  Tiles of a block-wide (rank-256) intra-lump update whose target columns lie
  // beyond the NEXT outer block are not needed by the next 4 panels: they form the deferred list
  // and may run on a second stream concurrently with the following levels.  They must be complete
  // before the update launch of level `waitDefLevel`-consumers (see below).
  int64_t defBegin, defEnd;      // into updTasks (deferred tiles of this level)
  // [defBegin, defMid): the deferred tiles in the columns of the outer block AFTER the next one,
  // i.e. the only ones the next block's own update launch must wait for; they run first
  int64_t defMid = 0;
  // DIRECT chain kernels (hip_kernels.h): set when the level holds one panel; directSeg >= 0 when
  // its non-deferred tiles [updBegin, updEnd) are exactly the tiles of that one intra segment
  int32_t directPanel = -1, directSeg = -1;
  // fuseNext: tile 0 of directSeg is the diagonal block of the NEXT level's (single) panel, whose
  // potrf is fused into this level's update launch (updateTileDirectPotrf); the next level then
  // starts at its trsm
  int32_t fuseNext = 0;
  // splitK > 0 (block-wide segment, fuseNext): the update of tile 0 by the block's first splitK
  // source columns (final since the previous levels) is done by one extra workgroup of THIS
  // level's trsm launch, so that the potrf workgroup of the update launch only applies the
  // columns of the last panel
  int32_t splitK = 0;
  // rawNext: the column block of the NEXT level's (single) panel below its diagonal block is
  // written by this level's direct update launch alone, which then also stores it to the chain's
  // staging buffer (chainStep reads the unsolved panel rows from there)
  int32_t rawNext = 0;
  int64_t waitDefLevel;          // index (within the same level list) of the level whose deferred
                                 // tiles must be complete before this level's update launch; -1
  // intra-block step of a chain whose outer block is followed by another one: the step may also
  // apply its panel's rank-nb update to the NEXT block's tile (0,0) (chainStep, extraDiag), so that
  // the block-last step's potrf workgroup does not have to apply the whole block from memory
  int32_t extraDiag = 0;
  // due-stream mode: level whose OPTIONAL lookahead units (forked two outer blocks earlier: plain
  // read-modify-write on far columns) must be complete before this level's due units start; -1
  int64_t optWaitLevel = -1;
  // gather overlap (ElimRangePlan::chunkItemPtr): chunk that must be complete before this level's
  // own launches / its due lookahead units / its optional ones; -1: none
  int32_t gatherNow = -1, gatherDue = -1, gatherOpt = -1;
  // persistent tail: 1 = this level's panel is the first of a tail (the launch factors tailPanels
  // panels from here, after a join with the lookahead streams), 2 = a panel inside a tail (nothing to launch)
  int32_t tail = 0, tailPanels = 0;
  // the level before a tail hands over every pending lookahead unit as due units: their launch must
  // also wait for the optional units of the PREVIOUS block boundary (plain read-modify-write on the
  // same columns)
  int32_t flushDue = 0;
};

// One work item of the gather-form sparse-elimination update: a target block (sj,si) of the
// factor and the list of source chain pairs (one per eliminated column holding both sj and si)
// whose products B_j * B_i^T are summed into it.  Pairs are referenced by 32-bit element offsets.
struct ElimGatherItem {
  int64_t tgtOff;     // data offset of the target block
  int32_t pairBegin;  // into elimPairOffJ / elimPairOffI
  int32_t pairEnd;
  int32_t tgtStride;  // row stride of the target lump
  int16_t rows, cols; // |sj|, |si|
  int16_t n;          // width of the source lumps of this item
  int16_t flags;      // bit0: target shared with other items (atomic), bit1: diagonal block
  uint32_t firstJ, firstI;  // offsets of the first pair (= elimPairOffJ/I[pairBegin]): most tiny
                            // items hold one or two pairs, and reading the list costs a dependent
                            // round trip (K2t)
};
constexpr int kGatherMaxElems = 256;   // rows*cols handled per wave (4 per lane)
// pairs per work item: longer lists are split (the pieces subtract from the target with atomics).
// One wave walks an item 8 pairs per memory round trip, so a 2048-pair item (the diagonal targets
// of a bundle-adjustment problem: every point a camera sees; 17 % of all pairs sit in 0.2 % of the
// targets) took 256 round trips ~ 0.5 ms and whatever was left of them when the rest of the grid had
// drained WAS the kernel's tail: BAL-871 1.73 ms at 2048, 1.49 at 512, 1.43 at 128, 1.42 at 64
// (BSP_GATHER_MAX_PAIRS overrides; 2048+ restores the atomic-free, deterministic form).
constexpr int kGatherMaxPairs = 64;  // default of HipPlanOptions::gatherMaxPairs (round 4: 128 -> 64, -0.05 ms on BAL-871)
constexpr uint32_t kGatherChunkElems = 0xffffffffu;  // optional slicing of the source columns into
                                                     // cache-sized passes: measured 2x SLOWER on
                                                     // BAL-871 (more items + atomics), so disabled

// One eliminated lump as the small-lump factor kernel wants it: everything behind one 16-byte load
// instead of a chain of dependent skeleton lookups (the column is one dense (n + rows) x n block:
// rows start right after the n x n diagonal block).
struct ElimLumpDesc {
  int64_t diagOff;
  int32_t n;
  int32_t rowsBelow;
};

// one sparse-elimination range restricted to the planned lump range
struct ElimRangePlan {
  int64_t lumpBegin, lumpEnd;
  int64_t chainBegin, chainEnd;  // absolute chain indices covered by the range
  int64_t chainLumpOff;          // offset into elimChainLump of chain `chainBegin`
  int32_t maxWidth;              // widest lump of the range
  int64_t descBegin = 0;         // offset into elimLumpDesc of lump `lumpBegin`
  std::vector<LevelRange> bigLevels;  // lumps wider than kElimSmallMax go through panels
  bool useGather = false;             // pair updates in gather form (atomic-free) ...
  int64_t itemBegin = 0, itemEnd = 0; // ... over these ElimGatherItems (one wave per item)
  int64_t tinyBegin = 0, tinyEnd = 0; // items with <= 16 target elements: 4 items per wave
  int64_t tiny9End = 0;               // [tinyBegin, tiny9End): target and both source blocks <= 9
                                      // elements (3x3 parameters): 7 items per wave
  int64_t ldsBegin = 0, ldsEnd = 0;   // items wider or taller than 16: LDS-staged kernel (K2g);
                                      // [itemBegin, itemEnd) go to the MFMA kernel (K2m)
  // OVERLAP WITH THE DENSE PHASE (round 6).  When every target of the range's gather lies in ONE wide
  // lump that is the whole dense part of the plan (a Schur complement whose cameras merged into one
  // lump), the MFMA items are grouped into CHUNKS of target column blocks (kOuterWidth columns each),
  // first columns first: chunk 0 runs on the execution stream, the others on a stream of their own
  // BESIDE the dense chain, and every dense launch waits (event) for the chunk of the last column
  // block it touches -- LevelRange::gatherNow / gatherDue / gatherOpt.  No atomics: a target block is
  // never worked on from both sides.
  std::vector<int64_t> chunkItemPtr;     // [chunks + 1], into elimItems; empty: one launch
  std::vector<int32_t> chunkOfColBlock;  // chunk of every outer column block of the target lump
  int64_t overlapLump = -1;
};

// Every switch the plan builder honours, resolved ONCE when a SymbolicCtx is created (from
// HipBackendOptions, backend_options.h) and recorded in the plan it produced: the launch code takes the
// schedule-shaping ones (dueStream, dueSplit) from the plan it runs, never from a second read of
// the environment, so the builder's "two streams may meet" bits and the launcher's choice of
// streams cannot disagree.
struct HipPlanOptions {
  bool dueStream = true;      // BSP_DUE_STREAM: due lookahead units on a stream of their own
  bool planTiming = false;    // BSP_TIMING: stderr laps of the gather plan
  int32_t gatherMaxPairs = kGatherMaxPairs;  // BSP_GATHER_MAX_PAIRS
  double bulkAhead = 0.8;     // BSP_BULK_AHEAD (round 4: 0.6 -> 0.8, -0.04 ms on BAL-871; profiles/r04_ab_plan_knobs.txt)
  bool gatherOverlap = false; // BSP_GATHER_OVERLAP=1: gather chunks beside the dense chain (measured: no gain, profiles/r06_ab_gather_overlap.txt)
  int32_t overlapFirst = 2;   // column blocks in chunk 0 (BSP_GATHER_OVERLAP_FIRST)
  int32_t overlapStep = 3;    // ... in every later chunk (BSP_GATHER_OVERLAP_STEP)
  int32_t overlapMinBlocks = 8;  // narrowest target lump, in column blocks
  // PERSISTENT TAIL (hip_tail_kernel.h): the last tailBlocks outer blocks of a lump that has nothing
  // below it and is at least tailMinBlocks blocks wide are factored by ONE flag-synchronised launch
  // (BSP_TAIL_BLOCKS; 0: the level schedule to the end)
  int32_t tailBlocks = 6;
  int32_t tailMinBlocks = 6;
  // NARROW root lumps (tailNarrowMin <= blocks < tailMinBlocks; GRID 82x82: 990 columns = 4 blocks):
  // everything after the first outer block is the tail.  Pays for ONE matrix (GRID 82x82 1.144 -> 1.089
  // ms), loses for batches (batch of 8: 2.23 -> 2.35 ms: 8 x 78 resident roles) -- like EVERY tail: factor()
  // of a batch runs a second plan without tails (HipSymbolicCtx::planFor, tag 2).  0: off.
  // profiles/r06_tail_narrow.txt.
  // A tail needs at least six panels, and -- like every tail -- panels that are alone in their levels.
  bool tailWholeNarrow = true;  // a narrow root lump that follows other levels: the whole lump (developer: BSP_TAIL_WHOLE=0)
  int32_t tailNarrowMin = 2;  // (developer: BSP_TAIL_NARROW_MIN; tailBlocks = 0 switches every tail off)
  int32_t solveSortWindow = 16;  // lumps per sorting window of the backward elimination lists (= a workgroup; developer: BSP_SOLVE_SORT_WINDOW)
  void applyDeveloperEnv();  // BSP_TIMING, chunk sizes of the gather-overlap experiment
};

struct HipPlanHost {
  HipPlanOptions opts;
  bool hasTail = false;  // some lump hands its last columns to the persistent tail launch
  int64_t startLump = 0, upToLump = 0;
  std::vector<ElimRangePlan> elimRanges;
  std::vector<int32_t> elimChainLump;  // lump of every chain inside elimination ranges
  std::vector<ElimLumpDesc> elimLumpDesc;  // every lump of every elimination range
  std::vector<ElimGatherItem> elimItems;
  std::vector<uint32_t> elimPairOffJ, elimPairOffI;

  std::vector<PanelDesc> panels;
  std::vector<SrcDesc> srcs;
  std::vector<SegDesc> segs;
  std::vector<int64_t> chainOffTab;
  std::vector<int32_t> rowChain, rowLocal, rowColOff;  // per chain row of every dense lump
  std::vector<int32_t> rowGlobal;                      // ... and its row index in the full matrix

  double tailUpdFlops = 0;  // update flops done inside persistent tail launches (not in updFlops)
  double updFlopsDirect = 0, elimPairOperandElems = 0, elimTargetElems = 0, trsmFlops = 0,
         potrfFlops = 0;  // part of updFlops launched through the direct chain kernels
  // of trsmFlops / potrfFlops: the part done inside chainStep launches / inside the previous
  // level's update launch (same decisions as launchLevels with the default switches)
  double trsmFlopsMerged = 0, potrfFlopsFused = 0;
  std::vector<int32_t> levelPanels;
  std::vector<TrsmTask> trsmTasks;
  std::vector<UpdTask> updTasks;
  std::vector<LevelRange> levels;

  // statistics
  double flops = 0;          // algorithmic flops of this plan (n^3/3 + r n^2 + r^2 n per lump)
  double updElems = 0;       // lower-trapezoid elements written by update tiles
  double updFlops = 0;       // 2 * nb * (lower-trapezoid elements), summed over segments
  double elimPairElems = 0;  // target elements touched by sparse-elimination pair updates
  double elimPairFlops = 0;  // 2 * n * (pair elements)
  double elimColElems = 0;   // numeric elements of the sparse-eliminated columns
  int64_t numLaunches = 0;
  int64_t maxPanelsInLevel = 0;
  int64_t maxChainRows = 0;  // max rows below a panel whose level sets rawNext (staging buffer rows)
  bool hasDeferred = false;  // some level carries lookahead (deferred) tiles
  double deferredFlops = 0;  // flops of the lookahead units (what the side streams run)
  int64_t numForkLevels = 0; // levels that fork lookahead units (one event round trip each)
  // The lookahead schedule costs a cross-queue event round trip per fork and per wait (~40-50 us of
  // idle execution stream each time the chain has caught up, rocprofv3 on GRID 82x82: 1.73 ms with
  // the side streams, 1.46 with the same launches in line); it pays when the units it moves off the
  // execution stream are worth more than that.  Per matrix; a batch multiplies the work per fork.
  // (measured, on / off: GRID 82x82 0.04 GF per fork 1.73 / 1.46 ms, small BAL 0.04 GF 0.80 / 0.78,
  //  64 x GRID 2.3 GF 12.60 / 12.50, BAL-871 4.8 GF 7.17 / 8.78, FLAT-50k 17 GF 28.6 / 31.9)
  static constexpr double kMinDeferredFlopsPerFork = 3e9;
  bool lookaheadPays(int batch = 1, double minFlopsPerFork = kMinDeferredFlopsPerFork) const {
    return numForkLevels > 0 && deferredFlops * batch >= minFlopsPerFork * double(numForkLevels);
  }
};

// Build the plan for factoring lumps [startLump, upToLump) (sparse-elimination ranges fully
// inside the interval are included); `sparseElimRanges` as stored by the Solver.
HipPlanHost buildHipPlan(const CoalescedBlockMatrixSkel& skel,
                         const std::vector<int64_t>& sparseElimRanges, int64_t startLump,
                         int64_t upToLump, const HipPlanOptions& opts = HipPlanOptions());

// Plan of ONE dense operation of the per-op boundary (NumericCtx::potrf / trsm, MatOps.h:124-127)
// on a row-major n x n block at offA followed (contiguously, as in Solver::factorLump,
// Solver.cpp:42-64) by k rows: potrfOnly -> Cholesky of the block, rows below untouched;
// otherwise -> the k rows are solved against the already factored block.
// vecOff: position of the block's first column in a right-hand side vector (the per-op
// SolveCtx::solveL / solveLt reuse the potrfOnly plan: its panels and row tiles).
// lda: row stride of the block and of the rows below it (0: the block is contiguous, lda = n);
// a span inside a wider lump has lda = lump width (NumericCtx::pseudoFactorSpans).
HipPlanHost buildDenseOpPlan(int64_t n, int64_t k, int64_t offA, bool potrfOnly,
                             int64_t vecOff = 0, int64_t lda = 0);

// Forward solve over a sparse-elimination range in gather form: the below-diagonal blocks of
// the small lumps listed per TARGET row span, so that a workgroup sums the products of up to
// 256 blocks of one span and issues one atomic per row (hip_solve_kernels.h, K-S2).
struct SolveGatherEntry {
  int64_t dataOff;  // data offset of the block (rows x n, row-major)
  int32_t xOff;     // position of the source lump in the vector
  int32_t n;        // width of the source lump
};
struct SolveGatherItem {
  int32_t entryBegin, entryEnd;  // <= 256 entries
  int32_t rowStart, rows;        // target span in the vector
  int32_t maxN, pad;             // widest source lump among the entries
};
// Backward solve over a range of <= 4-wide eliminated lumps (K-S3): the same blocks in their
// natural (lump-major) order, with the position of their rows in the vector, so that the kernel
// needs no skeleton lookups
struct SolveLumpBlock {
  int64_t dataOff;  // data offset of the block (rows x n, row-major)
  int32_t yOff;     // first row of the block's span in the vector
  int32_t rows;
};
struct SolveLumpDesc {
  int64_t diagOff;              // data offset of the n x n diagonal block
  int32_t blockBegin, blockEnd; // into the SolveLumpBlock list
  int32_t xOff, n;              // position in the vector, width
};
struct SolveGatherPlan {
  std::vector<SolveGatherEntry> entries;
  std::vector<SolveGatherItem> items;
  std::vector<std::pair<int64_t, int64_t>> rangeItems;  // item range of every elimination range
  std::vector<SolveLumpBlock> lumpBlocks;
  std::vector<SolveLumpDesc> lumpDescs;                  // one per lump of every range, in order
  std::vector<int64_t> rangeLumpDesc;                    // first SolveLumpDesc of every range
  // per range: the common width of its lumps (0: mixed, or wider than 4) and the first span any of their
  // blocks reaches (= from where K-S3t of hip_solve_wide.h copies; the number of spans if there is none)
  struct RangeWide {
    int32_t n;
    int64_t spanBelow, rowBelow;
  };
  std::vector<RangeWide> rangeWide;
};
SolveGatherPlan buildSolveGather(const CoalescedBlockMatrixSkel& skel, const HipPlanHost& plan);

}  // namespace BaSpaCho
