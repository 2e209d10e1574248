// MI355X backend: Ops / SymbolicCtx / NumericCtx / SolveCtx implemented with the hand-written
// HIP kernels of hip_kernels.h and the level-scheduled device plan of hip_plan.h.
// Takes the place of baspacho/baspacho/MatOpsCuda.cu (CudaSymbolicCtx :56-143, CudaNumericCtx
// :408-604, batched :605-725) without cuBLAS/cuSOLVER, per-op syncs, per-call allocations or
// per-lump host->device table uploads.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "backend_options.h"
#include "hip_backend.h"
#include "hip_kernels.h"
#include "hip_solve_kernels.h"
#include "hip_sweep_kernels.h"
#include "hip_solve_wide.h"
#include "hip_sweep_mfma.h"
#include "hip_tail_kernel.h"
#include "mat_ops.h"

namespace BaSpaCho {

using std::vector;

#define hipCHECK(expr)                                                                    \
  do {                                                                                    \
    hipError_t bsp_err_ = (expr);                                                         \
    if (bsp_err_ != hipSuccess) {                                                         \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(bsp_err_) + \
                               " in " #expr " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    }                                                                                     \
  } while (0)

namespace {

// owning device array
struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : ptr(o.ptr), bytes(o.bytes) { o.ptr = nullptr; o.bytes = 0; }
  ~DevBuf() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  void resize(size_t n) {
    if (n <= bytes) return;
    release();
    hipCHECK(hipMalloc(&ptr, n));
    bytes = n;
  }
  template <typename U>
  void upload(const vector<U>& v) {
    resize(std::max<size_t>(v.size() * sizeof(U), 16));
    if (!v.empty()) hipCHECK(hipMemcpy(ptr, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice));
  }
  template <typename U>
  const U* as() const { return reinterpret_cast<const U*>(ptr); }
};

// block solves through inverted diagonal blocks (denseLevels): per level list of a plan, the panels
// whose diagonal blocks are inverted (built and uploaded on the first solve)
// one step of the dense part of a solve (denseLevels): a level, a block group of a wide lump (K-B1i /
// K-B2, two launches per 256 columns) or a whole run of one-panel levels as ONE persistent sweep
// (hip_sweep_kernels.h)
struct SolveSchedItem {
  enum Kind : int32_t { kLevel = 0, kBlockGroup = 1, kSweep = 2 };
  int32_t kind = kLevel;
  int32_t slot = -1;  // first slot in the inverse scratch (block groups through inverses, sweeps)
  int64_t l = 0, e = 0;  // levels [l, e)
  hipk::SweepDesc sweep{};
  int32_t sweepFwdWgs = 0, sweepBwdWgs = 0;
};
// per level list of a plan and schedule mode (0: levels + block groups, 1: with sweeps): the steps,
// and the panels whose diagonal blocks are inverted (built and uploaded on the first solve)
struct SolveInvList {
  bool built = false;
  int64_t count = 0;
  DevBuf list;
  vector<SolveSchedItem> items;
  int32_t numSweeps = 0;
  int64_t sweepInstStride = 0;  // values per (right-hand side, batch entry) of the exchange buffer
  int64_t maxSweepWgs = 0, maxSweepWgsM = 0;  // (M: the matrix-core sweep, hip_sweep_mfma.h)
  int32_t maxSweepW = 0, maxSweepBelow = 0;
};

struct DevPlan {
  HipPlanHost host;
  // keyed by the address of a level list that lives inside `host` (host.levels, an elimination
  // range's bigLevels): the entries die with the plan, an address is never reused under them
  std::map<std::pair<const void*, int>, SolveInvList> solveInvLists;
  DevBuf panels, levelPanelDescs, trsmTasksFat, chainOffTab, rowChain, rowLocal, rowColOff, levelPanels, trsmTasks,
      updTasksFat, updTasksWide, elimChainLump, elimItems, elimPairOffJ, elimPairOffI, rowGlobal, elimLumpDesc;
  int64_t numUpdTasks = 0;
  vector<int64_t> slowPrefix;  // tasks [0, i) that updateTileBulk cannot take
  // split-K task lists of small multi-panel levels: level's updBegin -> [begin, end) in updTasksSplit
  static constexpr int64_t kSplitMaxTiles = 512;
  // (GRID 82x82, slices of 32 / 64 / 96 / 128 / 160 columns: 1.208 / 1.149 / 1.135 / 1.140 / 1.163 against
  //  1.180 ms without; profiles/r05_ab_split_k.txt)
  static constexpr int32_t kSplitMinK = 128, kSplitK = 96;
  DevBuf updTasksSplit;
  std::map<int64_t, std::pair<int64_t, int64_t>> splitRange;
  // POTRF FOLDED INTO THE TRSM LAUNCH of small multi-panel levels (trsmPanelPotrf, round 6): the levels
  // of host.levels that may take it (every panel has rows below, at most kFoldMaxTiles row tiles),
  // sorted by their number of row tiles -- the levels a call of batch b folds (tiles x b <=
  // kFoldMaxTiles) are then a PREFIX of this order, and so are their panels' descriptors in
  // foldDescs, which ONE potrfPanel launch at the end of the factorisation walks to store the factors
  static constexpr int64_t kFoldMaxTiles = 512;
  DevBuf foldDescs;
  vector<int32_t> foldRank;      // per level of host.levels: position in the sorted order, -1: not foldable
  vector<int64_t> foldTiles;     // sorted order: row tiles of the level
  vector<int64_t> foldPanelEnd;  // ... panels of the levels up to and including this one
  // forward-solve gather lists, built on the first solve that needs them
  bool solveGatherReady = false;
  SolveGatherPlan solveGather;
  DevBuf solveEntries, solveItems, solveLumpBlocks, solveLumpDescs;
  void ensureSolveGather(const CoalescedBlockMatrixSkel& skel) {
    if (solveGatherReady) return;
    solveGather = buildSolveGather(skel, host);
    solveEntries.upload(solveGather.entries);
    solveItems.upload(solveGather.items);
    solveLumpBlocks.upload(solveGather.lumpBlocks);
    solveLumpDescs.upload(solveGather.lumpDescs);
    vector<SolveGatherEntry>().swap(solveGather.entries);
    vector<SolveLumpBlock>().swap(solveGather.lumpBlocks);
    vector<SolveLumpDesc>().swap(solveGather.lumpDescs);
    solveGatherReady = true;
  }
  void upload() {
    panels.upload(host.panels);
    chainOffTab.upload(host.chainOffTab);
    rowChain.upload(host.rowChain);
    rowLocal.upload(host.rowLocal);
    rowColOff.upload(host.rowColOff);
    rowGlobal.upload(host.rowGlobal);
    levelPanels.upload(host.levelPanels);
    trsmTasks.upload(host.trsmTasks);
    {
      // the factor kernels of multi-panel levels read self-contained records (one uniform load
      // instead of index -> descriptor); the solve kernels keep the index lists
      vector<PanelDesc> lp(host.levelPanels.size());
      for (size_t i = 0; i < lp.size(); i++) lp[i] = host.panels[host.levelPanels[i]];
      levelPanelDescs.upload(lp);
      vector<TrsmTaskFat> tf(host.trsmTasks.size());
      for (size_t i = 0; i < tf.size(); i++) {
        const PanelDesc& pd = host.panels[host.trsmTasks[i].panel];
        tf[i] = TrsmTaskFat{pd.diagOff, pd.lda, pd.nb, pd.rowsBelow, host.trsmTasks[i].rowTile, 0, 0};
      }
      trsmTasksFat.upload(tf);
      // levels whose potrf may be folded into their trsm launch
      foldRank.assign(host.levels.size(), -1);
      vector<std::pair<int64_t, size_t>> cand;  // (row tiles, level)
      for (size_t li = 0; li < host.levels.size(); li++) {
        const LevelRange& lr = host.levels[li];
        const int64_t nT = lr.trsmEnd - lr.trsmBegin;
        if (lr.tail || lr.directPanel >= 0 || lr.panelEnd <= lr.panelBegin || nT < 1 || nT > kFoldMaxTiles) continue;
        bool allBelow = true;
        for (int64_t q = lr.panelBegin; q < lr.panelEnd; q++) allBelow = allBelow && lp[q].rowsBelow > 0;
        if (allBelow) cand.push_back({nT, li});
      }
      std::stable_sort(cand.begin(), cand.end());
      vector<PanelDesc> fd;
      foldTiles.clear();
      foldPanelEnd.clear();
      for (size_t k = 0; k < cand.size(); k++) {
        const LevelRange& lr = host.levels[cand[k].second];
        foldRank[cand[k].second] = (int32_t)k;
        fd.insert(fd.end(), lp.begin() + lr.panelBegin, lp.begin() + lr.panelEnd);
        foldTiles.push_back(cand[k].first);
        foldPanelEnd.push_back((int64_t)fd.size());
      }
      foldDescs.upload(fd);
    }
    {
      vector<UpdTaskFat> fat(host.updTasks.size());
      vector<UpdTaskWide> wide(host.updTasks.size());
      slowPrefix.assign(host.updTasks.size() + 1, 0);
      for (size_t i = 0; i < host.updTasks.size(); i++) {
        const UpdTask& t = host.updTasks[i];
        const SegDesc& sd = host.segs[t.seg];
        const SrcDesc& sr = host.srcs[sd.src];
        UpdTaskFat& f = fat[i];
        f.srcOff = sr.off;
        f.tgtBase = sd.tgtBase;
        f.lda = sr.lda;
        f.K = sr.K;
        f.rowsBelow = sr.rowsBelow;
        f.segEnd = sd.q0 + sd.m;
        f.rowTile = t.rowTile;
        f.colTile = t.colTile;
        f.tgtStride = sd.tgtStride;
        f.rowMin = sd.rowMin;
        f.atomic = t.atomic;
        f.fast = sd.kind == kSegIntra && sr.K > 0 && sr.K % hipk::kUpdChunk == 0;
        f.pad0 = f.pad1 = 0;
        slowPrefix[i + 1] = slowPrefix[i] + (f.fast ? 0 : 1);
        UpdTaskWide& w = wide[i];
        w.srcOff = sr.off;
        w.tgtBase = sd.tgtBase;
        w.chainTabPtr = sd.chainTabPtr;
        w.lda = sr.lda;
        w.K = sr.K;
        w.rowsBelow = sr.rowsBelow;
        w.nRest = sr.nRest;
        w.lumpRowBase = sr.lumpRowBase;
        w.kind = sd.kind;
        w.segEnd = sd.q0 + sd.m;
        w.tgtStride = sd.tgtStride;
        w.firstChainOrd = sd.firstChainOrd;
        w.rowMin = sd.rowMin;
        w.rowTile = t.rowTile;
        w.colTile = t.colTile;
        w.atomic = t.atomic;
        w.pad0 = w.pad1 = w.pad2 = w.pad3 = w.pad4 = 0;
      }
      updTasksFat.upload(fat);
      updTasksWide.upload(wide);
      // SPLIT-K lists (round 5) for the multi-panel levels of latency-bound structures.  A tile of a
      // block-wide source walks its K = 128 .. 256 columns in chunks of 32, one exposed memory round trip
      // per chunk (1.6 us, tools/trace_upd.py): 17-19 us per launch where a level of rank-64 tiles takes
      // 10, on a GPU that is mostly idle.  For small levels every such tile is listed again as ceil(K /
      // 96) tiles of at most 96 source columns each that accumulate with atomics; launchLevels takes
      // that list when the launch (tiles x batch) stays within a round of workgroups.
      vector<UpdTaskWide> split;
      auto addSplit = [&](const vector<LevelRange>& levels) {
        for (const LevelRange& lr : levels) {
          const int64_t n = lr.updEnd - lr.updBegin;
          if (lr.directPanel >= 0 || n <= 0 || n > kSplitMaxTiles) continue;
          int64_t extra = 0;
          for (int64_t i = lr.updBegin; i < lr.updEnd; i++) {
            if (wide[i].K >= kSplitMinK) extra += (wide[i].K + kSplitK - 1) / kSplitK - 1;
          }
          if (extra == 0 || n + extra > 2 * kSplitMaxTiles) continue;
          const int64_t b = (int64_t)split.size();
          for (int64_t i = lr.updBegin; i < lr.updEnd; i++) {
            const UpdTaskWide& w = wide[i];
            if (w.K < kSplitMinK) {
              split.push_back(w);
              continue;
            }
            for (int32_t k0 = 0; k0 < w.K; k0 += kSplitK) {
              UpdTaskWide p = w;
              p.srcOff = w.srcOff + k0;
              p.K = std::min<int32_t>(kSplitK, w.K - k0);
              p.atomic = w.atomic | 1;
              split.push_back(p);
            }
          }
          splitRange[lr.updBegin] = {b, (int64_t)split.size()};
        }
      };
      for (const auto& er : host.elimRanges) addSplit(er.bigLevels);
      addSplit(host.levels);
      updTasksSplit.upload(split);
    }
    elimChainLump.upload(host.elimChainLump);
    elimLumpDesc.upload(host.elimLumpDesc);
    elimItems.upload(host.elimItems);
    elimPairOffJ.upload(host.elimPairOffJ);
    elimPairOffI.upload(host.elimPairOffI);
    // the launch code only needs the descriptors it passes by value and the level / range
    // tables: release the host copies of everything that now lives on the device (BAL-871:
    // 0.3 GB of pair offsets and task lists)
    numUpdTasks = (int64_t)host.updTasks.size();
    auto drop = [](auto& v) { std::decay_t<decltype(v)>().swap(v); };
    drop(host.chainOffTab);
    drop(host.rowChain);
    drop(host.rowLocal);
    drop(host.rowColOff);
    drop(host.rowGlobal);
    drop(host.levelPanels);
    drop(host.trsmTasks);
    drop(host.updTasks);
    drop(host.elimChainLump);
    drop(host.elimLumpDesc);
    drop(host.elimItems);
    drop(host.elimPairOffJ);
    drop(host.elimPairOffI);
  }
};

// Pointer arrays of a batched call (one device pointer per matrix / vector) travel through a small
// ring of pinned host slots and device slots, copied on the execution stream: no synchronous
// hipMemcpy, no per-call allocation, and a slot is not rewritten while an earlier call's kernels
// may still read it (the reference re-uploads the array for every op, MatOpsCuda.cu:730,761,788).
struct PtrRing {
  static constexpr int kSlots = 32;
  char* host = nullptr;
  DevBuf dev;
  size_t slotBytes = 0;
  int next = 0;
  hipEvent_t ev[kSlots] = {};
  bool used[kSlots] = {};
  ~PtrRing() { release(); }
  void release() {
    for (int i = 0; i < kSlots; i++) {
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      ev[i] = nullptr;
      used[i] = false;
    }
    if (host) (void)hipHostFree(host);
    host = nullptr;
    dev.release();
    slotBytes = 0;
  }
  const void* push(const void* src, size_t bytes, hipStream_t stream) {
    if (bytes > slotBytes) {  // (first call, or a larger batch than ever before)
      if (slotBytes) hipCHECK(hipDeviceSynchronize());
      release();
      slotBytes = (std::max<size_t>(bytes, 512) + 255) & ~size_t(255);
      hipCHECK(hipHostMalloc((void**)&host, slotBytes * kSlots, hipHostMallocDefault));
      dev.resize(slotBytes * kSlots);
      for (int i = 0; i < kSlots; i++) hipCHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    const int s = next;
    next = (next + 1) % kSlots;
    if (used[s]) hipCHECK(hipEventSynchronize(ev[s]));
    std::memcpy(host + s * slotBytes, src, bytes);
    char* d = reinterpret_cast<char*>(dev.ptr) + s * slotBytes;
    hipCHECK(hipMemcpyAsync(d, host + s * slotBytes, bytes, hipMemcpyHostToDevice, stream));
    hipCHECK(hipEventRecord(ev[s], stream));
    used[s] = true;
    return d;
  }
};

struct HipSymElimCtx : SymElimCtx {
  int64_t lumpsBegin = 0, lumpsEnd = 0;
};

// launch recorder for the profiled mode (HIP events on the execution stream)
struct LaunchTimer {
  hipStream_t stream;
  HipKernelProfile* prof;
  struct Rec { int kind; hipEvent_t a, b; hipStream_t on; };
  vector<Rec> recs;
  LaunchTimer(hipStream_t s, HipKernelProfile* p) : stream(s), prof(p) {}
  // `on`: the stream the launch goes to (in-situ mode times the side-stream launches there)
  void begin(int kind, hipStream_t on = nullptr) {
    if (!prof) return;
    Rec r{kind, nullptr, nullptr, on ? on : stream};
    hipCHECK(hipEventCreate(&r.a));
    hipCHECK(hipEventCreate(&r.b));
    hipCHECK(hipEventRecord(r.a, r.on));
    recs.push_back(r);
  }
  void end() {
    if (!prof) return;
    hipCHECK(hipEventRecord(recs.back().b, recs.back().on));
  }
  void finish() {
    if (!prof) return;
    hipCHECK(hipStreamSynchronize(stream));  // (the side stream has been joined into it)
    // per class: sum of the launch durations, and the union of their intervals (device time
    // stamps relative to the first launch)
    vector<std::pair<float, float>> spans[kProfNumKinds];
    for (auto& r : recs) {
      float ms = 0, t0 = 0;
      hipCHECK(hipEventElapsedTime(&ms, r.a, r.b));
      if (&r != &recs.front()) hipCHECK(hipEventElapsedTime(&t0, recs.front().a, r.a));
      prof->ms[r.kind] += ms;
      prof->launches[r.kind]++;
      spans[r.kind].emplace_back(t0, t0 + ms);
    }
    for (int k = 0; k < kProfNumKinds; k++) {
      std::sort(spans[k].begin(), spans[k].end());
      float end = -1e30f;
      for (auto& sp : spans[k]) {
        if (sp.second <= end) continue;
        prof->busyMs[k] += sp.second - std::max(sp.first, end);
        end = sp.second;
      }
    }
    for (auto& r : recs) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
};

// The auxiliary streams of the lookahead schedule are shared by every Solver of the process, one
// set per device, created on first use and kept for the life of the process.  HIP multiplexes its
// streams onto a handful of hardware queues (4 by default): with streams of their own, a second
// Solver's side streams landed on the queue of the execution stream and its lookahead work ran
// behind the chain instead of beside it (64 x GRID 17.4 instead of 14.5 ms, the fp32 BAL-1723 factor
// 26 instead of 19 ms, while another Solver was merely alive: tools/exp_order.py).  One factor() at a
// time per Solver is the contract (Solver.h); two Solvers factoring concurrently share these
// streams and are merely ordered on them.
struct SharedStreams {
  hipStream_t side = nullptr, due = nullptr;
  hipStream_t batch[3] = {nullptr, nullptr, nullptr};  // further parts of a batch factored as concurrent sub-batches (normal priority)
};
inline SharedStreams& sharedStreams() {
  static std::mutex mu;
  static std::map<int, SharedStreams> perDevice;
  int dev = 0;
  hipCHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  SharedStreams& st = perDevice[dev];
  if (!st.side) {
    int least = 0, greatest = 0;
    hipCHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // lowest priority; neither stream priorities nor CU masks separate the bulk tiles from the
    // chain on this stack (hipExtStreamCreateWithCUMask is not honoured: a masked saturating kernel
    // ran on all 256 CUs, tools/throttle_probe.hip), what does is s_setprio inside the chain kernels
    hipCHECK(hipStreamCreateWithPriority(&st.side, hipStreamNonBlocking, least));
    // (highest priority for the due units: measured, no effect)
    hipCHECK(hipStreamCreateWithPriority(&st.due, hipStreamNonBlocking, least));
    for (hipStream_t& b : st.batch) hipCHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  }
  return st;
}

// per-op statistics of the per-op boundary (OpStat, Utils.h:48-121 of the reference): HIP events
// around the launches of one op, read back synchronously -- only while stats are enabled
struct OpTimer {
  OpStat& st;
  hipStream_t stream;
  hipEvent_t a = nullptr, b = nullptr;
  double s0, s1, s2;
  OpTimer(OpStat& st_, hipStream_t stream_, double s0_ = 0, double s1_ = 0, double s2_ = 0)
      : st(st_), stream(stream_), s0(s0_), s1(s1_), s2(s2_) {
    if (!st.enabled) return;
    hipCHECK(hipEventCreate(&a));
    hipCHECK(hipEventCreate(&b));
    hipCHECK(hipEventRecord(a, stream));
  }
  ~OpTimer() {
    if (!a) return;
    float ms = 0;
    if (hipEventRecord(b, stream) == hipSuccess && hipEventSynchronize(b) == hipSuccess &&
        hipEventElapsedTime(&ms, a, b) == hipSuccess) {
      st.add(ms * 1e-3, s0, s1, s2);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
};

// restores a flag on every exit (an exception from hipCHECK / launchLevels included)
struct FlagOff {
  bool& flag;
  const bool saved;
  explicit FlagOff(bool& f) : flag(f), saved(f) { flag = false; }
  ~FlagOff() { flag = saved; }
  FlagOff(const FlagOff&) = delete;
  FlagOff& operator=(const FlagOff&) = delete;
};

struct HipSymbolicCtx : SymbolicCtx {
  // `o`: the caller's switches with the environment already applied (backend_options.h)
  HipSymbolicCtx(const CoalescedBlockMatrixSkel& skel_, const vector<int64_t>& permutation_,
                 const HipBackendOptions& o)
      : skel(skel_), permutation(permutation_) {
    lookaheadEnabled = HipBackendOptions::on(o.lookahead, true);
    blockSolve = HipBackendOptions::on(o.blockSolve, true);
    solveInv = HipBackendOptions::on(o.solveInv, true);
    if (o.subBatchMin >= 0) subBatchMin = o.subBatchMin == 0 ? INT32_MAX : std::max(2, o.subBatchMin);
    if (o.subBatches >= 0) subBatchParts = std::max(2, o.subBatches);
    splitK = HipBackendOptions::on(o.splitK, true);
    sweepEnabled = HipBackendOptions::on(o.solveSweep, true);
    solveWide = HipBackendOptions::on(o.solveWide, true);
    gatherOverlap = HipBackendOptions::on(o.gatherOverlap, false);
    if (o.sweepMinWidth >= 0) sweepMinWidth = std::max(1, o.sweepMinWidth);
    lazyPlan = HipBackendOptions::on(o.lazyPlan, false);
    if (!std::isnan(o.lookaheadMinGF)) lookaheadMinFlops = 1e9 * o.lookaheadMinGF;
    // the plan builder's switches: handed to every buildHipPlan call and recorded in the plan;
    // launchLevels takes dueStream from the plan it runs
    planOpts.dueStream = HipBackendOptions::on(o.dueStream, planOpts.dueStream);
    if (o.gatherMaxPairs >= 0) planOpts.gatherMaxPairs = std::max(8, o.gatherMaxPairs);
    if (!std::isnan(o.bulkAhead)) planOpts.bulkAhead = o.bulkAhead;
    if (o.tailBlocks >= 0) planOpts.tailBlocks = o.tailBlocks;
    planOpts.gatherOverlap = gatherOverlap;
    planOpts.applyDeveloperEnv();
    // developer aids and experiment parameters that are not options of the product
    if (const char* e = std::getenv("BSP_TAIL_FLAGS")) tailFlags = std::atoi(e);
    if (const char* e = std::getenv("BSP_GATHER_OVERLAP_LDS")) gatherOverlapLds = (unsigned)std::max(0, std::atoi(e));
    if (const char* e = std::getenv("BSP_SWEEP_TRACE")) sweepTraceOn = e[0] != '0';
    if (const char* e = std::getenv("BSP_SWEEP_MFMA_MIN")) sweepMfmaMinRhs = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("BSP_SWEEP_MFMA_MIN_WIDTH")) sweepMfmaMinWidth = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("BSP_SOLVE_WIDE_MIN")) solveWideMinRhs = std::max(2, std::atoi(e));
    if (const char* e = std::getenv("BSP_POTRF_IN_TRSM")) potrfInTrsm = e[0] != '0';
    if (const char* e = std::getenv("BSP_TAIL_IN_BATCH")) tailInBatch = e[0] != '0';
    if (const char* e = std::getenv("BSP_FOLD_MAX_TILES")) foldMaxTiles = std::min<int64_t>(DevPlan::kFoldMaxTiles, std::max(1, std::atoi(e)));
  }

  virtual ~HipSymbolicCtx() override {
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    if (sweepHostErr) (void)hipHostFree(sweepHostErr);
  }

  virtual void setSparseElimRanges(const vector<int64_t>& ranges) override {
    sparseElimRanges = ranges;
    plans.clear();
  }

  virtual void prepareFactor(int64_t upToLump) override {
    try {
      prepareDevice(upToLump);
    } catch (const std::exception&) {
      // (eager preparation is an optimisation: whatever it could not do happens -- and fails
      //  loudly, if it has to -- in the first factor().  Nothing of the eager state has been handed
      //  out: release ALL of it and forget the device, so that a first real use on another device
      //  binds there as the lazy path of rounds 1-3 did)
      (void)hipGetLastError();
      releaseDeviceState();
      device = -1;
      eagerOnly = false;
      inPrepare = false;
    }
  }

  // everything this context has put on the GPU (a device buffer is freed on whatever device is
  // current; events and plans likewise)
  void releaseDeviceState() {
    plans.clear();
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    events.clear();
    nextEvent = 0;
    dinvScratch.release();
    rawScratch.release();
    yieldBuf.release();
    solveInvScratch.release();
    sweepXchg.release();
    tailCtl.release();
    tailDinv.release();
    for (DevBuf* b : {&dSpanStart, &dSpanToLump, &dLumpStart, &dSpanOffsetInLump, &dChainColPtr,
                      &dChainRowSpan, &dChainData, &dChainRowsTillEnd, &dBoardColPtr,
                      &dBoardChainColOrd, &dPermutation}) {
      b->release();
    }
    skelUploaded = false;
    shared = nullptr;
    sweepAttr[0] = sweepAttr[1] = 0;
  }

  // The reference builds its SymbolicCtx and every SymElimCtx in the Solver constructor
  // (Solver.cpp:24-40), so that its first factor() costs what every later one does.  Same here: when
  // a GPU is visible at construction, the full-range factor plan is built and uploaded, the skeleton
  // mirrors, scratch buffers, auxiliary streams and the event pool are created and the kernels' code
  // object is loaded (one empty launch) -- all on the device that is current NOW.  A Solver whose
  // first real use happens with another device current re-binds there (nothing of the eager state
  // has been handed out yet); plans of partial ranges stay lazy.  Without a GPU (symbolic analysis
  // on a host-only box) nothing happens here.
  void prepareDevice(int64_t upToLump) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
      (void)hipGetLastError();
      return;
    }
    if (lazyPlan) return;  // (A/B: the lazy behaviour of rounds 1-3)
    struct Scope {
      bool& f;
      explicit Scope(bool& f_) : f(f_) { f = true; }
      ~Scope() { f = false; }
    } scope(inPrepare);
    if (upToLump <= 0) return;
    DevPlan& plan = planFor(sparseElimRanges, 0, upToLump, /*tag=*/0);
    ensureSkelOnDevice();
    (void)streams();
    (void)yieldWord();
    // scratch of the chain kernels, sized for one fp64 matrix (grown later for batches)
    dinvScratch.resize((size_t)hipk::kDinvBatchStride * sizeof(double));
    if (plan.host.maxChainRows > 0) {
      rawScratch.resize((size_t)2 * plan.host.maxChainRows * kTile * sizeof(double));
    }
    const size_t wantEvents = 8 + 4 * (size_t)plan.host.numForkLevels;
    while (events.size() < wantEvents) {
      hipEvent_t e;
      hipCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events.push_back(e);
    }
    // code object load: one empty launch on a private non-blocking stream (the context's own
    // stream is the legacy default stream at construction: a launch + sync there would also wait for
    // every blocking stream of the process -- PyTorch's included -- and is illegal during a capture)
    {
      hipStream_t warm = nullptr;
      hipCHECK(hipStreamCreateWithFlags(&warm, hipStreamNonBlocking));
      hipk::warmupKernel<<<1, 64, 0, warm>>>();
      const hipError_t werr = hipStreamSynchronize(warm);
      (void)hipStreamDestroy(warm);
      hipCHECK(werr);
    }
    eagerOnly = true;
  }

  virtual void setStream(void* s) override { stream = (hipStream_t)s; }

  // Everything a Solver puts on the GPU (skeleton mirrors, plans, scratch, streams, events) lives
  // on the device that was current at its first use; later calls must run with that device current.
  void checkDevice() {
    int cur = -1;
    hipCHECK(hipGetDevice(&cur));
    if (device < 0) device = cur;
    if (cur != device && eagerOnly) {
      // only the constructor's eager state lives on `device`: move to the device of the first use
      releaseDeviceState();
      device = cur;
    }
    if (!inPrepare) eagerOnly = false;
    if (cur != device) {
      throw std::runtime_error("HIP backend: this Solver was first used on device " +
                               std::to_string(device) + " but the current device is " +
                               std::to_string(cur) + " (make its device current before the call)");
    }
  }

  void ensureSkelOnDevice() {
    if (skelUploaded) return;
    checkDevice();
    dSpanStart.upload(skel.spanStart);
    dSpanToLump.upload(skel.spanToLump);
    dLumpStart.upload(skel.lumpStart);
    dSpanOffsetInLump.upload(skel.spanOffsetInLump);
    dChainColPtr.upload(skel.chainColPtr);
    dChainRowSpan.upload(skel.chainRowSpan);
    dChainData.upload(skel.chainData);
    dChainRowsTillEnd.upload(skel.chainRowsTillEnd);
    dBoardColPtr.upload(skel.boardColPtr);
    dBoardChainColOrd.upload(skel.boardChainColOrd);
    dPermutation.upload(permutation);
    skelUploaded = true;
  }

  hipk::SkelDev skelDev() {
    ensureSkelOnDevice();
    hipk::SkelDev d;
    d.spanStart = dSpanStart.as<int64_t>();
    d.spanToLump = dSpanToLump.as<int64_t>();
    d.lumpStart = dLumpStart.as<int64_t>();
    d.spanOffsetInLump = dSpanOffsetInLump.as<int64_t>();
    d.chainColPtr = dChainColPtr.as<int64_t>();
    d.chainRowSpan = dChainRowSpan.as<int64_t>();
    d.chainData = dChainData.as<int64_t>();
    d.chainRowsTillEnd = dChainRowsTillEnd.as<int64_t>();
    d.boardColPtr = dBoardColPtr.as<int64_t>();
    d.boardChainColOrd = dBoardChainColOrd.as<int64_t>();
    return d;
  }

  virtual PermutedCoalescedAccessor deviceAccessor() override {
    ensureSkelOnDevice();
    PermutedCoalescedAccessor acc;
    acc.init(dSpanStart.as<int64_t>(), dSpanToLump.as<int64_t>(), dLumpStart.as<int64_t>(),
             dSpanOffsetInLump.as<int64_t>(), dChainColPtr.as<int64_t>(),
             dChainRowSpan.as<int64_t>(), dChainData.as<int64_t>(), dPermutation.as<int64_t>());
    return acc;
  }

  virtual SymElimCtxPtr prepareElimination(int64_t lumpsBegin, int64_t lumpsEnd) override {
    HipSymElimCtx* e = new HipSymElimCtx;
    e->lumpsBegin = lumpsBegin;
    e->lumpsEnd = lumpsEnd;
    return SymElimCtxPtr(e);
  }

  // plan of a fused factor over lumps [startLump, upToLump), built and uploaded on first use
  // tag 0: factor() / solve(); tag 1: one elimination range; tag 2: factor() of a BATCH when plan 0 has a
  // persistent tail -- the same plan without it.  The tail is a latency device for ONE matrix: in a batch
  // the chain steps of the matrices already run side by side, and batch x roles exceed what is resident
  // (batch of 8 GRID 82x82 2.23 -> 2.35 ms with it; the reference's FLAT families at batch 16 +22..+36 % per
  // matrix, profiles/r06_tail_batches.txt)
  DevPlan& planFor(const vector<int64_t>& ranges, int64_t startLump, int64_t upToLump, int tag) {
    checkDevice();
    auto key = std::make_tuple(tag, startLump, upToLump);
    auto it = plans.find(key);
    if (it == plans.end()) {
      std::unique_ptr<DevPlan> p(new DevPlan);
      HipPlanOptions po = planOpts;
      if (tag == 2) po.tailBlocks = 0;
      p->host = buildHipPlan(skel, ranges, startLump, upToLump, po);
      p->upload();
      it = plans.emplace(key, std::move(p)).first;
    }
    return *it->second;
  }

  virtual NumericCtxBase* createNumericCtxForType(std::type_index tIdx, int64_t tempBufSize,
                                                  int batchSize) override;

  virtual SolveCtxBase* createSolveCtxForType(std::type_index tIdx, int nRHS,
                                              int batchSize) override;

  // side streams (shared, see sharedStreams) + event pool of the lookahead schedule
  // (resolved once per Solver, after checkDevice pinned its device: sharedStreams() takes a global
  //  mutex and asks for the current device, several times per level on a launch-bound path)
  SharedStreams& streams() {
    if (!shared) {
      checkDevice();
      shared = &sharedStreams();
    }
    return *shared;
  }
  hipStream_t sideStream() { return streams().side; }
  // word of device memory through which the chain's potrf workgroup tells the bulk tiles which CU
  // it runs on (cooperative CU yield, hip_kernels.h)
  unsigned* yieldWord() {
    if (!yieldBuf.ptr) {
      const size_t bytes = 256;  // word 0: the potrf workgroup's CU
      yieldBuf.resize(bytes);
      hipCHECK(hipMemset(yieldBuf.ptr, 0, bytes));
    }
    return reinterpret_cast<unsigned*>(yieldBuf.ptr);
  }
  hipStream_t dueSideStream() { return streams().due; }
  hipEvent_t eventFromPool() {
    if (nextEvent == events.size()) {
      hipEvent_t e;
      hipCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events.push_back(e);
    }
    return events[nextEvent++];
  }
  void resetEventPool() { nextEvent = 0; }

  const CoalescedBlockMatrixSkel& skel;
  vector<int64_t> permutation;
  vector<int64_t> sparseElimRanges;
  hipStream_t stream = nullptr;
  HipKernelProfile* profile = nullptr;
  bool profileInSitu = false;  // profile with the lookahead schedule left on (two streams)
  bool lookaheadEnabled = true;
  bool lazyPlan = false;  // HipBackendOptions::lazyPlan
  // LDS padding of side-stream launches: three bulk workgroups per CU, so that a chain workgroup
  // always finds a slot
  static constexpr unsigned bulkExtraLds = 6 * 1024, dueExtraLds = 6 * 1024;
  bool forcePerOp = false;  // TESTING: drive factor() through the per-op boundary
  // TESTING, fault injection (bsp_test_set_fault, include/baspacho_amd_testing.h; never read from the
  // environment): the sparse-elimination update is not launched, so the factor is wrong and the
  // full-size checks must notice
  bool faultDropElimUpdate = false;
  SharedStreams* shared = nullptr;  // this device's auxiliary streams (streams())
  HipPlanOptions planOpts;     // the plan builder's switches (BSP_DUE_STREAM, BSP_BULK_AHEAD, ...)
  double lookaheadMinFlops = HipPlanHost::kMinDeferredFlopsPerFork;  // BSP_LOOKAHEAD_MIN_GF (0: side streams whenever a plan has lookahead units)
  bool blockSolve = true;   // wide lumps: triangular solves by outer block (BSP_BLOCK_SOLVE=0: by panel)
  bool tailInBatch = false;  // developer (BSP_TAIL_IN_BATCH=1): batches run the plan WITH the persistent tail (A/B record)
  bool potrfInTrsm = true;  // small multi-panel levels: potrf folded into the trsm launch (developer: BSP_POTRF_IN_TRSM=0)
  int64_t foldMaxTiles = 512;  // ... levels of at most this many row tiles x batch (developer: BSP_FOLD_MAX_TILES)
  bool splitK = true;       // split-K tile lists for small multi-panel levels (BSP_SPLIT_K=0: off)
  unsigned gatherOverlapLds = 0;  // dynamic LDS of the overlapped chunks' launches (throttle; BSP_GATHER_OVERLAP_LDS)
  bool gatherOverlap = false;  // gather chunks beside the dense chain (BSP_GATHER_OVERLAP=1; off: the plan is not chunked either)
  static constexpr int64_t splitKMaxWgs = 1024;  // ... launches of at most this many workgroups (a batch of 8: 2.26 = 2.26 ms at 1024, 2.48 at 4096)
  int subBatchMin = 16;     // batches of at least this many matrices are factored as concurrent sub-batches (BSP_SUB_BATCH_MIN; 0: never)
  int subBatchParts = 2;    // ... this many (BSP_SUB_BATCHES, at most 4, at least subBatchMin / 2 matrices each)
  vector<hipEvent_t> events;
  size_t nextEvent = 0;
  int device = -1;  // device of the first use (checkDevice)
  bool inPrepare = false;
  bool eagerOnly = false;  // so far only prepareDevice() has touched `device` (cleared by the first plan use)
  int traceLaunchId = 0;  // ordinal of the chain-step launches (trace builds only)
  // scratch that outlives the per-call NumericCtx / SolveCtx objects (those are created and
  // destroyed around every factor() / solve(), Solver.cpp:176,223, while their kernels may still be
  // queued): sized on first use, only ever grown
  DevBuf dinvScratch, rawScratch, yieldBuf;
  // block solves through inverted diagonal blocks (denseLevels): the inverses of one solve call, and
  // per level list the panels whose diagonal blocks are inverted (uploaded once)
  DevBuf solveInvScratch;
  uint64_t solveInvGen = 0;  // bumped by every launch that writes solveInvScratch
  bool solveInv = true;  // BSP_SOLVE_INV=0: the substitution kernels of rounds 1-2
  // ---- persistent sweeps over wide lumps (hip_sweep_kernels.h, round 6)
  bool sweepEnabled = true;   // BSP_SOLVE_SWEEP=0: the multi-launch block path
  int sweepMinWidth = 768;    // runs of one-panel levels at least this wide (BSP_SWEEP_MIN_WIDTH)
  // ... and the narrowest run of columns it takes (ten right-hand sides, matrix-core sweep against the block path:
  // 2751 / 2814 columns +23 / +22 %, 6101 +7 %, 2442 +3 %; 7839 -8 %, 9756 -20 %; developer: BSP_SWEEP_MFMA_MIN_WIDTH)
  int sweepMfmaMinWidth = 7000;
  int sweepMfmaMinRhs = 8;    // right-hand sides from which the matrix-core sweep takes over (BAL-871: equal to the multi-launch path at 6, -5 % at 10, -10 % at 16; developer: BSP_SWEEP_MFMA_MIN)
  bool sweepBroken = false;   // a sweep timed out or cannot be launched here: multi-launch path for good
  int sweepFault = 0;         // TESTING (bsp_test_set_fault kind 2): spine of block 1 never publishes
  double sweepSpinLimitS = 2.0;  // watchdog: a spin that lasts longer aborts the launch
  int tailFlags = 1;          // TailDesc::flags (BSP_TAIL_FLAGS)
  DevBuf tailCtl, tailDinv;   // persistent tail (hip_tail_kernel.h): control words, inverted diagonal blocks
  bool solveWide = true;      // backward elimination pass with the right-hand sides across the lanes (hip_solve_wide.h)
  int solveWideMinRhs = 2;    // (developer: BSP_SOLVE_WIDE_MIN)
  size_t solveWideMaxBytes = (size_t)1 << 30;
  DevBuf solveWideBuf;        // [row][16] copy of the rows below an elimination range (hip_solve_wide.h)
  DevBuf sweepXchg;           // control words + exchange values of one denseLevels call
  DevBuf sweepTrace;          // developer aid (BSP_SWEEP_TRACE=1): clock stamps of the spines of the last sweep
  bool sweepTraceOn = false;
  unsigned* sweepHostErr = nullptr;     // pinned host word a timed-out sweep raises ...
  unsigned* sweepHostErrDev = nullptr;  // ... and its device alias
  int sweepCapacity = 0;      // workgroups of a sweep launch that are resident at once (one per CU)
  size_t sweepMaxLds = 0;
  int sweepAttr[2] = {0, 0};  // per value size (8, 4): 0 not tried, 1 ready, -1 failed
  struct RunCounters {
    int64_t sweepLaunches = 0, sweepTimeouts = 0, splitListsUsed = 0, subBatchesEnqueued = 0,
            lookaheadForks = 0, gatherChunksOverlapped = 0, tailLaunches = 0, sweepMfmaLaunches = 0, invReused = 0, solveWideLaunches = 0, potrfFoldedLevels = 0;
  } counters;
  // a timed-out persistent launch (solve sweep, factor tail) is reported ONCE, by the next factor()
  // or solve() on this Solver; the persistent kernels are then retired for good
  void checkAsyncError() {
    if (sweepHostErr && *reinterpret_cast<volatile unsigned*>(sweepHostErr) != 0u) {
      *reinterpret_cast<volatile unsigned*>(sweepHostErr) = 0u;
      sweepBroken = true;
      counters.sweepTimeouts++;
      if (planOpts.tailBlocks > 0) {
        (void)hipDeviceSynchronize();  // (nothing of the old plans may still be queued)
        planOpts.tailBlocks = 0;
        plans.clear();
      }
      throw std::runtime_error(
          "HIP backend: a persistent launch (solve sweep / factor tail) of an EARLIER call on this "
          "Solver timed out (watchdog); the result of that call is invalid.  The persistent kernels "
          "are retired for this Solver: it takes the multi-launch paths from now on");
    }
  }
  bool sweepUsable() {
    checkAsyncError();
    return sweepEnabled && !sweepBroken && solveInv && blockSolve;
  }
  long long* tailTrace() {  // developer aid (BSP_SWEEP_TRACE=1): the spines' clock stamps
    if (!sweepTraceOn) return nullptr;
    if (!sweepTrace.ptr) {
      sweepTrace.resize(8 * 4096 * sizeof(long long));
      hipCHECK(hipMemset(sweepTrace.ptr, 0, 8 * 4096 * sizeof(long long)));
    }
    return reinterpret_cast<long long*>(sweepTrace.ptr);
  }
  // pinned host word a watchdog raises (persistent sweeps and tails), and its device alias
  unsigned* asyncErrWord() {
    if (!sweepHostErr) {
      void* hp = nullptr;
      hipCHECK(hipHostMalloc(&hp, 64, hipHostMallocMapped));
      sweepHostErr = reinterpret_cast<unsigned*>(hp);
      *sweepHostErr = 0u;
      void* dp = nullptr;
      hipCHECK(hipHostGetDevicePointer(&dp, hp, 0));
      sweepHostErrDev = reinterpret_cast<unsigned*>(dp);
    }
    return sweepHostErrDev;
  }
  template <typename BT>
  bool sweepReady() {
    int& st = sweepAttr[sizeof(BT) == 8 ? 0 : 1];
    if (st == 0) {
      st = -1;
      checkDevice();
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        sweepCapacity = prop.multiProcessorCount;
        sweepMaxLds = std::min<size_t>(prop.maxSharedMemoryPerMultiProcessor, 160 * 1024) - 1024;
        const int lds = (int)sweepMaxLds;
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&hipk::solveSweep<BT, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&hipk::solveSweep<BT, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&hipk::solveSweepM<BT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&hipk::solveSweepM<BT, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        if (ok && !sweepHostErr) {
          void* hp = nullptr;
          ok = hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess;
          if (ok) {
            sweepHostErr = reinterpret_cast<unsigned*>(hp);
            *sweepHostErr = 0u;
            void* dp = nullptr;
            ok = hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess;
            sweepHostErrDev = reinterpret_cast<unsigned*>(dp);
          }
        }
        if (ok) st = 1;
      }
      if (st < 0) (void)hipGetLastError();
    }
    return st > 0;
  }
#ifndef BSP_UPD_PREFETCH_MAX_WGS
#define BSP_UPD_PREFETCH_MAX_WGS 2048
#endif
  static constexpr int64_t updPrefetchMaxWgs = BSP_UPD_PREFETCH_MAX_WGS;  // updateTile<PREFETCH> for launches of up to this many workgroups
  PtrRing ptrRing;
  std::map<std::pair<int64_t, int64_t>, std::pair<std::unique_ptr<DevBuf>, size_t>> addMvTileLists;

  bool skelUploaded = false;
  DevBuf dSpanStart, dSpanToLump, dLumpStart, dSpanOffsetInLump, dChainColPtr, dChainRowSpan,
      dChainData, dChainRowsTillEnd, dBoardColPtr, dBoardChainColOrd, dPermutation;
  std::map<std::tuple<int, int64_t, int64_t>, std::unique_ptr<DevPlan>> plans;
};

template <typename T>
struct HipNumericCtx : NumericCtx<T> {
  using BT = BaseType<T>;

  HipNumericCtx(HipSymbolicCtx& sym_, int batchSize_, int64_t tempBufSize_)
      : sym(sym_), batchSize(batchSize_), tempBufSize(std::max<int64_t>(tempBufSize_, 1)) {}

  // single matrix: pointer by value; batch: the device-pointer array is uploaded once per call
  hipk::DataRef<BT> makeRef(T* data);

  // extraLds: dynamic LDS requested on top of the kernel's static 35 KB.  Bulk (deferred) launches
  // ask for 6 KB so that only THREE of their workgroups fit on a CU (3 x 41 KB), leaving 37 KB for
  // a workgroup of the critical-path kernels (trsm needs 33.5 KB); with four resident bulk
  // workgroups the trsm of the next panel was starved for the whole bulk update (273 us vs 17 us).
  // atomicMask: which bits of a task's `atomic` field count (bit 1 = "launches of two side streams
  // may meet": only in due-stream mode)
  void launchUpdate(DevPlan& plan, int64_t begin, int64_t end, hipk::DataRef<BT> ref,
                    hipStream_t stream, BT* altTarget = nullptr, int64_t altStride = 0,
                    unsigned extraLds = 0, int atomicMask = 1) {
    if (altTarget == nullptr && end < (int64_t)plan.slowPrefix.size() &&
        plan.slowPrefix[end] == plan.slowPrefix[begin]) {
      // (32 KB of static LDS instead of updateTile's 34.8: 3 KB more padding keeps it at three
      //  workgroups per CU next to a chain workgroup)
      const unsigned pad = extraLds ? extraLds + 3072 : 0;
      // (cooperative CU yield, hip_kernels.h: side-stream launches of a single matrix only)
      const unsigned* yf = (extraLds && batchSize == 1) ? sym.yieldWord() : nullptr;
      hipk::updateTileBulk<BT><<<dim3((unsigned)(end - begin), (unsigned)batchSize), 256, pad,
                                stream>>>(plan.updTasksFat.as<UpdTaskFat>() + begin, ref, yf, atomicMask);
      return;
    }
    // (launches of at most ~2 rounds of workgroups: the latency-bound variant, hip_kernels.h)
    const int64_t wgs = (end - begin) * (int64_t)batchSize;
    const bool few = wgs <= sym.updPrefetchMaxWgs;
    auto kern = few ? hipk::updateTile<BT, true> : hipk::updateTile<BT, false>;
    // (Asking for enough dynamic LDS that only ceil(workgroups / CUs) of a small launch fit on a CU,
    //  in case the dispatcher packs them four to a CU: no effect at batch 1 / 8 / 64 -- it does not.)
    kern<<<dim3((unsigned)(end - begin), (unsigned)batchSize), 256, extraLds, stream>>>(
        plan.updTasksWide.as<UpdTaskWide>() + begin,
        plan.chainOffTab.as<int64_t>(), plan.rowChain.as<int32_t>(), plan.rowLocal.as<int32_t>(),
        plan.rowColOff.as<int32_t>(), ref, altTarget, altStride, atomicMask);
  }

  // One level = potrf -> trsm -> update on the execution stream.  Deferred (lookahead) tiles go to
  // the side stream after the level's trsm and are joined back by events where the plan says so.
  // (profiled runs serialise the launches on the execution stream, unless the in-situ mode asks
  //  for the real multi-stream schedule with every launch timed on the stream it runs on)
  bool lookaheadOn() const {
    return (sym.profile == nullptr || sym.profileInSitu) && sym.lookaheadEnabled;
  }

  void launchLevels(DevPlan& plan, const vector<LevelRange>& levels, hipk::DataRef<BT> ref,
                    LaunchTimer& timer) {
    const dim3 gy(1, (unsigned)batchSize, 1);
    // (side streams only when the lookahead units are worth their forks, HipPlanHost::lookaheadPays;
    //  otherwise the same launches go in line)
    const bool lookahead = lookaheadOn() && plan.host.lookaheadPays(batchSize, sym.lookaheadMinFlops);
    vector<hipEvent_t> defDone(levels.size(), nullptr);  // due units of a level complete
    vector<hipEvent_t> optDone(levels.size(), nullptr);  // ... its optional units (due-stream mode)
    // (fp32: its atomics cost more than the second stream returns -- BAL-871 5.86 against 5.22 ms,
    //  BAL-1723 20.5 against 18.8 -- so single-precision calls keep the one-side-stream order, and
    //  the tasks' "two streams may meet" bit is masked off)
    const bool dueStream = plan.host.opts.dueStream && sizeof(BT) == 8;
    const int sideMask = dueStream ? 3 : 1;
    // levels whose potrf is folded into their trsm launch in THIS call (DevPlan::foldRank): the first
    // nFold levels of the plan's order by row tiles
    int64_t nFold = 0;
    if (sym.potrfInTrsm && &levels == &plan.host.levels) {
      while (nFold < (int64_t)plan.foldTiles.size() &&
             plan.foldTiles[nFold] * (int64_t)batchSize <= sym.foldMaxTiles) {
        nFold++;
      }
    }
    // fork level of the same lump's previous block (-1: none)
    auto prevFork = [&](int64_t f) -> int64_t { return f >= 0 ? levels[f].waitDefLevel : -1; };
    auto waitDeferred = [&](int64_t f) {
      // everything the side streams owe to the column block this level is about to touch
      if (f < 0 || !defDone[f]) return;
      hipCHECK(hipStreamWaitEvent(sym.stream, defDone[f], 0));
      const int64_t pf = prevFork(f);
      if (dueStream && pf >= 0 && optDone[pf]) hipCHECK(hipStreamWaitEvent(sym.stream, optDone[pf], 0));
    };
    bool potrfFused = false;  // this level's potrf ran inside the previous level's update launch
    bool sideUsed = false, dueUsed = false;
    // inverted diagonal blocks of the chain panels: written by a panel's potrf, read by its trsm
    // (two slots per matrix, alternating from panel to panel)
    // (a batch factored as concurrent sub-batches, factorRange: every sub-batch its own slice)
    const int64_t scratchBatch = std::max<int64_t>(batchSize, subBatchTotal);
    sym.dinvScratch.resize((size_t)scratchBatch * hipk::kDinvBatchStride * sizeof(BT));
    BT* dinvBase = const_cast<BT*>(sym.dinvScratch.as<BT>()) + (size_t)subBatchBase * hipk::kDinvBatchStride;
    int dinvSlot = 0;  // slot of the current level's panel
    // staging buffer of the chain (chainStep): unsolved rows of the current / next panel
    const int64_t rawSlot = plan.host.maxChainRows * kTile;
    BT* rawBase = nullptr;
    if (rawSlot > 0) {
      sym.rawScratch.resize((size_t)scratchBatch * 2 * rawSlot * sizeof(BT));
      rawBase = const_cast<BT*>(sym.rawScratch.as<BT>()) + (size_t)subBatchBase * 2 * rawSlot;
    }
    bool rawValid = false;  // the previous level staged this level's panel rows
    // early tile-(0,0) updates of the next outer block (LevelRange::extraDiag): how many panels of
    // the current block have applied theirs, in order -- the block-last step's potrf workgroup
    // applies the rest from memory
    int extraApplied = 0;
    bool extraBroken = false;
    // gather overlap: chunks each stream has already waited for (events of one stream complete in order)
    const bool chunks = &levels == &plan.host.levels && !gatherChunkDone.empty();
    int waitedExec = 0, waitedDue = 0, waitedSide = 0;
    hipEvent_t lastOpt = nullptr;  // the most recent optional lookahead launch (side stream)
    auto waitChunk = [&](hipStream_t st, int& waited, int32_t chunk) {
      if (!chunks || chunk <= waited || chunk >= (int32_t)gatherChunkDone.size()) return;
      hipCHECK(hipStreamWaitEvent(st, gatherChunkDone[chunk], 0));
      waited = chunk;
    };
    for (size_t li = 0; li < levels.size(); li++) {
      const LevelRange& lr = levels[li];
      const unsigned nP = (unsigned)(lr.panelEnd - lr.panelBegin);
      const unsigned nT = (unsigned)(lr.trsmEnd - lr.trsmBegin);
      const bool direct = lr.directPanel >= 0;
      waitChunk(sym.stream, waitedExec, lr.gatherNow);
      if (lr.tail) {
        if (lr.tail == 1) {
          // PERSISTENT TAIL (hip_tail_kernel.h): everything the lookahead streams still owe to these
          // columns first, then one launch for all the panels of the tail
          auto joinStream = [&](hipStream_t st) {
            hipEvent_t e = sym.eventFromPool();
            hipCHECK(hipEventRecord(e, st));
            hipCHECK(hipStreamWaitEvent(sym.stream, e, 0));
          };
          if (sideUsed) joinStream(sym.sideStream());
          if (dueUsed) joinStream(sym.dueSideStream());
          sideUsed = dueUsed = false;
          const PanelDesc& p0 = plan.host.panels[lr.directPanel];
          hipk::TailDesc td;
          td.diagOff = p0.diagOff;
          td.lda = p0.lda;
          td.K = p0.nb + p0.nRest;
          td.nP = (td.K + kTile - 1) / kTile;
          td.ctlStride = hipk::tailCtlWords(td.nP);
          td.flags = sym.tailFlags;
          td.pad = 0;
          BASPACHO_CHECK_EQ(td.nP, lr.tailPanels);
          // control words and exchange slots of every matrix in one buffer, armed by one memset
          const size_t ctlBytes = ((size_t)td.ctlStride * scratchBatch * sizeof(unsigned) + 255) / 256 * 256;
          const size_t xchPerMat = (size_t)td.nP * kTile * kTile * sizeof(BT);
          sym.tailCtl.resize(ctlBytes + xchPerMat * scratchBatch);
          sym.tailDinv.resize((size_t)scratchBatch * td.nP * hipk::kDinvSlot * sizeof(BT));
          unsigned* ctl = reinterpret_cast<unsigned*>(sym.tailCtl.ptr) + (size_t)subBatchBase * td.ctlStride;
          BT* xch = reinterpret_cast<BT*>(reinterpret_cast<char*>(sym.tailCtl.ptr) + ctlBytes + xchPerMat * subBatchBase);
          BT* tdinv = const_cast<BT*>(sym.tailDinv.as<BT>()) + (size_t)subBatchBase * td.nP * hipk::kDinvSlot;
          if (subBatchTotal > 0) {  // (a sub-batch arms its own slices only)
            hipCHECK(hipMemsetAsync(ctl, 0xff, (size_t)td.ctlStride * batchSize * sizeof(unsigned), sym.stream));
            hipCHECK(hipMemsetAsync(xch, 0xff, xchPerMat * batchSize, sym.stream));
          } else {
            hipCHECK(hipMemsetAsync(sym.tailCtl.ptr, 0xff, ctlBytes + xchPerMat * scratchBatch, sym.stream));
          }
          timer.begin(kProfChainUpdate);
          hipk::tailFactor<BT><<<dim3((unsigned)hipk::tailRoles(td.nP), gy.y), 256, 0, sym.stream>>>(
              td, ref, tdinv, ctl, xch, sym.asyncErrWord(), (long long)(sym.sweepSpinLimitS * 1e8), sym.tailTrace());
          timer.end();
          sym.counters.tailLaunches++;
        }
        rawValid = false;
        potrfFused = false;
        continue;
      }
      const int slot = dinvSlot;
      dinvSlot ^= 1;
      BT* dinvCur = dinvBase + slot * hipk::kDinvSlot;
      BT* dinvNext = dinvBase + (slot ^ 1) * hipk::kDinvSlot;
      BT* rawCur = rawBase ? rawBase + slot * rawSlot : nullptr;
      BT* rawNext = rawBase ? rawBase + (slot ^ 1) * rawSlot : nullptr;
      const bool fold = nFold > 0 && !potrfFused && plan.foldRank[li] >= 0 && plan.foldRank[li] < nFold;
      if (potrfFused) {
        potrfFused = false;
      } else if (fold) {
        sym.counters.potrfFoldedLevels++;  // (trsmPanelPotrf below; the factors are stored after the last level)
      } else if (nP) {
        timer.begin(kProfPotrf);
        if (direct) {
          hipk::potrfPanelDirect<BT><<<dim3(1, gy.y), 256, 0, sym.stream>>>(
              plan.host.panels[lr.directPanel], ref, dinvCur);
        } else {
          hipk::potrfPanel<BT><<<dim3(nP, gy.y), 256, 0, sym.stream>>>(
              plan.levelPanelDescs.as<PanelDesc>() + lr.panelBegin, ref);
        }
        timer.end();
      }
      // (decided before the trsm launch: with splitK one extra workgroup of that launch already
      //  touches tile 0 of the block-wide segment)
      const bool fuse = direct && lr.directSeg >= 0 && lr.fuseNext && li + 1 < levels.size() &&
                        levels[li + 1].directPanel >= 0 && lr.updEnd > lr.updBegin;
      int splitK = (fuse && nT) ? lr.splitK : 0;
      const bool directUpd = direct && lr.directSeg >= 0 && lr.updEnd > lr.updBegin;
      // does this level's update stage the next panel's rows?
      const bool stage = rawBase && directUpd && lr.rawNext && li + 1 < levels.size() &&
                         levels[li + 1].directPanel >= 0;
      const PanelDesc nextPanel = (stage || fuse) ? plan.host.panels[levels[li + 1].directPanel]
                                                  : PanelDesc{};
      // one launch for trsm + update (+ next potrf): intra-block step whose rows were staged
      // (inside an outer block: K = nb, every source column is solved in the launch; block-last
      //  step: the block-wide source, its leading K - nb columns are final in memory)
      bool merged = false;
      int64_t memOff = 0;
      int kMem = 0;
      if (rawValid && directUpd && nT) {
        const SegDesc& sd = plan.host.segs[lr.directSeg];
        const SrcDesc& sr = plan.host.srcs[sd.src];
        const PanelDesc& pdc = plan.host.panels[lr.directPanel];
        if (!sd.outer && sr.K == pdc.nb) {
          merged = true;
        } else if (sd.outer == 1 && sr.K > pdc.nb && (sr.K - pdc.nb) % kTile == 0 &&
                   sr.rowsBelow == pdc.rowsBelow && sr.lda == pdc.lda) {
          merged = true;
          memOff = sr.off;
          kMem = sr.K - pdc.nb;
        }
      }
      if (merged) splitK = 0;
      bool waitedDef = false;
      if (splitK && lookahead && lr.waitDefLevel >= 0 && defDone[lr.waitDefLevel]) {
        waitDeferred(lr.waitDefLevel);
        waitedDef = true;
      }
      if (nT && !merged) {
        timer.begin(kProfTrsm);
        if (splitK) {
          const SegDesc& sd = plan.host.segs[lr.directSeg];
          SrcDesc part = plan.host.srcs[sd.src];
          part.K = splitK;
          hipk::trsmPanelDirectPlus<BT><<<dim3(nT + 1, gy.y), 256, 0, sym.stream>>>(
              plan.host.panels[lr.directPanel], part, sd, ref, dinvCur);
        } else if (direct) {
          hipk::trsmPanelDirect<BT><<<dim3(nT, gy.y), 256, 0, sym.stream>>>(
              plan.host.panels[lr.directPanel], ref, dinvCur);
        } else if (fold) {
          hipk::trsmPanelPotrf<BT><<<dim3(nT, gy.y), 256, 0, sym.stream>>>(
              plan.trsmTasksFat.as<TrsmTaskFat>() + lr.trsmBegin, ref);
        } else {
          hipk::trsmPanel<BT><<<dim3(nT, gy.y), 256, 0, sym.stream>>>(
              plan.trsmTasksFat.as<TrsmTaskFat>() + lr.trsmBegin, ref);
        }
        timer.end();
      }
      // (a second wait on the same event would still cost a ~6 us bubble between the launches)
      if (!waitedDef && lookahead && lr.waitDefLevel >= 0 && defDone[lr.waitDefLevel]) {
        waitDeferred(lr.waitDefLevel);
      }
      const bool anyDef = lr.defEnd > lr.defBegin;
      auto forkSide = [&]() {
        // (under the source-ordered schedule the fork came AFTER the level's own update launch,
        //  whose "now" tiles ran 3x faster alone than beside freshly started bulk tiles; with the
        //  deadline-ordered units the side stream is busy either way and the earlier fork wins)
        hipEvent_t fork = sym.eventFromPool();
        hipCHECK(hipEventRecord(fork, sym.stream));
        hipCHECK(hipStreamWaitEvent(sym.sideStream(), fork, 0));
        sym.counters.lookaheadForks++;
        // first the tiles the next block's own update must wait for, then (event) the rest:
        // the side stream keeps running them while the chain goes on, and the next block's
        // deferred tiles queue up right behind
        if (dueStream) {
          // due units on their own stream, beside the optional ones (both accumulate with atomics
          // where they can meet: hip_plan.cpp, pushUnit).  They must not overtake the optional
          // units forked two blocks ago, whose far columns are plain read-modify-write.
          hipStream_t due = sym.dueSideStream();
          hipCHECK(hipStreamWaitEvent(due, fork, 0));
          waitChunk(due, waitedDue, lr.gatherDue);
          waitChunk(sym.sideStream(), waitedSide, lr.gatherOpt);
          if (lr.flushDue && lastOpt) hipCHECK(hipStreamWaitEvent(due, lastOpt, 0));
          const int64_t f2 = lr.optWaitLevel;
          if (f2 >= 0 && optDone[f2]) hipCHECK(hipStreamWaitEvent(due, optDone[f2], 0));
          if (lr.defMid > lr.defBegin) {
            timer.begin(kProfUpdate, due);
            launchUpdate(plan, lr.defBegin, lr.defMid, ref, due, nullptr, 0, sym.dueExtraLds, sideMask);
            timer.end();
          }
          defDone[li] = sym.eventFromPool();
          hipCHECK(hipEventRecord(defDone[li], due));
          if (lr.defEnd > lr.defMid) {
            timer.begin(kProfUpdate, sym.sideStream());
            launchUpdate(plan, lr.defMid, lr.defEnd, ref, sym.sideStream(), nullptr, 0, sym.bulkExtraLds, sideMask);
            timer.end();
            optDone[li] = sym.eventFromPool();
            hipCHECK(hipEventRecord(optDone[li], sym.sideStream()));
            lastOpt = optDone[li];
          }
          sideUsed = dueUsed = true;
          return;
        }
        waitChunk(sym.sideStream(), waitedSide, std::max(lr.gatherDue, lr.gatherOpt));
        if (lr.defMid > lr.defBegin) {
          timer.begin(kProfUpdate, sym.sideStream());
          launchUpdate(plan, lr.defBegin, lr.defMid, ref, sym.sideStream(), nullptr, 0, sym.bulkExtraLds, sideMask);
          timer.end();
        }
        defDone[li] = sym.eventFromPool();
        hipCHECK(hipEventRecord(defDone[li], sym.sideStream()));
        if (lr.defEnd > lr.defMid) {
          timer.begin(kProfUpdate, sym.sideStream());
          launchUpdate(plan, lr.defMid, lr.defEnd, ref, sym.sideStream(), nullptr, 0, sym.bulkExtraLds, sideMask);
          timer.end();
        }
        sideUsed = true;
      };
      // (the lookahead units read this level's panel columns and write columns the level's own
      //  update does not touch: they can be forked before it)
      const bool forkEarly = !merged;  // (merged: the panel is solved in the launch)
      if (lookahead && anyDef && forkEarly) forkSide();
      const int64_t updBegin = lr.updBegin;
      if (lr.updEnd > updBegin) {
        timer.begin(direct && lr.directSeg >= 0 ? kProfChainUpdate : kProfUpdate);
        const unsigned nUpd = (unsigned)(lr.updEnd - updBegin);
        if (merged) {
          int kMem0 = kMem, extra = 0;
          if (kMem == 0) {  // intra-block step
            kMem0 = 0;
            if (lr.extraDiag && !extraBroken && fuse) {
              extra = 1;
              extraApplied++;
            } else {
              extraBroken = true;
            }
          } else {          // block-last step: the panels that applied theirs are done
            kMem0 = std::max(0, kMem - kTile * extraApplied);
            extraApplied = 0;
            extraBroken = false;
          }
          hipk::chainStep<BT><<<dim3(nUpd + extra, gy.y), 256, 0, sym.stream>>>(
              plan.host.panels[lr.directPanel], plan.host.segs[lr.directSeg], (int)nUpd, nextPanel,
              fuse ? 1 : 0, ref, rawCur, stage ? rawNext : nullptr, 2 * rawSlot, dinvCur, dinvNext,
              memOff, kMem, (lookahead && batchSize == 1) ? sym.yieldWord() : nullptr,
              sym.traceLaunchId++, kMem0, extra);
          potrfFused = fuse;
        } else if (fuse) {
          if (extraApplied > 0) {
            throw std::runtime_error("HIP backend: internal error (early diagonal updates applied "
                                     "before a step that is not merged)");
          }
          extraBroken = true;
          const SegDesc& sd = plan.host.segs[lr.directSeg];
          hipk::updateTileDirectPotrf<BT><<<dim3(nUpd, gy.y), 256, 0, sym.stream>>>(
              plan.host.srcs[sd.src], sd, (int)nUpd, nextPanel, ref, splitK, dinvNext,
              stage ? rawNext : nullptr, 2 * rawSlot);
          potrfFused = true;
        } else if (direct && lr.directSeg >= 0) {
          if (extraApplied > 0) {
            throw std::runtime_error("HIP backend: internal error (early diagonal updates applied "
                                     "before a step that is not merged)");
          }
          extraBroken = true;
          const SegDesc& sd = plan.host.segs[lr.directSeg];
          hipk::updateTileDirect<BT><<<dim3(nUpd, gy.y), 256, 0, sym.stream>>>(
              plan.host.srcs[sd.src], sd, (int)nUpd, ref, stage ? rawNext : nullptr, nextPanel.nb,
              2 * rawSlot);
        } else {
          auto sp = sym.splitK ? plan.splitRange.find(updBegin) : plan.splitRange.end();
          if (sp != plan.splitRange.end() &&
              (sp->second.second - sp->second.first) * (int64_t)batchSize <= sym.splitKMaxWgs) {
            // (one round of workgroups at most: the K slices of a tile run side by side)
            sym.counters.splitListsUsed++;
            hipk::updateTile<BT, true><<<dim3((unsigned)(sp->second.second - sp->second.first),
                                              (unsigned)batchSize), 256, 0, sym.stream>>>(
                plan.updTasksSplit.as<UpdTaskWide>() + sp->second.first, plan.chainOffTab.as<int64_t>(),
                plan.rowChain.as<int32_t>(), plan.rowLocal.as<int32_t>(), plan.rowColOff.as<int32_t>(), ref,
                nullptr, 0, 1);
          } else {
            launchUpdate(plan, updBegin, lr.updEnd, ref, sym.stream);
          }
        }
        timer.end();
      }
      rawValid = stage;
      if (lookahead && anyDef && !forkEarly) forkSide();
      if (!lookahead) {  // (the same two launches as the lookahead schedule, on the main stream)
        if (lr.defMid > lr.defBegin) {
          timer.begin(kProfUpdate);
          launchUpdate(plan, lr.defBegin, lr.defMid, ref, sym.stream);
          timer.end();
        }
        if (lr.defEnd > lr.defMid) {
          timer.begin(kProfUpdate);
          launchUpdate(plan, lr.defMid, lr.defEnd, ref, sym.stream);
          timer.end();
        }
      }
    }
    if (nFold > 0) {
      // the diagonal blocks of the folded levels: factored (again) and stored, all in one launch --
      // nothing in the factorisation reads them, the solves do
      timer.begin(kProfPotrf);
      hipk::potrfPanel<BT><<<dim3((unsigned)plan.foldPanelEnd[nFold - 1], gy.y), 256, 0, sym.stream>>>(
          plan.foldDescs.as<PanelDesc>(), ref);
      timer.end();
    }
    if (sideUsed) {  // join: everything on the side stream(s) happens-before what follows
      hipEvent_t join = sym.eventFromPool();
      hipCHECK(hipEventRecord(join, sym.sideStream()));
      hipCHECK(hipStreamWaitEvent(sym.stream, join, 0));
    }
    if (dueUsed) {
      hipEvent_t join = sym.eventFromPool();
      hipCHECK(hipEventRecord(join, sym.dueSideStream()));
      hipCHECK(hipStreamWaitEvent(sym.stream, join, 0));
    }
    if (chunks) {  // (the last level waited for the last chunk already; a plan without levels did not)
      waitChunk(sym.stream, waitedExec, (int32_t)gatherChunkDone.size() - 1);
      gatherChunkDone.clear();
    }
  }

  void launchElim(DevPlan& plan, const ElimRangePlan& er, hipk::DataRef<BT> ref,
                  LaunchTimer& timer) {
    hipk::SkelDev sk = sym.skelDev();
    const unsigned gy = (unsigned)batchSize;
    const int64_t nLumps = er.lumpEnd - er.lumpBegin;
    if (nLumps <= 0) return;
    const unsigned gF = (unsigned)((nLumps + 3) / 4);
    timer.begin(kProfElimFactor);
    if (er.maxWidth <= 4) {
      const int64_t perWg = 4 * hipk::kTinyPerWave;
      hipk::elimFactorTinyStaged<BT><<<dim3((unsigned)((nLumps + perWg - 1) / perWg), gy), 256, 0,
                                      sym.stream>>>(
          plan.elimLumpDesc.as<ElimLumpDesc>() + er.descBegin, ref, (int)nLumps);
    } else if (er.maxWidth <= 8) {
      hipk::elimFactorSmall<BT, 8><<<dim3(gF, gy), 256, 0, sym.stream>>>(sk, ref, er.lumpBegin,
                                                                        er.lumpEnd);
    } else {
      hipk::elimFactorSmall<BT, kElimSmallMax><<<dim3(gF, gy), 256, 0, sym.stream>>>(
          sk, ref, er.lumpBegin, er.lumpEnd);
    }
    timer.end();
    launchLevels(plan, er.bigLevels, ref, timer);
    const int64_t nChains = er.chainEnd - er.chainBegin;
    // FAULT INJECTION (bsp_test_set_fault, tests only): the whole sparse-elimination update is
    // dropped -- the factor of everything the eliminated columns touch is then wrong, and the
    // full-size parity tests must notice (tests/test_full_size_gpu.py)
    if (sym.faultDropElimUpdate) {
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true)) {
        fprintf(stderr, "baspacho_amd: FAULT INJECTION ACTIVE -- sparse-elimination update dropped (tests only)\n");
      }
      return;
    }
    if (er.useGather) {
      const int64_t nItems = er.itemEnd - er.itemBegin;
      // (extraLds: overlapped chunks may be throttled to fewer workgroups per CU, BSP_GATHER_OVERLAP_LDS)
      auto gatherRange = [&](int64_t b, int64_t e, hipStream_t st, unsigned extraLds = 0) {
        if (e <= b) return;
        timer.begin(kProfElimUpdate, st);
        hipk::elimGatherMfma<BT><<<dim3((unsigned)((e - b + 3) / 4), gy), 256, extraLds, st>>>(
            plan.elimItems.as<ElimGatherItem>() + b, plan.elimPairOffJ.as<uint32_t>(),
            plan.elimPairOffI.as<uint32_t>(), ref, (int)(e - b));
        timer.end();
      };
      gatherChunkDone.clear();
      if (!er.chunkItemPtr.empty()) {
        // GATHER OVERLAP (hip_plan.h, ElimRangePlan::chunkItemPtr): chunk 0 here, the others on a
        // stream of their own beside the dense chain, one event per chunk; launchLevels makes every
        // dense launch wait for the chunk of the last column block it touches
        const size_t nChunks = er.chunkItemPtr.size() - 1;
        const bool overlap = er.overlapLump >= 0 && sym.gatherOverlap && lookaheadOn() &&
                             plan.host.lookaheadPays(batchSize, sym.lookaheadMinFlops);
        if (overlap) {
          hipStream_t gs = sym.streams().batch[0];
          hipEvent_t factored = sym.eventFromPool();
          hipCHECK(hipEventRecord(factored, sym.stream));
          hipCHECK(hipStreamWaitEvent(gs, factored, 0));
          gatherRange(er.chunkItemPtr[0], er.chunkItemPtr[1], sym.stream);
          gatherChunkDone.assign(nChunks, nullptr);
          for (size_t c = 1; c < nChunks; c++) {
            gatherRange(er.chunkItemPtr[c], er.chunkItemPtr[c + 1], gs, sym.gatherOverlapLds);
            gatherChunkDone[c] = sym.eventFromPool();
            hipCHECK(hipEventRecord(gatherChunkDone[c], gs));
          }
          sym.counters.gatherChunksOverlapped += (int64_t)nChunks - 1;
        } else {
          for (size_t c = 0; c < nChunks; c++) gatherRange(er.chunkItemPtr[c], er.chunkItemPtr[c + 1], sym.stream);
        }
      } else {
        gatherRange(er.itemBegin, er.itemBegin + nItems, sym.stream);
      }
      const int64_t nWide = er.ldsEnd - er.ldsBegin;
      if (nWide > 0) {
        timer.begin(kProfElimUpdate);
        hipk::elimGather<BT><<<dim3((unsigned)((nWide + 3) / 4), gy), 256, 0, sym.stream>>>(
            plan.elimItems.as<ElimGatherItem>() + er.ldsBegin, plan.elimPairOffJ.as<uint32_t>(),
            plan.elimPairOffI.as<uint32_t>(), ref, (int)nWide);
        timer.end();
      }
      const int64_t nTiny9 = er.tiny9End - er.tinyBegin, nTiny = er.tinyEnd - er.tiny9End;
      if (nTiny9 > 0) {  // 7 items per wave, 28 per workgroup
        timer.begin(kProfElimUpdate);
        hipk::elimGatherTiny<BT, 9><<<dim3((unsigned)((nTiny9 + 27) / 28), gy), 256, 0, sym.stream>>>(
            plan.elimItems.as<ElimGatherItem>() + er.tinyBegin, plan.elimPairOffJ.as<uint32_t>(),
            plan.elimPairOffI.as<uint32_t>(), ref, (int)nTiny9);
        timer.end();
      }
      if (nTiny > 0) {
        timer.begin(kProfElimUpdate);
        hipk::elimGatherTiny<BT, 16><<<dim3((unsigned)((nTiny + 15) / 16), gy), 256, 0, sym.stream>>>(
            plan.elimItems.as<ElimGatherItem>() + er.tiny9End, plan.elimPairOffJ.as<uint32_t>(),
            plan.elimPairOffI.as<uint32_t>(), ref, (int)nTiny);
        timer.end();
      }
    } else if (nChains > 0) {
      timer.begin(kProfElimUpdate);
      hipk::elimUpdate<BT><<<dim3((unsigned)((nChains + 3) / 4), gy), 256, 0, sym.stream>>>(
          sk, plan.elimChainLump.as<int32_t>() + er.chainLumpOff, ref, er.chainBegin, er.chainEnd);
      timer.end();
    }
  }

  virtual bool hasFusedFactor() const override { return !sym.forcePerOp; }

  // every launch of a factor over `plan`, in order, on sym.stream and the auxiliary streams
  void enqueueFactor(DevPlan& plan, hipk::DataRef<BT> ref, LaunchTimer& timer) {
    sym.resetEventPool();
    // CONCURRENT SUB-BATCHES (round 5).  A large batch of a latency-bound structure is three dependent
    // launches per tree level, each with a ramp and a tail in which most of the GPU idles; two halves
    // of the batch on two streams fill each other's (64 x GRID 82x82: 10.76 -> 10.2 ms with two Solver
    // clones on two streams, profiles/r3_concurrent_batch.py re-run in round 5; four halves: host
    // enqueue bound, slower).  The second half runs on the auxiliary stream, which such a plan does
    // not use for lookahead; matrices are independent, the halves share nothing but the plan.
    const bool lookahead = lookaheadOn() && plan.host.lookaheadPays(batchSize, sym.lookaheadMinFlops);
    if (ref.many && !lookahead && sym.profile == nullptr && batchSize >= sym.subBatchMin) {
      const int total = batchSize;
      const int parts = std::max(2, std::min({sym.subBatchParts, 4, total / std::max(1, sym.subBatchMin / 2)}));
      hipStream_t mainStream = sym.stream;
      hipEvent_t fork = sym.eventFromPool();
      hipCHECK(hipEventRecord(fork, mainStream));  // (the pointer array has been copied on this stream)
      struct Restore {
        HipNumericCtx& c;
        hipStream_t st;
        int bs;
        ~Restore() {
          c.sym.stream = st;
          c.batchSize = bs;
          c.subBatchBase = c.subBatchTotal = 0;
        }
      } restore{*this, mainStream, total};
      subBatchTotal = total;
      for (int g = 0; g < parts; g++) {
        const int b0 = (int)((int64_t)total * g / parts), b1 = (int)((int64_t)total * (g + 1) / parts);
        subBatchBase = b0;
        batchSize = b1 - b0;
        sym.stream = g ? sym.streams().batch[g - 1] : mainStream;
        if (g) hipCHECK(hipStreamWaitEvent(sym.stream, fork, 0));
        sym.counters.subBatchesEnqueued++;
        hipk::DataRef<BT> sub{nullptr, ref.many + subBatchBase};
        for (const ElimRangePlan& er : plan.host.elimRanges) launchElim(plan, er, sub, timer);
        launchLevels(plan, plan.host.levels, sub, timer);
        if (g) {
          hipEvent_t join = sym.eventFromPool();
          hipCHECK(hipEventRecord(join, sym.stream));
          hipCHECK(hipStreamWaitEvent(mainStream, join, 0));
        }
      }
      return;
    }
    for (const ElimRangePlan& er : plan.host.elimRanges) launchElim(plan, er, ref, timer);
    launchLevels(plan, plan.host.levels, ref, timer);
  }

  // (factor() as ONE captured hipGraph launch was built in round 2 and measured no faster on this
  //  stack -- GRID 82x82 1.66-1.79 against 1.74-1.77 ms, BAL-871 7.15 = 7.16: the runtime plays a
  //  graph back as the same packets on the same queues -- and removed in round 4)
  virtual void factorRange(T* data, int64_t startLump, int64_t upToLump) override {
    sym.checkAsyncError();
    DevPlan* planPtr = &sym.planFor(sym.sparseElimRanges, startLump, upToLump, /*tag=*/0);
    if (batchSize > 1 && planPtr->host.hasTail && !sym.tailInBatch) {
      planPtr = &sym.planFor(sym.sparseElimRanges, startLump, upToLump, /*tag=*/2);
    }
    DevPlan& plan = *planPtr;
    hipk::DataRef<BT> ref = makeRef(data);
    LaunchTimer timer(sym.stream, sym.profile);
    enqueueFactor(plan, ref, timer);
    hipCHECK(hipGetLastError());
    timer.finish();
  }

  // host array of the matrices' device pointers (one entry for a single matrix)
  void dataPointers(T* data, vector<BT*>& out);

  virtual void doElimination(const SymElimCtx& elimData, T* data, int64_t lumpsBegin,
                             int64_t lumpsEnd) override {
    const HipSymElimCtx* e = dynamic_cast<const HipSymElimCtx*>(&elimData);
    BASPACHO_CHECK_NOTNULL(e);
    BASPACHO_CHECK_EQ(e->lumpsBegin, lumpsBegin);
    BASPACHO_CHECK_EQ(e->lumpsEnd, lumpsEnd);
    DevPlan& plan = sym.planFor({lumpsBegin, lumpsEnd}, lumpsBegin, lumpsEnd, /*tag=*/1);
    hipk::DataRef<BT> ref = makeRef(data);
    LaunchTimer timer(sym.stream, sym.profile);
    sym.resetEventPool();
    for (const ElimRangePlan& er : plan.host.elimRanges) launchElim(plan, er, ref, timer);
    hipCHECK(hipGetLastError());
    timer.finish();
  }

  // ---- per-op boundary (MatOps.h:113-136).  factor() never goes through these (fused path); they
  // exist so that a driver written against the reference's NumericCtx -- e.g. the per-op loop of
  // Solver.cpp:198-218, kept in solver.cpp -- runs on this backend op by op.  Each call builds a
  // one-off device plan: correct, not fast.
  DevPlan& adoptPlan(HipPlanHost&& host) {
    opPlans.emplace_back(new DevPlan);
    opPlans.back()->host = std::move(host);
    opPlans.back()->upload();
    return *opPlans.back();
  }

  virtual void potrf(int64_t n, T* data, int64_t offA) override {
    sym.potrfBiggestN = std::max(sym.potrfBiggestN, n);
    DevPlan& plan = adoptPlan(buildDenseOpPlan(n, 0, offA, /*potrfOnly=*/true));
    OpTimer opTimer(sym.potrfStat, sym.stream, (double)n);
    LaunchTimer timer(sym.stream, nullptr);
    {
      FlagOff noLookahead(sym.lookaheadEnabled);
      launchLevels(plan, plan.host.levels, makeRef(data), timer);
    }
    hipCHECK(hipGetLastError());
  }

  virtual void trsm(int64_t n, int64_t k, T* data, int64_t offA, int64_t offB) override {
    // every caller on the factor path passes the rows that directly follow the diagonal block
    // (Solver.cpp:43-64); the blocked solve relies on that contiguity
    BASPACHO_CHECK_EQ(offB, offA + n * n);
    DevPlan& plan = adoptPlan(buildDenseOpPlan(n, k, offA, /*potrfOnly=*/false));
    OpTimer opTimer(sym.trsmStat, sym.stream, (double)n, (double)k);
    LaunchTimer timer(sym.stream, nullptr);
    {
      FlagOff noLookahead(sym.lookaheadEnabled);
      launchLevels(plan, plan.host.levels, makeRef(data), timer);
    }
    hipCHECK(hipGetLastError());
  }

  virtual void saveSyrkGemm(int64_t m, int64_t n, int64_t k, const T* data, int64_t offset) override {
    BASPACHO_CHECK_LE(m * n, tempBufSize);
    temp.resize((size_t)(tempBufSize * batchSize) * sizeof(BT));
    hipCHECK(hipMemsetAsync(temp.ptr, 0, (size_t)(tempBufSize * batchSize) * sizeof(BT), sym.stream));
    HipPlanHost host;
    SrcDesc sr{};
    sr.off = offset;
    sr.lda = (int32_t)k;
    sr.K = (int32_t)k;
    sr.rowsBelow = (int32_t)n;
    sr.nRest = (int32_t)n;
    host.srcs.push_back(sr);
    SegDesc sd{};
    sd.src = 0;
    sd.kind = kSegIntra;
    sd.q0 = 0;
    sd.m = (int32_t)m;
    sd.tgtBase = 0;
    sd.tgtStride = (int32_t)m;
    host.segs.push_back(sd);
    for (int32_t cT = 0; cT < m; cT += kTile) {
      for (int32_t rT = cT; rT < n; rT += kTile) host.updTasks.push_back({0, rT, cT, 0});
    }
    DevPlan& plan = adoptPlan(std::move(host));
    sym.gemmCalls++;
    OpTimer opTimer(sym.sygeStat, sym.stream, (double)m, (double)n, (double)k);
    // temp := -(P P^T) on the lower trapezoid (the strictly upper part of the leading m x m block
    // "doesn't matter", MatOps.h:129); assemble() adds it
    launchUpdate(plan, 0, plan.numUpdTasks, makeRef(const_cast<T*>(data)),
                 sym.stream, temp.as<BT>() ? const_cast<BT*>(temp.as<BT>()) : nullptr, tempBufSize);
    hipCHECK(hipGetLastError());
  }

  virtual void prepareAssemble(int64_t targetLump) override {
    hipk::SkelDev sk = sym.skelDev();
    spanToChainOffset.resize((size_t)(sym.skel.numSpans() + 1) * sizeof(int64_t));
    const int64_t n = sym.skel.chainColPtr[targetLump + 1] - sym.skel.chainColPtr[targetLump];
    hipk::prepareAssembleKernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, sym.stream>>>(
        sk, reinterpret_cast<int64_t*>(spanToChainOffset.ptr), targetLump);
  }

  virtual void assemble(T* data, int64_t rectRowBegin, int64_t dstStride, int64_t srcColDataOffset,
                        int64_t srcRectWidth, int64_t numBlockRows, int64_t numBlockCols) override {
    hipk::SkelDev sk = sym.skelDev();
    OpTimer opTimer(sym.asmblStat, sym.stream, (double)numBlockRows, (double)numBlockCols);
    hipk::assembleKernel<BT><<<dim3((unsigned)numBlockRows, (unsigned)batchSize), 256, 0, sym.stream>>>(
        sk, spanToChainOffset.as<int64_t>(), temp.as<BT>(), tempBufSize, makeRef(data),
        rectRowBegin, dstStride, srcColDataOffset, srcRectWidth, numBlockRows, numBlockCols);
    hipCHECK(hipGetLastError());
  }

  virtual void pseudoFactorSpans(T* data, int64_t spanBegin, int64_t spanEnd) override {
    if (spanEnd <= spanBegin) return;
    const CoalescedBlockMatrixSkel& sk = sym.skel;
    // spans of up to kElimSmallMax columns: one wave per span (diagonal block in LDS, rows in
    // registers).  Wider spans (the reference has no width limit, MatOpsCuda.cu:188-233) go through
    // the panel potrf / trsm kernels, one dense-op plan per span; runs of narrow spans in between
    // are batched into one launch each.
    hipk::DataRef<BT> ref = makeRef(data);
    auto narrowRun = [&](int64_t b, int64_t e) {
      if (e <= b) return;
      hipk::pseudoFactorSpansKernel<BT><<<dim3((unsigned)((e - b + 3) / 4), (unsigned)batchSize), 256,
                                          0, sym.stream>>>(sym.skelDev(), ref, b, e);
    };
    FlagOff noLookahead(sym.lookaheadEnabled);
    int64_t runBegin = spanBegin;
    for (int64_t s = spanBegin; s < spanEnd; s++) {
      const int64_t n = sk.spanStart[s + 1] - sk.spanStart[s];
      if (n <= kElimSmallMax) continue;
      narrowRun(runBegin, s);
      runBegin = s + 1;
      const int64_t lump = sk.spanToLump[s];
      const int64_t lda = sk.lumpStart[lump + 1] - sk.lumpStart[lump];
      const int64_t c0 = sk.chainColPtr[lump], nCh = sk.chainColPtr[lump + 1] - c0;
      const int64_t idxInLump = s - sk.lumpToSpan[lump];
      // (the chains of a lump column follow each other: the rows below the span's diagonal block
      //  start n rows after it)
      const int64_t offD = sk.chainData[c0 + idxInLump] + sk.spanOffsetInLump[s];
      const int64_t rowsBelow = sk.chainRowsTillEnd[c0 + nCh - 1] - sk.chainRowsTillEnd[c0 + idxInLump];
      LaunchTimer timer(sym.stream, nullptr);
      DevPlan& pf = adoptPlan(buildDenseOpPlan(n, 0, offD, /*potrfOnly=*/true, 0, lda));
      launchLevels(pf, pf.host.levels, ref, timer);
      if (rowsBelow > 0) {
        DevPlan& ps = adoptPlan(buildDenseOpPlan(n, rowsBelow, offD, /*potrfOnly=*/false, 0, lda));
        launchLevels(ps, ps.host.levels, ref, timer);
      }
    }
    narrowRun(runBegin, spanEnd);
    hipCHECK(hipGetLastError());
  }

  HipSymbolicCtx& sym;
  int batchSize;
  int subBatchBase = 0, subBatchTotal = 0;  // enqueueFactor: the sub-batch being enqueued
  // gather overlap: events of the gather chunks of THIS factor call that run beside the dense chain
  // (launchElim -> launchLevels; [0] is null: chunk 0 runs on the execution stream)
  vector<hipEvent_t> gatherChunkDone;
  int64_t tempBufSize = 0;
  DevBuf temp, spanToChainOffset;  // per-op boundary only (saveSyrkGemm / prepareAssemble)
  vector<std::unique_ptr<DevPlan>> opPlans;
};

template <>
hipk::DataRef<double> HipNumericCtx<double>::makeRef(double* data) {
  return {data, nullptr};
}
template <>
hipk::DataRef<float> HipNumericCtx<float>::makeRef(float* data) {
  return {data, nullptr};
}
template <>
void HipNumericCtx<double>::dataPointers(double* data, vector<double*>& out) { out = {data}; }
template <>
void HipNumericCtx<float>::dataPointers(float* data, vector<float*>& out) { out = {data}; }
template <>
void HipNumericCtx<vector<double*>>::dataPointers(vector<double*>* data, vector<double*>& out) {
  BASPACHO_CHECK_EQ((int)data->size(), batchSize);
  out = *data;
}
template <>
void HipNumericCtx<vector<float*>>::dataPointers(vector<float*>* data, vector<float*>& out) {
  BASPACHO_CHECK_EQ((int)data->size(), batchSize);
  out = *data;
}
template <>
hipk::DataRef<double> HipNumericCtx<vector<double*>>::makeRef(vector<double*>* data) {
  BASPACHO_CHECK_EQ((int)data->size(), batchSize);
  return {nullptr, (double* const*)sym.ptrRing.push(data->data(), data->size() * sizeof(double*),
                                                    sym.stream)};
}
template <>
hipk::DataRef<float> HipNumericCtx<vector<float*>>::makeRef(vector<float*>* data) {
  BASPACHO_CHECK_EQ((int)data->size(), batchSize);
  return {nullptr, (float* const*)sym.ptrRing.push(data->data(), data->size() * sizeof(float*),
                                                   sym.stream)};
}

NumericCtxBase* HipSymbolicCtx::createNumericCtxForType(std::type_index tIdx, int64_t tempSize,
                                                        int batch) {
  if (tIdx == std::type_index(typeid(double))) return new HipNumericCtx<double>(*this, 1, tempSize);
  if (tIdx == std::type_index(typeid(float))) return new HipNumericCtx<float>(*this, 1, tempSize);
  if (tIdx == std::type_index(typeid(vector<double*>))) {
    return new HipNumericCtx<vector<double*>>(*this, batch, tempSize);
  }
  if (tIdx == std::type_index(typeid(vector<float*>))) {
    return new HipNumericCtx<vector<float*>>(*this, batch, tempSize);
  }
  return nullptr;
}

// Triangular solves on the device, fused over a lump range (SolveCtx extension of mat_ops.h).
template <typename T>
struct HipSolveCtx : SolveCtx<T> {
  using BT = BaseType<T>;
  HipSolveCtx(HipSymbolicCtx& sym_, int nRHS_, int batch_) : sym(sym_), nRHS(nRHS_), batch(batch_) {}

  // (TESTING switch hipBackendForcePerOp: the solver then drives the reference's op-by-op loops,
  //  Solver.cpp:303-328,357-381,419-448, over the per-op virtuals below)
  virtual bool hasFusedSolve() const override { return !sym.forcePerOp; }

  dim3 grid(unsigned x) const { return dim3(x, (unsigned)nRHS, (unsigned)batch); }

  template <bool BACKWARD>
  void elimRange(DevPlan& plan, const ElimRangePlan& er, hipk::SolveRef<BT> ref) {
    hipk::SkelDev sk = sym.skelDev();
    const int64_t nLumps = er.lumpEnd - er.lumpBegin;
    if (nLumps <= 0) return;
    auto small = [&] {
      const dim3 gL = grid((unsigned)((nLumps + 255) / 256));
      plan.ensureSolveGather(sym.skel);
      if (BACKWARD) {
        if (er.maxWidth <= 4) {
          const int64_t d0 = plan.solveGather.rangeLumpDesc[&er - plan.host.elimRanges.data()];
          // (several right-hand sides per workgroup: the blocks of L are fetched once per group)
          auto launch = [&](auto kern, int rb) {
            kern<<<dim3((unsigned)((nLumps + 15) / 16), (unsigned)((nRHS + rb - 1) / rb), (unsigned)batch),
                   256, 0, sym.stream>>>(plan.solveLumpDescs.as<SolveLumpDesc>() + d0,
                                         plan.solveLumpBlocks.as<SolveLumpBlock>(), ref, (int)nLumps, nRHS);
          };
          const auto wide = plan.solveGather.rangeWide[&er - plan.host.elimRanges.data()];
          const int64_t wideRows = sym.skel.order() - wide.rowBelow;
          const int64_t wideSpans = (int64_t)sym.skel.spanStart.size() - 1 - wide.spanBelow;
          const unsigned wideGroups = (unsigned)((nRHS + hipk::kWideRhs - 1) / hipk::kWideRhs);
          const size_t wideBytes = (size_t)hipk::wideGroupStride(wideRows) * wideGroups * (size_t)batch * sizeof(BT);
          if (nRHS == 1) {
            hipk::solveElimLumpsLt<BT><<<grid((unsigned)((nLumps + 15) / 16)), 256, 0, sym.stream>>>(
                plan.solveLumpDescs.as<SolveLumpDesc>() + d0, plan.solveLumpBlocks.as<SolveLumpBlock>(),
                ref, (int)nLumps);
          } else if (sym.solveWide && wide.n >= 1 && nRHS >= sym.solveWideMinRhs &&
                     wideBytes <= sym.solveWideMaxBytes) {
            // right-hand sides across the lanes (hip_solve_wide.h): the rows below the range, final by
            // now, go span by span into [16][rows] blocks first
            sym.solveWideBuf.resize(wideBytes);
            BT* yW = const_cast<BT*>(sym.solveWideBuf.as<BT>());
            if (wideSpans > 0) {
              hipk::solveRowsToWide<BT><<<dim3((unsigned)((wideSpans + 3) / 4), wideGroups, (unsigned)batch),
                                          256, 0, sym.stream>>>(ref, nRHS, sk.spanStart, wide.spanBelow,
                                                                wideSpans, wide.rowBelow, wideRows, yW);
            }
            auto launchW = [&](auto kern) {
              kern<<<dim3((unsigned)((nLumps + 15) / 16), wideGroups, (unsigned)batch), 256, 0, sym.stream>>>(
                  plan.solveLumpDescs.as<SolveLumpDesc>() + d0, plan.solveLumpBlocks.as<SolveLumpBlock>(),
                  ref, (int)nLumps, nRHS, yW, wide.rowBelow, wideRows);
            };
            switch (wide.n) {
              case 1: launchW(hipk::solveElimLumpsLtWide<BT, 1>); break;
              case 2: launchW(hipk::solveElimLumpsLtWide<BT, 2>); break;
              case 3: launchW(hipk::solveElimLumpsLtWide<BT, 3>); break;
              default: launchW(hipk::solveElimLumpsLtWide<BT, 4>); break;
            }
            sym.counters.solveWideLaunches++;
          } else if (nRHS <= 2) {
            launch(hipk::solveElimLumpsLtMulti<BT, 2>, 2);
          } else if (nRHS <= 4) {
            launch(hipk::solveElimLumpsLtMulti<BT, 4>, 4);
          } else {
            launch(hipk::solveElimLumpsLtMulti<BT, 8>, 8);
          }
        } else {
          hipk::solveElimSmall<BT, true><<<gL, 256, 0, sym.stream>>>(sk, ref, er.lumpBegin,
                                                                    er.lumpEnd);
        }
        return;
      }
      const auto items = plan.solveGather.rangeItems[&er - plan.host.elimRanges.data()];
      // (every right-hand side in the thread / 16 right-hand sides per workgroup: L is read once)
      hipk::solveElimDiagL<BT><<<dim3((unsigned)((nLumps + 255) / 256), 1, (unsigned)batch), 256, 0,
                                 sym.stream>>>(sk, ref, er.lumpBegin, er.lumpEnd, nRHS);
      if (items.second > items.first) {
        hipk::solveElimGatherL<BT><<<dim3((unsigned)(items.second - items.first),
                                          (unsigned)((nRHS + 15) / 16), (unsigned)batch),
                                     256, 0, sym.stream>>>(
            plan.solveItems.as<SolveGatherItem>() + items.first,
            plan.solveEntries.as<SolveGatherEntry>(), ref, nRHS);
      }
    };
    // lumps of a range are mutually independent: the order small/wide does not matter
    if (!BACKWARD) {
      small();
      denseLevels<false>(plan, er.bigLevels, ref);
    } else {
      denseLevels<true>(plan, er.bigLevels, ref);
      small();
    }
  }

  // The steps of a level list.  Consecutive one-panel levels that make up one outer block of a wide
  // lump are solved as a block (2 launches per 256 columns instead of 8); withSweeps: a run of
  // one-panel levels of one lump that is at least sweepMinWidth columns wide is ONE persistent launch.
  void buildSchedule(DevPlan& plan, const vector<LevelRange>& levels, bool withSweeps,
                     SolveInvList& ent) {
    static_assert(hipk::kSolveBlock == kOuterWidth, "block solve steps = outer blocks of the plan");
    const int64_t nL = (int64_t)levels.size();
    vector<PanelDesc> list;
    const bool inverses = sym.solveInv && sym.blockSolve;
    auto colOf = [&](const PanelDesc& p) { return p.lda - p.nRest - p.nb; };  // column inside its lump
    for (int64_t l = 0; l < nL;) {
      SolveSchedItem it;
      it.l = l;
      it.e = l + 1;
      if (withSweeps && levels[l].directPanel >= 0) {
        const PanelDesc& p0 = plan.host.panels[levels[l].directPanel];
        int64_t e = l + 1;
        int w = p0.nb;
        while (e < nL && levels[e].directPanel >= 0) {
          const PanelDesc& pe = plan.host.panels[levels[e].directPanel];
          if (pe.lump != p0.lump || colOf(pe) != colOf(p0) + w) break;
          if (plan.host.panels[levels[e - 1].directPanel].nb != kPanelWidth) break;
          w += pe.nb;
          e++;
        }
        if (w >= sym.sweepMinWidth) {
          const PanelDesc& last = plan.host.panels[levels[e - 1].directPanel];
          it.kind = SolveSchedItem::kSweep;
          it.e = e;
          it.slot = (int32_t)list.size();
          hipk::SweepDesc& sd = it.sweep;
          sd.diagOff = p0.diagOff;
          sd.lda = p0.lda;
          sd.w = w;
          sd.rowsBelow = last.rowsBelow;
          sd.nRest = last.nRest;
          sd.lumpRowBase = last.lumpRowBase;
          sd.vecOff = p0.vecOff;
          sd.nBlocks = (w + hipk::kSweepW - 1) / hipk::kSweepW;
          sd.invSlot = it.slot;
          sd.xchgOff = (int32_t)ent.sweepInstStride;
          sd.ticketOff = 0;  // (set per call: depends on the number of instances)
          ent.sweepInstStride += 2 * (int64_t)sd.nBlocks * hipk::kSweepW;
          it.sweepFwdWgs = hipk::kSweepFwdGroup * sd.nBlocks +
                           (sd.rowsBelow + hipk::kSweepFarRows - 1) / hipk::kSweepFarRows;
          it.sweepBwdWgs = hipk::kSweepBwdGroup * sd.nBlocks;
          ent.maxSweepWgs = std::max<int64_t>(ent.maxSweepWgs, std::max(it.sweepFwdWgs, it.sweepBwdWgs));
          ent.maxSweepWgsM = std::max<int64_t>(ent.maxSweepWgsM,
                                               std::max<int64_t>(it.sweepFwdWgs, hipk::kSweepFwdGroup * sd.nBlocks));
          ent.maxSweepW = std::max(ent.maxSweepW, sd.w);
          ent.maxSweepBelow = std::max(ent.maxSweepBelow, sd.rowsBelow);
          ent.numSweeps++;
          for (int64_t q = l; q < e; q++) list.push_back(plan.host.panels[levels[q].directPanel]);
          ent.items.push_back(it);
          l = e;
          continue;
        }
      }
      if (sym.blockSolve && levels[l].directPanel >= 0) {
        const PanelDesc& p0 = plan.host.panels[levels[l].directPanel];
        const int c0 = colOf(p0);
        if (c0 % kOuterWidth == 0) {
          int64_t e = l + 1;
          while (e < nL && levels[e].directPanel >= 0) {
            const PanelDesc& pe = plan.host.panels[levels[e].directPanel];
            const int ce = colOf(pe);
            if (pe.lump != p0.lump || ce != c0 + (int)(e - l) * kPanelWidth || ce >= c0 + kOuterWidth) {
              break;
            }
            e++;
          }
          it.e = e;
        }
      }
      if (it.e - it.l >= 2) {
        it.kind = SolveSchedItem::kBlockGroup;
        if (inverses) {
          it.slot = (int32_t)list.size();
          for (int64_t q = it.l; q < it.e; q++) list.push_back(plan.host.panels[levels[q].directPanel]);
        }
      }
      ent.items.push_back(it);
      l = it.e;
    }
    ent.count = (int64_t)list.size();
    if (ent.count) ent.list.upload(list);
    ent.built = true;
  }

  // may this call run the sweeps of `ent`?  Every workgroup of a sweep launch should be resident at
  // once (correct either way -- roles are dealt by ticket -- but a far role that starts late has a
  // whole row strip to catch up on at one CU's bandwidth), and its LDS must fit.
  // (several right-hand sides: the matrix-core sweep, 16 right-hand sides per set of workgroups)
  // (the matrix-core sweep pays on LONG runs only: a step is 16-17 us whatever the width below it, the block
  //  path's two launches per 256 columns cost less while the rows below are few; profiles/r06_solve10_families.txt)
  bool sweepMfma(const SolveInvList& ent) const {
    return nRHS >= sym.sweepMfmaMinRhs && ent.maxSweepW >= sym.sweepMfmaMinWidth;
  }
  bool sweepsFit(const SolveInvList& ent) {
    if (ent.numSweeps == 0) return false;
    if (!sym.sweepReady<BT>()) return false;
    if (sweepMfma(ent)) {
      const int64_t inst = (int64_t)((nRHS + hipk::kSweepR - 1) / hipk::kSweepR) * batch;
      return ent.maxSweepWgsM * inst <= sym.sweepCapacity && hipk::sweepMLdsBytes<BT>() <= sym.sweepMaxLds;
    }
    const int64_t inst = (int64_t)nRHS * batch;
    if (ent.maxSweepWgs * inst > sym.sweepCapacity) return false;
    const size_t lds = std::max(hipk::sweepLdsBytes<BT>(ent.maxSweepW, ent.maxSweepBelow, false),
                                hipk::sweepLdsBytes<BT>(ent.maxSweepW, ent.maxSweepBelow, true));
    return lds <= sym.sweepMaxLds;
  }

  template <bool BACKWARD>
  void denseLevels(DevPlan& plan, const vector<LevelRange>& levels, hipk::SolveRef<BT> ref) {
    // schedule: with persistent sweeps when the plan has wide runs and this call fits the GPU
    SolveInvList* entp = nullptr;
    if (sym.sweepUsable()) {
      SolveInvList& es = plan.solveInvLists[{&levels, 1}];
      if (!es.built) buildSchedule(plan, levels, true, es);
      if (sweepsFit(es)) entp = &es;
    }
    if (!entp) {
      SolveInvList& eo = plan.solveInvLists[{&levels, 0}];
      if (!eo.built) buildSchedule(plan, levels, false, eo);
      entp = &eo;
    }
    SolveInvList& ent = *entp;
    const bool sweeps = ent.numSweeps > 0 && entp == &plan.solveInvLists[{&levels, 1}];
    const int64_t nG = (int64_t)ent.items.size();
    // round 3 (BSP_SOLVE_INV=0 disables): the block triangles through inverted 64 x 64 diagonal
    // blocks -- one launch inverts the diagonal block of every panel of every block group and sweep
    // (the list of those panels is built and uploaded once per level list), the block kernel then
    // needs one round trip (hip_solve_kernels.h, K-B0 / K-B1i).  Round 6: the same launch arms the
    // exchange buffer of the sweeps.
    const BT* invBase = nullptr;
    int64_t invBatchStride = 0;
    BT* xchg = nullptr;
    hipk::SweepShared sh{};
    const bool mfma = sweeps && sweepMfma(ent);
    const int64_t rhsGroups = mfma ? (nRHS + hipk::kSweepR - 1) / hipk::kSweepR : nRHS;
    const int64_t nInst = rhsGroups * batch;
    const int64_t instStride = ent.sweepInstStride * (mfma ? hipk::kSweepR : 1);
    if (ent.count > 0) {
      invBatchStride = ent.count * kPanelWidth * kPanelWidth;
      sym.solveInvScratch.resize((size_t)(invBatchStride * batch) * sizeof(BT));
      BT* scratch = const_cast<BT*>(sym.solveInvScratch.as<BT>());
      unsigned long long* arm = nullptr;
      int64_t armWords = 0;
      if (sweeps) {
        const size_t ctlBytes = (((size_t)(1 + ent.numSweeps * nInst) * sizeof(unsigned)) + 255) / 256 * 256;
        const size_t bytes = (ctlBytes + (size_t)(nInst * instStride) * sizeof(BT) + 7) / 8 * 8;
        sym.sweepXchg.resize(bytes);
        arm = reinterpret_cast<unsigned long long*>(sym.sweepXchg.ptr);
        armWords = (int64_t)(bytes / 8);
        xchg = reinterpret_cast<BT*>(reinterpret_cast<char*>(sym.sweepXchg.ptr) + ctlBytes);
        sh.ctl = reinterpret_cast<unsigned*>(sym.sweepXchg.ptr);
        sh.hostErr = sym.sweepHostErrDev;
        sh.spinLimit = (long long)(sym.sweepSpinLimitS * 1e8);
        sh.instStride = instStride;
        sh.fault = sym.sweepFault;
        sh.pad = 0;
        sh.trace = nullptr;
        if (sym.sweepTraceOn) {
          sym.sweepTrace.resize(8 * 4096 * sizeof(long long));
          sh.trace = reinterpret_cast<long long*>(sym.sweepTrace.ptr);
        }
      }
      // (solve() = solveL then solveLt through ONE context, on the same factor: the backward pass finds
      //  the inverses its forward pass left in the scratch -- nothing else has written it in between --
      //  and only arms the exchange buffer)
      if (invCacheList == &ent && invCacheGen == sym.solveInvGen && invCacheGen != 0) {
        if (arm) hipCHECK(hipMemsetAsync(arm, 0xff, (size_t)armWords * 8, sym.stream));
        sym.counters.invReused++;
      } else {
        hipk::solveInvertPanels<BT><<<dim3((unsigned)ent.count, 1, (unsigned)batch), 64, 0, sym.stream>>>(
            ent.list.as<PanelDesc>(), scratch, invBatchStride, ref, arm, armWords);
        invCacheList = &ent;
        invCacheGen = ++sym.solveInvGen;
      }
      invBase = scratch;
    }
    int32_t sweepOrd = 0;
    for (int64_t gi = 0; gi < nG; gi++) {
      const int64_t gIdx = BACKWARD ? nG - 1 - gi : gi;
      const SolveSchedItem& item = ent.items[gIdx];
      if (item.kind == SolveSchedItem::kSweep) {
        hipk::SweepDesc sd = item.sweep;
        sd.ticketOff = (int32_t)((BACKWARD ? ent.numSweeps - 1 - sweepOrd : sweepOrd) * nInst);
        sweepOrd++;
        if (mfma) {
          const unsigned gx = (unsigned)(BACKWARD ? hipk::kSweepFwdGroup * sd.nBlocks : item.sweepFwdWgs);
          hipk::solveSweepM<BT, BACKWARD><<<dim3(gx, (unsigned)rhsGroups, (unsigned)batch), 256,
                                           hipk::sweepMLdsBytes<BT>(), sym.stream>>>(
              sd, nRHS, invBase, invBatchStride, xchg, sh, plan.rowGlobal.as<int32_t>(), ref);
          sym.counters.sweepLaunches++;
          sym.counters.sweepMfmaLaunches++;
          continue;
        }
        const size_t lds = hipk::sweepLdsBytes<BT>(sd.w, sd.rowsBelow, BACKWARD);
        const unsigned gx = (unsigned)(BACKWARD ? item.sweepBwdWgs : item.sweepFwdWgs);
        hipk::solveSweep<BT, BACKWARD><<<grid(gx), 256, lds, sym.stream>>>(
            sd, invBase, invBatchStride, xchg, sh, plan.rowGlobal.as<int32_t>(), ref);
        sym.counters.sweepLaunches++;
        continue;
      }
      if (item.kind == SolveSchedItem::kBlockGroup) {
        const PanelDesc& first = plan.host.panels[levels[item.l].directPanel];
        const PanelDesc& last = plan.host.panels[levels[item.e - 1].directPanel];
        const int w = (int)(item.e - 1 - item.l) * kPanelWidth + last.nb;
        const unsigned nT = (unsigned)((last.rowsBelow + kTile - 1) / kTile);
        if (invBase && item.slot >= 0) {
          const BT* inv = invBase + (int64_t)item.slot * kPanelWidth * kPanelWidth;
          if (!BACKWARD) {
            hipk::solveTriBlockInv<BT, false><<<grid(1), 256, 0, sym.stream>>>(first, w, inv, invBatchStride, ref);
            if (nT) {
              hipk::solveGemvBlockL<BT><<<grid(nT), 256, 0, sym.stream>>>(
                  first, last, w, plan.rowGlobal.as<int32_t>(), ref);
            }
          } else {
            if (nT) {
              hipk::solveGemvBlockLt<BT><<<grid(nT), 256, 0, sym.stream>>>(
                  first, last, w, plan.rowGlobal.as<int32_t>(), ref);
            }
            hipk::solveTriBlockInv<BT, true><<<grid(1), 256, 0, sym.stream>>>(first, w, inv, invBatchStride, ref);
          }
          continue;
        }
        if (!BACKWARD) {
          hipk::solveTriBlock<BT, false><<<grid(1), 256, 0, sym.stream>>>(first, w, ref);
          if (nT) {
            hipk::solveGemvBlockL<BT><<<grid(nT), 256, 0, sym.stream>>>(
                first, last, w, plan.rowGlobal.as<int32_t>(), ref);
          }
        } else {
          if (nT) {
            hipk::solveGemvBlockLt<BT><<<grid(nT), 256, 0, sym.stream>>>(
                first, last, w, plan.rowGlobal.as<int32_t>(), ref);
          }
          hipk::solveTriBlock<BT, true><<<grid(1), 256, 0, sym.stream>>>(first, w, ref);
        }
        continue;
      }
      const LevelRange& lr = levels[item.l];
      const unsigned nP = (unsigned)(lr.panelEnd - lr.panelBegin);
      const unsigned nT = (unsigned)(lr.trsmEnd - lr.trsmBegin);
      if (!nP) continue;
      const dim3 gP = grid(nP), gT = grid(nT);
      if (!BACKWARD) {
        hipk::solveTriPanel<BT, false><<<gP, 256, 0, sym.stream>>>(
            plan.panels.as<PanelDesc>(), plan.levelPanels.as<int32_t>() + lr.panelBegin, ref);
        if (nT) {
          hipk::solveGemvL<BT><<<gT, 256, 0, sym.stream>>>(
              plan.panels.as<PanelDesc>(), plan.trsmTasks.as<TrsmTask>() + lr.trsmBegin,
              plan.rowGlobal.as<int32_t>(), ref);
        }
      } else {
        if (nT) {
          hipk::solveGemvLt<BT><<<gT, 256, 0, sym.stream>>>(
              plan.panels.as<PanelDesc>(), plan.trsmTasks.as<TrsmTask>() + lr.trsmBegin,
              plan.rowGlobal.as<int32_t>(), ref);
        }
        hipk::solveTriPanel<BT, true><<<gP, 256, 0, sym.stream>>>(
            plan.panels.as<PanelDesc>(), plan.levelPanels.as<int32_t>() + lr.panelBegin, ref);
      }
    }
  }

  // single matrix: the pointers themselves; batch: the two pointer arrays, uploaded per call
  hipk::SolveRef<BT> makeRef(const T* data, T* C, int64_t ldc);

  virtual void solveLRange(const T* data, int64_t startLump, int64_t upToLump, T* C,
                           int64_t ldc) override {
    DevPlan& plan = sym.planFor(sym.sparseElimRanges, startLump, upToLump, /*tag=*/0);
    hipk::SolveRef<BT> ref = makeRef(data, C, ldc);
    for (const ElimRangePlan& er : plan.host.elimRanges) elimRange<false>(plan, er, ref);
    denseLevels<false>(plan, plan.host.levels, ref);
    hipCHECK(hipGetLastError());
  }

  virtual void solveLtRange(const T* data, int64_t startLump, int64_t upToLump, T* C,
                            int64_t ldc) override {
    DevPlan& plan = sym.planFor(sym.sparseElimRanges, startLump, upToLump, /*tag=*/0);
    hipk::SolveRef<BT> ref = makeRef(data, C, ldc);
    denseLevels<true>(plan, plan.host.levels, ref);
    for (auto it = plan.host.elimRanges.rbegin(); it != plan.host.elimRanges.rend(); ++it) {
      elimRange<true>(plan, *it, ref);
    }
    hipCHECK(hipGetLastError());
  }

  virtual void addMvRange(const T* data, int64_t startLump, int64_t upToLump, const T* in,
                          int64_t inStride, T* out, int64_t outStride, BaseType<T> alpha) override {
    addMvImpl(data, startLump, upToLump, in, inStride, out, outStride, alpha);
  }
  // (single-matrix types only; the batch types have no addMvFrom in the reference either)
  void addMvImpl(const BT* data, int64_t startLump, int64_t upToLump, const BT* in,
                 int64_t inStride, BT* out, int64_t outStride, BT alpha) {
    const CoalescedBlockMatrixSkel& sk = sym.skel;
    // (lump, row tile) list of the range: built and uploaded once per range, kept by the Solver
    auto& ent = sym.addMvTileLists[{startLump, upToLump}];
    if (!ent.first) {
      vector<int64_t> tiles;
      for (int64_t l = startLump; l < upToLump; l++) {
        const int64_t c0 = sk.chainColPtr[l], nCh = sk.chainColPtr[l + 1] - c0;
        const int64_t rows = sk.chainRowsTillEnd[c0 + nCh - 1];
        const int64_t n = sk.lumpStart[l + 1] - sk.lumpStart[l];
        for (int64_t r = 0; r < rows; r += kTile) {
          for (int64_t c = 0; c < n; c += hipk::kMvCols) {
            if (r < n && c > r + kTile - 1) break;  // entirely above the diagonal
            tiles.push_back(l);
            tiles.push_back(r);
            tiles.push_back(c);
          }
        }
      }
      ent.first.reset(new DevBuf);
      ent.first->upload(tiles);
      ent.second = tiles.size() / 3;
    }
    if (ent.second == 0) return;
    hipk::addMvKernel<BT><<<dim3((unsigned)ent.second, (unsigned)nRHS), 256, 0, sym.stream>>>(
        sym.skelDev(), ent.first->as<int64_t>(), data, in, inStride, out, outStride, alpha);
    hipCHECK(hipGetLastError());
  }
  template <typename V>
  void addMvImpl(const vector<V*>*, int64_t, int64_t, const vector<V*>*, int64_t, vector<V*>*,
                 int64_t, V) {
    throw std::runtime_error("HIP backend: addMvFrom is defined for single matrices only");
  }

  // ---- per-op boundary (MatOps.h:139-168; reference kernels MatOpsCuda.cu:836-1181).  solve() never
  // goes through these (fused path); they exist so that a driver written against the reference's
  // SolveCtx -- the op-by-op loops of Solver.cpp:303-328,357-381,419-448, kept in solver.cpp -- runs
  // on this backend.  Correct, not fast: the dense triangular solves build a one-off device plan.
  hipk::SolveRef<BT> matOnly(const T* data) { return refOf(data, (T*)nullptr, 0); }
  hipk::SolveRef<BT> vecOnly(const T* v, int64_t ld) { return refOf((const T*)nullptr, const_cast<T*>(v), ld); }
  hipk::SolveRef<BT> refOf(const BT* data, BT* C, int64_t ldc) { return {data, C, nullptr, nullptr, ldc}; }
  template <typename V>
  hipk::SolveRef<V> refOf(const vector<V*>* data, vector<V*>* C, int64_t ldc) {
    hipk::SolveRef<V> r{nullptr, nullptr, nullptr, nullptr, ldc};
    if (data) {
      BASPACHO_CHECK_EQ((int)data->size(), batch);
      r.mats = (const V* const*)sym.ptrRing.push(data->data(), data->size() * sizeof(V*), sym.stream);
    }
    if (C) {
      BASPACHO_CHECK_EQ((int)C->size(), batch);
      r.vecs = (V* const*)sym.ptrRing.push(C->data(), C->size() * sizeof(V*), sym.stream);
    }
    return r;
  }
  BT* tmpBuf() {  // order x nRHS values per batch entry, as CpuBaseSolveCtx::tmpBuf
    tmp.resize((size_t)tmpStride() * batch * sizeof(BT));
    return const_cast<BT*>(tmp.as<BT>());
  }
  int64_t tmpStride() const { return sym.skel.order() * nRHS; }

  virtual void sparseElimSolveL(const SymElimCtx& elimData, const T* data, int64_t lumpsBegin,
                                int64_t lumpsEnd, T* C, int64_t ldc) override {
    const HipSymElimCtx* e = dynamic_cast<const HipSymElimCtx*>(&elimData);
    BASPACHO_CHECK_NOTNULL(e);
    BASPACHO_CHECK_EQ(e->lumpsBegin, lumpsBegin);
    BASPACHO_CHECK_EQ(e->lumpsEnd, lumpsEnd);
    DevPlan& plan = sym.planFor({lumpsBegin, lumpsEnd}, lumpsBegin, lumpsEnd, /*tag=*/1);
    hipk::SolveRef<BT> ref = makeRef(data, C, ldc);
    for (const ElimRangePlan& er : plan.host.elimRanges) elimRange<false>(plan, er, ref);
    hipCHECK(hipGetLastError());
  }
  virtual void sparseElimSolveLt(const SymElimCtx& elimData, const T* data, int64_t lumpsBegin,
                                 int64_t lumpsEnd, T* C, int64_t ldc) override {
    const HipSymElimCtx* e = dynamic_cast<const HipSymElimCtx*>(&elimData);
    BASPACHO_CHECK_NOTNULL(e);
    BASPACHO_CHECK_EQ(e->lumpsBegin, lumpsBegin);
    BASPACHO_CHECK_EQ(e->lumpsEnd, lumpsEnd);
    DevPlan& plan = sym.planFor({lumpsBegin, lumpsEnd}, lumpsBegin, lumpsEnd, /*tag=*/1);
    hipk::SolveRef<BT> ref = makeRef(data, C, ldc);
    for (auto it = plan.host.elimRanges.rbegin(); it != plan.host.elimRanges.rend(); ++it) {
      elimRange<true>(plan, *it, ref);
    }
    hipCHECK(hipGetLastError());
  }

  virtual void symm(const T* data, int64_t offset, int64_t n, const T* C, int64_t offC, int64_t ldc,
                    T* D, int64_t ldd, BaseType<T> alpha) override {
    if (n <= 0) return;
    hipk::perOpSymm<BT><<<grid((unsigned)((n + 3) / 4)), 256, 0, sym.stream>>>(
        matOnly(data), offset, n, vecOnly(C, ldc), offC, vecOnly(D, ldd), alpha);
    hipCHECK(hipGetLastError());
  }

  // dense triangular solve with the n x n block at `offset`: the block's panels and row tiles as a
  // one-off plan (the same kernels as the fused path, panel by panel)
  template <bool BACKWARD>
  void denseTriSolve(const T* data, int64_t offset, int64_t n, T* C, int64_t offC, int64_t ldc) {
    if (n <= 0) return;
    opPlans.emplace_back(new DevPlan);
    DevPlan& plan = *opPlans.back();
    plan.host = buildDenseOpPlan(n, 0, offset, /*potrfOnly=*/true, /*vecOff=*/offC);
    vector<LevelRange> levels = plan.host.levels;  // (upload() keeps the level table)
    plan.upload();
    {
      FlagOff noBlocks(sym.blockSolve);  // (the block kernels expect the chain flags of a factor plan)
      denseLevels<BACKWARD>(plan, levels, makeRef(data, C, ldc));
    }
    hipCHECK(hipGetLastError());
  }
  virtual void solveL(const T* data, int64_t offset, int64_t n, T* C, int64_t offC,
                      int64_t ldc) override {
    denseTriSolve<false>(data, offset, n, C, offC, ldc);
  }
  virtual void solveLt(const T* data, int64_t offset, int64_t n, T* C, int64_t offC,
                       int64_t ldc) override {
    denseTriSolve<true>(data, offset, n, C, offC, ldc);
  }

  virtual void gemv(const T* data, int64_t offset, int64_t nRows, int64_t nCols, const T* A,
                    int64_t offA, int64_t lda, BaseType<T> alpha) override {
    if (nRows <= 0) return;
    BASPACHO_CHECK_LE(nRows, sym.skel.order());
    hipk::perOpGemv<BT><<<grid((unsigned)((nRows + 3) / 4)), 256, 0, sym.stream>>>(
        matOnly(data), offset, nRows, nCols, vecOnly(A, lda), offA, alpha, tmpBuf(), tmpStride(), nRHS);
    hipCHECK(hipGetLastError());
  }
  virtual void gemvT(const T* data, int64_t offset, int64_t nRows, int64_t nCols, T* A,
                     int64_t offA, int64_t lda, BaseType<T> alpha) override {
    if (nRows <= 0 || nCols <= 0) return;
    const int64_t colTiles = (nCols + 63) / 64;
    const int64_t rowChunks = (nRows + hipk::kPerOpRowChunk - 1) / hipk::kPerOpRowChunk;
    hipk::perOpGemvT<BT><<<grid((unsigned)(colTiles * rowChunks)), 256, 0, sym.stream>>>(
        matOnly(data), offset, nRows, nCols, vecOnly(A, lda), offA, alpha, tmpBuf(), tmpStride(), nRHS);
    hipCHECK(hipGetLastError());
  }
  virtual void assembleVec(int64_t chainColPtr, int64_t numColItems, T* C, int64_t ldc) override {
    if (numColItems <= 0) return;
    hipk::perOpAssembleVec<BT, false><<<grid((unsigned)numColItems), 256, 0, sym.stream>>>(
        sym.skelDev(), chainColPtr, vecOnly(C, ldc), tmpBuf(), tmpStride(), nRHS);
    hipCHECK(hipGetLastError());
  }
  virtual void assembleVecT(const T* C, int64_t ldc, int64_t chainColPtr,
                            int64_t numColItems) override {
    if (numColItems <= 0) return;
    hipk::perOpAssembleVec<BT, true><<<grid((unsigned)numColItems), 256, 0, sym.stream>>>(
        sym.skelDev(), chainColPtr, vecOnly(C, ldc), tmpBuf(), tmpStride(), nRHS);
    hipCHECK(hipGetLastError());
  }

  // ---- fragmented ops (MatOps.h:170-184, MatOpsFast.cpp:613-1018): whole MV / L / L^T sweeps over
  // a range of spans that are lumps of their own, one right-hand side, contiguous vector.  On this
  // backend they are the fused range kernels (a range of unmerged lumps is just a range).
  virtual bool hasFragmentedOps() override { return true; }
  virtual void fragmentedMV(const T* data, const T* x, int64_t spanBegin, int64_t spanEnd, T* y,
                            BaseType<T> alpha) override {
    checkFragmented(spanBegin, spanEnd);
    addMvImpl(data, spanBegin, spanEnd, x, sym.skel.order(), y, sym.skel.order(), alpha);
  }
  virtual void fragmentedSolveL(const T* data, int64_t spanBegin, int64_t spanEnd, T* y) override {
    checkFragmented(spanBegin, spanEnd);
    solveLRange(data, spanBegin, spanEnd, y, sym.skel.order());
  }
  virtual void fragmentedSolveLt(const T* data, int64_t spanBegin, int64_t spanEnd, T* y) override {
    checkFragmented(spanBegin, spanEnd);
    solveLtRange(data, spanBegin, spanEnd, y, sym.skel.order());
  }
  void checkFragmented(int64_t spanBegin, int64_t spanEnd) const {
    BASPACHO_CHECK_EQ(nRHS, 1);
    BASPACHO_CHECK_LE(spanBegin, spanEnd);
    BASPACHO_CHECK_LE(spanEnd, sym.skel.numLumps());
    // spans in the range are lumps of their own (Solver.cpp:299,352,413)
    BASPACHO_CHECK_EQ(sym.skel.lumpToSpan[spanBegin], spanBegin);
    BASPACHO_CHECK_EQ(sym.skel.lumpToSpan[spanEnd] - sym.skel.lumpToSpan[spanBegin], spanEnd - spanBegin);
  }

  HipSymbolicCtx& sym;
  int nRHS, batch;
  const SolveInvList* invCacheList = nullptr;  // inverses this context computed last (denseLevels)
  uint64_t invCacheGen = 0;
  DevBuf tmp;
  vector<std::unique_ptr<DevPlan>> opPlans;
};

template <>
hipk::SolveRef<double> HipSolveCtx<double>::makeRef(const double* data, double* C, int64_t ldc) {
  return {data, C, nullptr, nullptr, ldc};
}
template <>
hipk::SolveRef<float> HipSolveCtx<float>::makeRef(const float* data, float* C, int64_t ldc) {
  return {data, C, nullptr, nullptr, ldc};
}
template <>
hipk::SolveRef<double> HipSolveCtx<vector<double*>>::makeRef(const vector<double*>* data,
                                                             vector<double*>* C, int64_t ldc) {
  return refOf(data, C, ldc);
}
template <>
hipk::SolveRef<float> HipSolveCtx<vector<float*>>::makeRef(const vector<float*>* data,
                                                           vector<float*>* C, int64_t ldc) {
  return refOf(data, C, ldc);
}

SolveCtxBase* HipSymbolicCtx::createSolveCtxForType(std::type_index tIdx, int nRHS, int batch) {
  if (tIdx == std::type_index(typeid(double))) return new HipSolveCtx<double>(*this, nRHS, 1);
  if (tIdx == std::type_index(typeid(float))) return new HipSolveCtx<float>(*this, nRHS, 1);
  if (tIdx == std::type_index(typeid(vector<double*>))) {
    return new HipSolveCtx<vector<double*>>(*this, nRHS, batch);
  }
  if (tIdx == std::type_index(typeid(vector<float*>))) {
    return new HipSolveCtx<vector<float*>>(*this, nRHS, batch);
  }
  return nullptr;
}

struct HipOps : Ops {
  explicit HipOps(const HipBackendOptions& o) : options(o) {}
  virtual SymbolicCtxPtr createSymbolicCtx(const CoalescedBlockMatrixSkel& skel,
                                           const vector<int64_t>& permutation) override {
    return SymbolicCtxPtr(new HipSymbolicCtx(skel, permutation, options));
  }
  HipBackendOptions options;
};

}  // namespace

OpsPtr hipOps(const HipBackendOptions* options) {
  if (options) return OpsPtr(new HipOps(*options));
  HipBackendOptions o;  // (callers that build a Solver from a skeleton: defaults + environment)
  o.applyEnv();
  return OpsPtr(new HipOps(o));
}

int hipBackendReadExtents(unsigned long long* out, int maxLaunches) {
#if defined(BSP_KTRACE) && defined(BSP_TRACE_TILE)
  hipCHECK(hipDeviceSynchronize());
  const int n = std::min(maxLaunches, 2048);
  hipCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(hipk::bspLaunchExtent), sizeof(unsigned long long) * 4 * n));
  // reset: minima to the largest value, maxima to zero
  std::vector<unsigned long long> init(2048 * 4, 0ull);
  for (int i = 0; i < 2048; i++) init[4 * i] = ~0ull;
  hipCHECK(hipMemcpyToSymbol(HIP_SYMBOL(hipk::bspLaunchExtent), init.data(), sizeof(unsigned long long) * 4 * 2048));
  return n;
#else
  (void)out;
  (void)maxLaunches;
  return 0;
#endif
}

int hipBackendReadTrace(long long* out, int maxRecords) {
#if defined(BSP_KTRACE) || defined(BSP_TRACE_UPD)
  unsigned n = 0;
  hipCHECK(hipDeviceSynchronize());
  hipCHECK(hipMemcpyFromSymbol(&n, HIP_SYMBOL(hipk::bspTraceCount), sizeof n));
  const int cnt = (int)std::min<unsigned>(n, (unsigned)std::min(maxRecords, 8192));
  hipCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(hipk::bspTrace), sizeof(long long) * hipk::kTraceW * cnt));
  n = 0;
  hipCHECK(hipMemcpyToSymbol(HIP_SYMBOL(hipk::bspTraceCount), &n, sizeof n));
  return cnt;
#else
  (void)out;
  (void)maxRecords;
  return 0;
#endif
}

double hipBackendMfmaF64ProbeTflops() {
  const int blocks = 1024, iters = 4000;
  double* out = nullptr;
  hipCHECK(hipMalloc(&out, sizeof(double) * 256 * blocks));
  hipEvent_t a, b;
  hipCHECK(hipEventCreate(&a));
  hipCHECK(hipEventCreate(&b));
  hipk::mfmaF64Probe<<<blocks, 256>>>(out, iters);  // warm-up
  hipCHECK(hipEventRecord(a, nullptr));
  hipk::mfmaF64Probe<<<blocks, 256>>>(out, iters);
  hipCHECK(hipEventRecord(b, nullptr));
  hipCHECK(hipEventSynchronize(b));
  float ms = 0;
  hipCHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(out);
  return double(blocks) * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12;
}

void hipBackendForcePerOp(SymbolicCtx& sym, bool on) {
  HipSymbolicCtx* h = dynamic_cast<HipSymbolicCtx*>(&sym);
  BASPACHO_CHECK_NOTNULL(h);
  h->forcePerOp = on;
}

std::vector<int64_t> hipBackendPlanLevels(SymbolicCtx& sym, int64_t startLump, int64_t upToLump) {
  HipSymbolicCtx* h = dynamic_cast<HipSymbolicCtx*>(&sym);
  BASPACHO_CHECK_NOTNULL(h);
  HipPlanHost p = buildHipPlan(h->skel, h->sparseElimRanges, startLump, upToLump, h->planOpts);
  std::vector<int64_t> out;
  auto emit = [&](const vector<LevelRange>& levels, int64_t range) {
    for (const LevelRange& lr : levels) {
      int64_t maxNb = 0, maxRows = 0, sumRows = 0;
      for (int64_t q = lr.panelBegin; q < lr.panelEnd; q++) {
        const PanelDesc& pd = p.panels[p.levelPanels[q]];
        maxNb = std::max<int64_t>(maxNb, pd.nb);
        maxRows = std::max<int64_t>(maxRows, pd.rowsBelow);
        sumRows += pd.rowsBelow;
      }
      const int64_t row[8] = {range, lr.panelEnd - lr.panelBegin, maxNb, maxRows, lr.trsmEnd - lr.trsmBegin,
                              lr.updEnd - lr.updBegin, lr.defEnd - lr.defBegin, sumRows};
      out.insert(out.end(), row, row + 8);
    }
  };
  for (size_t r = 0; r < p.elimRanges.size(); r++) emit(p.elimRanges[r].bigLevels, (int64_t)r);
  emit(p.levels, -1);
  return out;
}

void hipBackendSetFault(SymbolicCtx& sym, int kind) {
  HipSymbolicCtx* h = dynamic_cast<HipSymbolicCtx*>(&sym);
  BASPACHO_CHECK_NOTNULL(h);
  h->faultDropElimUpdate = kind == 1;
  // kind 2: the spine of block 1 of every persistent solve sweep never publishes its x, with the
  // watchdog set to 50 ms: the launch must end by itself and the NEXT solve must report it
  h->sweepFault = kind == 2 ? 2 : 0;
  h->sweepSpinLimitS = kind == 2 ? 0.05 : 2.0;
}

int hipBackendReadSweepTrace(SymbolicCtx& sym, long long* out, int maxBlocks) {
  HipSymbolicCtx* h = dynamic_cast<HipSymbolicCtx*>(&sym);
  BASPACHO_CHECK_NOTNULL(h);
  if (!h->sweepTrace.ptr) return 0;
  hipCHECK(hipDeviceSynchronize());
  const int n = std::min(maxBlocks, 4096);
  hipCHECK(hipMemcpy(out, h->sweepTrace.ptr, (size_t)n * 4 * sizeof(long long), hipMemcpyDeviceToHost));
  return n;
}

HipRunCounters hipBackendRunCounters(SymbolicCtx& sym) {
  HipSymbolicCtx* h = dynamic_cast<HipSymbolicCtx*>(&sym);
  BASPACHO_CHECK_NOTNULL(h);
  HipRunCounters c;
  c.sweepLaunches = h->counters.sweepLaunches;
  c.sweepTimeouts = h->counters.sweepTimeouts;
  c.splitListsUsed = h->counters.splitListsUsed;
  c.subBatchesEnqueued = h->counters.subBatchesEnqueued;
  c.lookaheadForks = h->counters.lookaheadForks;
  c.gatherChunksOverlapped = h->counters.gatherChunksOverlapped;
  c.tailLaunches = h->counters.tailLaunches;
  c.sweepMfmaLaunches = h->counters.sweepMfmaLaunches;
  c.solveWideLaunches = h->counters.solveWideLaunches;
  c.invReused = h->counters.invReused;
  c.potrfFoldedLevels = h->counters.potrfFoldedLevels;
  c.sweepsRetired = h->sweepBroken ? 1 : 0;
  c.sweepErrorPending = (h->sweepHostErr && *reinterpret_cast<volatile unsigned*>(h->sweepHostErr)) ? 1 : 0;
  return c;
}

void hipBackendSetProfile(SymbolicCtx& sym, HipKernelProfile* prof, bool inSitu) {
  HipSymbolicCtx* h = dynamic_cast<HipSymbolicCtx*>(&sym);
  BASPACHO_CHECK_NOTNULL(h);
  h->profile = prof;
  h->profileInSitu = prof != nullptr && inSitu;
}

HipPlanStats hipBackendPlanStats(SymbolicCtx& sym, int64_t startLump, int64_t upToLump) {
  HipSymbolicCtx* h = dynamic_cast<HipSymbolicCtx*>(&sym);
  BASPACHO_CHECK_NOTNULL(h);
  // host-only: does not touch the device
  HipPlanHost p = buildHipPlan(h->skel, h->sparseElimRanges, startLump, upToLump, h->planOpts);
  HipPlanStats s;
  s.flops = p.flops;
  s.updElems = p.updElems;
  s.updFlops = p.updFlops;
  s.elimPairElems = p.elimPairElems;
  s.elimPairFlops = p.elimPairFlops;
  s.elimColElems = p.elimColElems;
  s.updFlopsDirect = p.updFlopsDirect;
  s.elimPairOperandElems = p.elimPairOperandElems;
  s.elimTargetElems = p.elimTargetElems;
  s.trsmFlops = p.trsmFlops;
  s.potrfFlops = p.potrfFlops;
  s.trsmFlopsMerged = p.trsmFlopsMerged;
  s.potrfFlopsFused = p.potrfFlopsFused;
  s.numLaunches = p.numLaunches;
  s.numLevels = (int64_t)p.levels.size();
  s.numPanels = (int64_t)p.panels.size();
  s.numSegs = (int64_t)p.segs.size();
  s.numUpdTasks = (int64_t)p.updTasks.size();
  s.numTrsmTasks = (int64_t)p.trsmTasks.size();
  s.chainTabEntries = (int64_t)p.chainOffTab.size();
  s.maxPanelsInLevel = p.maxPanelsInLevel;
  int64_t atomicTasks = 0;
  for (auto& t : p.updTasks) atomicTasks += t.atomic ? 1 : 0;
  s.numAtomicUpdTasks = atomicTasks;
  s.numForkLevels = p.numForkLevels;
  s.deferredFlops = p.deferredFlops;
  s.tailUpdFlops = p.tailUpdFlops;
  for (const LevelRange& lr : p.levels) s.numTailPanels += lr.tail ? 1 : 0;
  return s;
}

}  // namespace BaSpaCho
