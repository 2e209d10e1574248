#include "hip_plan.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <atomic>
#include <map>
#include <thread>
#include <tuple>

namespace BaSpaCho {

using std::vector;

namespace {

struct LumpCols {
  int64_t width, diagOff, rowsBelow, diagChains, chain0, nChains;
};

LumpCols lumpCols(const CoalescedBlockMatrixSkel& sk, int64_t l) {
  LumpCols g;
  g.width = sk.lumpStart[l + 1] - sk.lumpStart[l];
  g.chain0 = sk.chainColPtr[l];
  g.nChains = sk.chainColPtr[l + 1] - g.chain0;
  g.diagChains = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
  g.diagOff = sk.chainData[g.chain0];
  g.rowsBelow =
      sk.chainRowsTillEnd[g.chain0 + g.nChains - 1] - sk.chainRowsTillEnd[g.chain0 + g.diagChains - 1];
  return g;
}

// Gather form of the pair updates of one elimination range (see ElimGatherItem): enumerate every
// (column l, chains i<=j) pair, bucket by target chain, order by (target span, source width) and
// cut into work items.  Falls back (useGather=false -> atomic scatter kernel) when offsets do not
// fit 32 bits or a target block is larger than a wave handles.
// sort every bucket [ptr[c], ptr[c+1]) of `v` with `less`, buckets dealt to a few host threads
// (the pair lists of a bundle-adjustment problem hold tens of millions of entries)
template <typename V, typename Less>
void sortBuckets(V& v, const std::vector<int64_t>& ptr, Less less) {
  const int64_t nB = (int64_t)ptr.size() - 1;
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (hw == 1 || v.size() < ((size_t)1 << 16)) {
    for (int64_t c = 0; c < nB; c++) std::sort(v.begin() + ptr[c], v.begin() + ptr[c + 1], less);
    return;
  }
  std::atomic<int64_t> next{0};
  auto work = [&] {
    for (;;) {
      const int64_t c0 = next.fetch_add(8);
      if (c0 >= nB) return;
      for (int64_t c = c0; c < std::min(nB, c0 + 8); c++) {
        if (ptr[c + 1] > ptr[c]) std::sort(v.begin() + ptr[c], v.begin() + ptr[c + 1], less);
      }
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < hw; t++) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
}

void buildElimGather(const CoalescedBlockMatrixSkel& sk, HipPlanHost& plan, ElimRangePlan& er) {
  er.useGather = false;
  if (sk.dataSize() >= (int64_t(1) << 32)) return;
  const bool timing = plan.opts.planTiming;
  auto tic = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::cerr << "[elim gather plan] " << what << ": "
              << std::chrono::duration<double>(now - tic).count() << " s" << std::endl;
    tic = now;
  };
  struct Pair {
    int32_t si, width;  // column span of the target block, width of the source lump
    uint32_t offJ, offI;
  };
  // map span -> chain index inside the current target lump (rebuilt when the target changes)
  std::vector<int64_t> chainOfSpan(sk.numSpans(), -1);
  int64_t mappedLump = -1;
  auto mapTarget = [&](int64_t t) {
    if (t == mappedLump) return;
    for (int64_t c = sk.chainColPtr[t]; c < sk.chainColPtr[t + 1]; c++) chainOfSpan[sk.chainRowSpan[c]] = c;
    mappedLump = t;
  };
  // enumerate(f): f(targetChain, si, width, offJ, offI) for every pair; false if unsupported
  auto enumerate = [&](auto&& f) -> bool {
    for (int64_t l = er.lumpBegin; l < er.lumpEnd; l++) {
      LumpCols g = lumpCols(sk, l);
      if (g.width > 255) return false;
      const int64_t cBegin = g.chain0 + g.diagChains, cEnd = g.chain0 + g.nChains;
      auto srcOff = [&](int64_t c) { return sk.chainData[c]; };
      for (int64_t i = cBegin; i < cEnd; i++) {
        const int64_t si = sk.chainRowSpan[i];
        const int64_t siSize = sk.spanStart[si + 1] - sk.spanStart[si];
        mapTarget(sk.spanToLump[si]);
        for (int64_t j = i; j < cEnd; j++) {
          const int64_t sj = sk.chainRowSpan[j];
          if ((sk.spanStart[sj + 1] - sk.spanStart[sj]) * siSize > kGatherMaxElems) return false;
          const int64_t tc = chainOfSpan[sj];
          f(tc, si, g.width, srcOff(j), srcOff(i));
        }
      }
    }
    return true;
  };
  // pass 1: count per target chain; pass 2: fill the buckets in place
  const int64_t nChainsTot = (int64_t)sk.chainRowSpan.size();
  std::vector<int64_t> bucketPtr(nChainsTot + 1, 0);
  int64_t nPairs = 0;
  if (!enumerate([&](int64_t tc, int64_t, int64_t, int64_t, int64_t) {
        bucketPtr[tc + 1]++;
        nPairs++;
      })) {
    return;
  }
  lap("count pairs");
  // (pairBegin / pairEnd are int32 indices into the plan-wide pair streams, which accumulate over
  //  all elimination ranges: the RUNNING total has to fit, not just this range; otherwise the
  //  range keeps the scatter kernel, useGather = false)
  if ((int64_t)plan.elimPairOffJ.size() + nPairs >= (int64_t)INT32_MAX) return;
  for (int64_t c = 0; c < nChainsTot; c++) bucketPtr[c + 1] += bucketPtr[c];
  std::vector<Pair> sorted((size_t)nPairs);
  {
    std::vector<int64_t> cursor(bucketPtr.begin(), bucketPtr.end() - 1);
    mappedLump = -1;
    enumerate([&](int64_t tc, int64_t si, int64_t width, int64_t oj, int64_t oi) {
      sorted[cursor[tc]++] = Pair{(int32_t)si, (int32_t)width, (uint32_t)oj, (uint32_t)oi};
    });
  }
  lap("bucket by target chain");
  plan.elimPairOffJ.reserve(plan.elimPairOffJ.size() + (size_t)nPairs);
  plan.elimPairOffI.reserve(plan.elimPairOffI.size() + (size_t)nPairs);
  er.itemBegin = (int64_t)plan.elimItems.size();
  const int64_t maxPairs = std::max<int64_t>(8, plan.opts.gatherMaxPairs);
  vector<int64_t> itemRowTag;  // target chain of every emitted item
  sortBuckets(sorted, bucketPtr, [](const Pair& x, const Pair& y) {
    return x.si != y.si ? x.si < y.si : x.width < y.width;
  });
  lap("sort pairs");
  for (int64_t c = 0; c < nChainsTot; c++) {
    const int64_t b = bucketPtr[c], e = bucketPtr[c + 1];
    if (b == e) continue;
    const int64_t sj = sk.chainRowSpan[c];
    const int64_t rows = sk.spanStart[sj + 1] - sk.spanStart[sj];
    int64_t q = b;
    while (q < e) {
      int64_t q1 = q;  // [q, q1): same target block
      while (q1 < e && sorted[q1].si == sorted[q].si) q1++;
      const int64_t si = sorted[q].si;
      const int64_t t = sk.spanToLump[si];
      const size_t firstItem = plan.elimItems.size();
      {
        const double cols = double(sk.spanStart[si + 1] - sk.spanStart[si]);
        plan.elimTargetElems += sj == si ? double(rows) * (rows + 1) / 2 : double(rows) * cols;
      }
      int64_t u = q;
      while (u < q1) {  // split by source width and by length
        int64_t u1 = u;
        const uint32_t chunkOfU = sorted[u].offJ / kGatherChunkElems;
        while (u1 < q1 && sorted[u1].width == sorted[u].width && u1 - u < maxPairs &&
               sorted[u1].offJ / kGatherChunkElems == chunkOfU) {
          u1++;
        }
        ElimGatherItem it{};
        it.tgtOff = sk.chainData[c] + sk.spanOffsetInLump[si];
        it.pairBegin = (int32_t)plan.elimPairOffJ.size();
        for (int64_t k = u; k < u1; k++) {
          plan.elimPairOffJ.push_back(sorted[k].offJ);
          plan.elimPairOffI.push_back(sorted[k].offI);
        }
        it.pairEnd = (int32_t)plan.elimPairOffJ.size();
        it.firstJ = sorted[u].offJ;
        it.firstI = sorted[u].offI;
        it.tgtStride = (int32_t)(sk.lumpStart[t + 1] - sk.lumpStart[t]);
        it.rows = (int16_t)rows;
        it.cols = (int16_t)(sk.spanStart[si + 1] - sk.spanStart[si]);
        it.n = (int16_t)sorted[u].width;
        it.flags = (int16_t)(sj == si ? 2 : 0);
        plan.elimItems.push_back(it);
        itemRowTag.push_back(c);
        u = u1;
      }
      if (plan.elimItems.size() - firstItem > 1) {
        for (size_t k = firstItem; k < plan.elimItems.size(); k++) plan.elimItems[k].flags |= 1;
      }
      q = q1;
    }
  }
  er.itemEnd = (int64_t)plan.elimItems.size();
  er.useGather = true;
  // items whose target block has at most 16 elements (e.g. 3x3 blocks of automatically detected
  // ranges) go to the kernel that packs four items per wave: move them behind the others
  {
    vector<ElimGatherItem> large, tiny, tiny9, wide;
    vector<int64_t> tagL;
    for (int64_t k = er.itemBegin; k < er.itemEnd; k++) {
      const ElimGatherItem& it = plan.elimItems[k];
      if (int(it.rows) * int(it.cols) <= 16) {
        const int big = std::max(int(it.rows), int(it.cols));
        (big * std::max(big, int(it.n)) <= 9 ? tiny9 : tiny).push_back(it);
      } else if (it.rows > 16 || it.cols > 16) {
        wide.push_back(it);  // does not fit one 16x16 MFMA tile
      } else {
        large.push_back(it);
        tagL.push_back(itemRowTag[k - er.itemBegin]);
      }
    }
    auto dst = plan.elimItems.begin() + er.itemBegin;
    dst = std::copy(large.begin(), large.end(), dst);
    dst = std::copy(tiny9.begin(), tiny9.end(), dst);
    dst = std::copy(tiny.begin(), tiny.end(), dst);
    std::copy(wide.begin(), wide.end(), dst);
    itemRowTag = tagL;
    er.itemEnd = er.itemBegin + (int64_t)large.size();
    er.tinyBegin = er.itemEnd;
    er.tiny9End = er.tinyBegin + (int64_t)tiny9.size();
    er.tinyEnd = er.tiny9End + (int64_t)tiny.size();
    er.ldsBegin = er.tinyEnd;
    er.ldsEnd = er.ldsBegin + (int64_t)wide.size();
  }
  // XCD-aware order (speed only).  A workgroup takes 4 consecutive items and
  // workgroup b runs on XCD b % 8, each XCD with its own 4 MB L2.  All items of one target ROW (same
  // sj) read the same B_j source blocks, so a row is handed to ONE XCD (row r -> XCD r % 8, rows
  // balance the load statistically) instead of being sprayed over all eight L2s (measured L2 hit
  // rate 26 %).
  const int64_t g0 = 0, g1 = er.itemEnd - er.itemBegin;
  if (g1 - g0 >= 512) {
    const int64_t nItems = g1 - g0;
    vector<ElimGatherItem> tmp(plan.elimItems.begin() + er.itemBegin + g0,
                               plan.elimItems.begin() + er.itemBegin + g1);
    vector<vector<int64_t>> perXcd(8);
    int64_t row = -1, rowKey = -1;
    for (int64_t k = 0; k < nItems; k++) {
      if (itemRowTag[g0 + k] != rowKey) {
        rowKey = itemRowTag[g0 + k];
        row++;
      }
      perXcd[row % 8].push_back(k);
    }
    size_t cursor[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t remaining = nItems, out = er.itemBegin + g0;
    while (remaining > 0) {
      for (int x = 0; x < 8 && remaining > 0; x++) {
        int src = x;
        if (cursor[src] >= perXcd[src].size()) {  // this XCD's list is exhausted: steal
          size_t best = 0;
          for (int y = 0; y < 8; y++) {
            const size_t left = perXcd[y].size() - cursor[y];
            if (left > best) {
              best = left;
              src = y;
            }
          }
        }
        for (int q = 0; q < 4 && cursor[src] < perXcd[src].size(); q++) {
          plan.elimItems[out++] = tmp[perXcd[src][cursor[src]++]];
          remaining--;
        }
      }
    }
    BASPACHO_CHECK_EQ(out, er.itemBegin + g1);
  }
  lap("sort + emit items");
}

struct PanelBuild {
  int32_t panel;
  int32_t level;
};

// ---- dense-lump plan (DenseLumpPlan, hip_plan.h) ---------------------------------------------
// Rows of the lump column: [0, n) the diagonal region, [n, n + r) the rows below (boards).  Column
// blocks b = 0 .. NB-1 of kOuterWidth columns.  Who applies source block s to a 64-row tile starting
// at row rho (block i = rho / 256 while rho < n) of column block c > s:
//   rho < 256 (s + 2)            the chain of block s (window);
//   256 (s + 2) <= rho < 256 (s + 3), rho < n   the hand-over of block s (H2);
//   everything else              a bulk tile, released by T(s) / H1(s), with deadline
//                                dl = min(i - 2, c) for rho < n (the fork whose hand-over or T(c)
//                                touches the tile first), dl = c for the rows below the lump.
// At fork f (after the chain of block f) every released bulk tile with dl <= f + 1 is launched as the
// DUE list; further ones follow in deadline order as long as the fork's time budget lasts.
constexpr double kDlBulkTilesPerUs = 15.7;   // rank-256 fp64 tiles per microsecond beside the chain
constexpr double kDlChainBlockUs = 110.0;    // one outer block of the chain, hand-over included
constexpr double kDlTrsmTileCost = 2.2;      // a trsmBlock row tile in units of one update tile

struct DlUnit {
  int32_t s, rho, c, dl;  // source block, first row of the 64-row tile, column block, deadline
};

DenseLumpPlan buildDenseLump(const CoalescedBlockMatrixSkel& sk, HipPlanHost& plan, int64_t l,
                             const LumpCols& g, int32_t lumpRowBase,
                             const vector<SegDesc>& boardSegTemplates, double bulkAhead, int32_t group) {
  group = std::max(1, group);
  DenseLumpPlan dl;
  const int64_t n = g.width, R = g.width + g.rowsBelow;
  const int32_t NB = (int32_t)((n + kOuterWidth - 1) / kOuterWidth);
  dl.lump = (int32_t)l;
  dl.diagOff = g.diagOff;
  dl.n = (int32_t)n;
  dl.rowsTotal = (int32_t)R;
  auto blockWidth = [&](int64_t b) { return (int32_t)std::min<int64_t>(kOuterWidth, n - b * kOuterWidth); };

  // ---- chain steps
  for (int32_t b = 0; b < NB; b++) {
    const int64_t col0 = (int64_t)b * kOuterWidth, bw = blockWidth(b);
    const int64_t windowEnd = std::min<int64_t>(n, col0 + 2 * kOuterWidth);
    DlBlock blk{};
    blk.diagOff = g.diagOff + col0 * n + col0;
    blk.lda = (int32_t)n;
    blk.width = (int32_t)bw;
    blk.slot0 = (int32_t)dl.steps.size();
    blk.col0 = (int32_t)col0;
    for (int64_t c0 = col0; c0 < col0 + bw; c0 += kPanelWidth) {
      const int32_t nb = (int32_t)std::min<int64_t>(kPanelWidth, col0 + bw - c0);
      DlStep st{};
      st.pd.diagOff = g.diagOff + c0 * n + c0;
      st.pd.lda = (int32_t)n;
      st.pd.nb = nb;
      st.pd.rowsBelow = (int32_t)(windowEnd - c0 - nb);
      st.pd.nRest = st.pd.rowsBelow;
      st.pd.lumpRowBase = lumpRowBase;
      st.pd.lump = (int32_t)l;
      st.pd.vecOff = (int32_t)(sk.lumpStart[l] + c0);
      st.src.off = st.pd.diagOff + (int64_t)nb * n;
      st.src.lda = (int32_t)n;
      st.src.K = nb;
      st.src.rowsBelow = st.pd.rowsBelow;
      st.src.nRest = st.pd.rowsBelow;
      st.src.lumpRowBase = lumpRowBase;
      st.sd.src = -1;
      st.sd.kind = kSegIntra;
      st.sd.q0 = 0;
      st.sd.m = st.pd.rowsBelow;
      st.sd.tgtBase = g.diagOff + (c0 + nb) * n + (c0 + nb);
      st.sd.tgtStride = (int32_t)n;
      st.sd.lump = (int32_t)l;
      st.nTasks = 0;
      for (int32_t cT = 0; cT < st.sd.m; cT += kTile) st.nTasks += (st.pd.rowsBelow - cT + kTile - 1) / kTile;
      st.block = b;
      st.slot = (int32_t)dl.steps.size();
      dl.maxWindowRows = std::max(dl.maxWindowRows, st.pd.rowsBelow);
      // algorithmic work of the window tiles (plan statistics)
      {
        const double m = st.sd.m, rows = st.pd.rowsBelow;
        const double elems = m * rows - m * (m - 1) / 2;
        plan.updElems += elems;
        plan.updFlops += 2.0 * nb * elems;
        plan.updFlopsDirect += 2.0 * nb * elems;
      }
      dl.steps.push_back(st);
    }
    dl.blocks.push_back(blk);
  }
  dl.numSlots = (int32_t)dl.steps.size();
  for (size_t k = 0; k < dl.steps.size(); k++) {
    DlStep& st = dl.steps[k];
    if (k + 1 < dl.steps.size() && st.nTasks > 0) {
      st.next = dl.steps[k + 1].pd;
      st.stage = 1;
      // (a narrower next panel does not fill tile 0: the rows below it inside that tile would not be
      //  staged by the potrf workgroup; one tile: nothing to run beside the potrf)
      st.fuse = st.next.nb == kTile && st.nTasks >= 2;
    }
  }

  // ---- hand-over descriptors
  for (int32_t b = 0; b + 2 < NB; b++) {
    DlBlock& blk = dl.blocks[b];
    const int64_t rho0 = (int64_t)(b + 1) * kOuterWidth;  // first row of the next block
    const int64_t end = std::min<int64_t>(n, rho0 + 2 * kOuterWidth);
    blk.h2Src.off = g.diagOff + rho0 * n + blk.col0;
    blk.h2Src.lda = (int32_t)n;
    blk.h2Src.K = blk.width;
    blk.h2Src.rowsBelow = (int32_t)(end - rho0);
    blk.h2Src.nRest = blk.h2Src.rowsBelow;
    blk.h2Src.lumpRowBase = lumpRowBase;
    blk.h2Seg.src = -1;
    blk.h2Seg.kind = kSegIntra;
    blk.h2Seg.q0 = 0;
    blk.h2Seg.m = blk.h2Src.rowsBelow;
    blk.h2Seg.tgtBase = g.diagOff + rho0 * n + rho0;
    blk.h2Seg.tgtStride = (int32_t)n;
    blk.h2Seg.rowMin = kOuterWidth;
    blk.h2Seg.lump = (int32_t)l;
    blk.h2RowTile0 = kOuterWidth;
    blk.h2Tiles = 0;
    for (int32_t q = kOuterWidth; q < blk.h2Src.rowsBelow; q += kTile) blk.h2Tiles += q / kTile + 1;
    blk.h2Next = dl.steps[dl.blocks[b + 1].slot0].pd;
    const double m = blk.h2Seg.m, rows = blk.h2Src.rowsBelow, top = kOuterWidth;
    const double elems = (m * rows - m * (m - 1) / 2) - (top * top - top * (top - 1) / 2);
    plan.updElems += elems;
    plan.updFlops += 2.0 * blk.width * elems;
    plan.updFlopsDirect += 2.0 * blk.width * elems;
  }

  // ---- bulk units (s, row tile, c)
  vector<vector<DlUnit>> released(NB);  // by source block
  for (int32_t s = 0; s + 1 < NB; s++) {
    for (int64_t rho = (int64_t)(s + 3) * kOuterWidth; rho < n; rho += kTile) {
      const int32_t i = (int32_t)(rho / kOuterWidth);
      for (int32_t c = s + 1; c <= i; c++) released[s].push_back({s, (int32_t)rho, c, std::min(i - 2, c)});
    }
    for (int64_t rho = n; rho < R; rho += kTile) {
      for (int32_t c = s + 1; c < NB; c++) released[s].push_back({s, (int32_t)rho, c, c});
    }
  }
  // segments of the bulk tasks: per (source blocks s0..s1, column block), diagonal region and rows
  // below.  The sources a target has waiting are always CONSECUTIVE blocks, i.e. consecutive columns:
  // one task of rank 256 (s1 - s0 + 1) takes them all -- one read-modify-write of the target instead of
  // one per source block, and no two writers of a target inside a launch (no atomics).
  std::map<std::tuple<int32_t, int32_t, int32_t>, int32_t> segDiag, segBelow;
  auto segOf = [&](int32_t s0, int32_t s1, int32_t c, bool below) {
    auto& m = below ? segBelow : segDiag;
    const auto key = std::make_tuple(s0, s1, c);
    auto it = m.find(key);
    if (it != m.end()) return it->second;
    const int64_t blockEnd = (int64_t)(s1 + 1) * kOuterWidth;
    SrcDesc fs{};
    fs.off = g.diagOff + blockEnd * n + (int64_t)s0 * kOuterWidth;
    fs.lda = (int32_t)n;
    fs.K = (int32_t)((s1 - s0 + 1) * kOuterWidth);
    fs.nRest = (int32_t)(n - blockEnd);
    fs.rowsBelow = (int32_t)(below ? R - blockEnd : n - blockEnd);
    fs.lumpRowBase = lumpRowBase;
    plan.srcs.push_back(fs);
    SegDesc u{};
    u.src = (int32_t)plan.srcs.size() - 1;
    u.kind = kSegIntra;
    u.outer = 2;
    u.lump = (int32_t)l;
    u.q0 = (int32_t)((int64_t)c * kOuterWidth - blockEnd);
    u.m = blockWidth(c);
    u.tgtBase = g.diagOff + blockEnd * n + blockEnd;
    u.tgtStride = (int32_t)n;
    plan.segs.push_back(u);
    return m[key] = (int32_t)plan.segs.size() - 1;
  };
  // tasks of a list of units: the sources of one target merged, longest tasks first (a launch ends
  // with its longest tile), then by source, row tile and column tile (row-tile-major: the column tiles
  // of a row share the row operand)
  auto emitUnits = [&](vector<DlUnit>& units) -> std::pair<int64_t, int64_t> {
    const int64_t begin = (int64_t)plan.updTasks.size();
    std::sort(units.begin(), units.end(), [](const DlUnit& x, const DlUnit& y) {
      return x.rho != y.rho ? x.rho < y.rho : (x.c != y.c ? x.c < y.c : x.s < y.s);
    });
    struct Merged { int32_t s0, s1, rho, c; };
    vector<Merged> merged;
    for (const DlUnit& u : units) {
      if (!merged.empty() && merged.back().rho == u.rho && merged.back().c == u.c && merged.back().s1 + 1 == u.s) {
        merged.back().s1 = u.s;
      } else {
        merged.push_back({u.s, u.s, u.rho, u.c});
      }
    }
    std::stable_sort(merged.begin(), merged.end(), [](const Merged& x, const Merged& y) {
      const int32_t kx = x.s1 - x.s0, ky = y.s1 - y.s0;
      return kx != ky ? kx > ky : (x.s0 != y.s0 ? x.s0 < y.s0 : (x.rho != y.rho ? x.rho < y.rho : x.c < y.c));
    });
    std::map<std::pair<int32_t, int32_t>, int32_t> writers;
    for (const Merged& u : merged) writers[{u.rho, u.c}]++;
    for (const Merged& u : merged) {
      const bool below = u.rho >= n;
      const int32_t seg = segOf(u.s0, u.s1, u.c, below);
      const int64_t blockEnd = (int64_t)(u.s1 + 1) * kOuterWidth;
      const int32_t rT = (int32_t)(u.rho - blockEnd);
      const int32_t c0 = plan.segs[seg].q0, cEnd = c0 + plan.segs[seg].m;
      const int32_t atomic = writers[{u.rho, u.c}] > 1 ? 1 : 0;
      for (int32_t cT = c0; cT < cEnd && cT <= rT; cT += kTile) {
        plan.updTasks.push_back(UpdTask{seg, rT, cT, atomic});
        const double rows = std::min<double>(kTile, (below ? R : n) - u.rho);
        const double cols = std::min<int32_t>(kTile, cEnd - cT);
        const double elems = cT == rT ? cols * (cols + 1) / 2 + std::max(0.0, rows - cols) * cols : rows * cols;
        const double K = plan.srcs[plan.segs[seg].src].K;
        plan.updElems += elems;
        plan.updFlops += 2.0 * K * elems;
        plan.deferredFlops += 2.0 * K * elems;
      }
    }
    return {begin, (int64_t)plan.updTasks.size()};
  };
  // XCD-contiguous permutation of a task range (as emitLevels' xcdOrder)
  auto xcdOrder = [&](int64_t begin, int64_t end) {
    const int64_t cnt = end - begin;
    if (cnt < 64) return;
    vector<UpdTask> tmp(plan.updTasks.begin() + begin, plan.updTasks.begin() + end);
    const int64_t base = cnt / 8, extra = cnt % 8;
    int64_t chunkStart[9];
    chunkStart[0] = 0;
    for (int x = 0; x < 8; x++) chunkStart[x + 1] = chunkStart[x] + base + (x < extra ? 1 : 0);
    for (int64_t q = 0; q < cnt; q++) plan.updTasks[begin + q] = tmp[chunkStart[q % 8] + q / 8];
  };

  // ---- operations in enqueue order
  int32_t numEvents = 0;
  auto op = [&](int32_t kind, int32_t stream, int32_t a) -> DlOp& {
    DlOp o{};
    o.kind = kind;
    o.stream = stream;
    o.a = a;
    dl.ops.push_back(o);
    return dl.ops.back();
  };
  // the panels of a block left of a column act on the rows of a trsmBlock launch inside the kernel:
  // update work in the plan's accounting (2 w_l w_p per row and panel pair l < p)
  auto countBlockTrsm = [&](int32_t b, int64_t rows) {
    const DlBlock& blk = dl.blocks[b];
    double pairs = 0, left = 0;
    for (int32_t c0 = 0; c0 < blk.width; c0 += kPanelWidth) {
      const double w = std::min<int32_t>(kPanelWidth, blk.width - c0);
      pairs += left * w;
      left += w;
    }
    plan.updFlops += 2.0 * double(rows) * pairs;
    plan.updElems += double(rows) * std::max(0, blk.width - kPanelWidth);
  };
  bool potrfFused = false, rawValid = false;
  int32_t evDuePrev = -1;
  vector<DlUnit> pool;
  // cost of a unit in rank-256 tiles; the side streams' work is spread evenly over the forks left
  auto unitCost = [&](const DlUnit& u) {
    return u.rho >= n ? 4.0 : std::min<double>(4.0, double(u.rho / kTile - u.c * 4 + 1));
  };
  double costTotal = 0, costLaunched = 0, costReleased = 0;
  for (int32_t s = 0; s < NB; s++) {
    for (const DlUnit& u : released[s]) costTotal += unitCost(u);
    const int64_t tB = std::min<int64_t>(n, (int64_t)(s + 3) * kOuterWidth);
    if (tB < R) costTotal += kDlTrsmTileCost * double((R - tB + kTile - 1) / kTile);
  }
  vector<int32_t> optDone(NB, -1);  // event after the optional / board launches of a fork
  for (int32_t b = 0; b < NB; b++) {
    const DlBlock& blk = dl.blocks[b];
    const int32_t kEnd = b + 1 < NB ? dl.blocks[b + 1].slot0 : (int32_t)dl.steps.size();
    bool h1Ridden = false;
    for (int32_t k = blk.slot0; k < kEnd; k++) {
      const DlStep& st = dl.steps[k];
      if (!potrfFused) op(kDlPotrf, 0, k);
      if (st.nTasks > 0) {
        if (rawValid) {
          // the block's last step also carries the hand-over's block solve (H1: the rows of row block
          // b + 2 against the whole block -- its last panel was factored by the previous launch) as
          // extra workgroups: one launch less on the execution stream per outer block
          const int64_t hb = (int64_t)(b + 2) * kOuterWidth, he = std::min<int64_t>(n, hb + kOuterWidth);
          const bool ride = k == kEnd - 1 && hb < n && blk.width == kOuterWidth;
          if (ride && evDuePrev >= 0) op(kDlWait, 0, evDuePrev);
          DlOp& o = op(kDlStep, 0, k);
          if (ride) {
            o.rowBegin = (int32_t)hb;
            o.rowEnd = (int32_t)he;
            h1Ridden = true;
          }
          plan.trsmFlopsMerged += double(st.pd.rowsBelow) * st.pd.nb * st.pd.nb;
        } else {
          op(kDlTrsmPanel, 0, k);
          op(kDlStepUpd, 0, k);
        }
        if (st.fuse) plan.potrfFlopsFused += double(st.next.nb) * st.next.nb * st.next.nb / 3.0;
        potrfFused = st.fuse;
        rawValid = st.stage;
      } else {
        potrfFused = false;
        rawValid = false;
      }
    }
    // fork of block b
    const int64_t h1Begin = (int64_t)(b + 2) * kOuterWidth, h1End = std::min<int64_t>(n, h1Begin + kOuterWidth);
    const bool hasHand = h1Begin < n;
    const int64_t tBegin = std::min<int64_t>(n, (int64_t)(b + 3) * kOuterWidth), tEnd = R;
    const bool hasT = tBegin < tEnd;
    // bulk lists.  DUE: every released unit whose target the execution stream touches at the next
    // fork (dl <= b + 1).  OPTIONAL: units with dl >= b + 3 in deadline order -- never dl = b + 2, so
    // that the due launch of fork b + 1 cannot meet this launch on a target and only has to wait for
    // the optional launch of fork b - 1 -- for an even share of the work that is left.
    pool.insert(pool.end(), released[b].begin(), released[b].end());
    for (const DlUnit& u : released[b]) costReleased += unitCost(u);
    vector<DlUnit> due, opt, rest, keep;
    const bool lastFork = b + 1 >= NB;
    // (optional launches go out every `group` forks and take targets at least group + 3 forks away:
    //  the due launches of the forks in between never meet them on a target, so the chain runs on
    //  while a long optional launch -- rank 256 x group, one read-modify-write per target -- is at work)
    const bool groupFork = (b + 1) % group == 0;
    for (const DlUnit& u : pool) {
      (u.dl <= b + 1 || lastFork ? due : (groupFork && u.dl >= b + group + 3 ? rest : keep)).push_back(u);
    }
    double dueCost = 0;
    for (const DlUnit& u : due) dueCost += unitCost(u);
    const double tCost = hasT ? kDlTrsmTileCost * double((tEnd - tBegin + kTile - 1) / kTile) : 0.0;
    const double share = bulkAhead * (costTotal - costLaunched) / double(std::max(1, NB - 1 - b));
    double budget = share - dueCost - tCost;
    std::sort(rest.begin(), rest.end(), [](const DlUnit& x, const DlUnit& y) {
      return x.dl != y.dl ? x.dl < y.dl : (x.c != y.c ? x.c < y.c : (x.rho != y.rho ? x.rho < y.rho : x.s < y.s));
    });
    size_t take = 0;
    double optCost = 0;
    while (take < rest.size() && budget > 0) {
      const double cst = unitCost(rest[take]);
      budget -= cst;
      optCost += cst;
      take++;
    }
    // (the sources of one target stay together: a target's units are adjacent in the order above)
    while (take > 0 && take < rest.size() && rest[take].rho == rest[take - 1].rho && rest[take].c == rest[take - 1].c) {
      optCost += unitCost(rest[take]);
      take++;
    }
    opt.assign(rest.begin(), rest.begin() + take);
    keep.insert(keep.end(), rest.begin() + take, rest.end());
    pool.swap(keep);
    costLaunched += dueCost + optCost + tCost;
    // board targets of this block (rows below the lump x rows below the lump, scatter addressing)
    int64_t boardBegin = (int64_t)plan.updTasks.size(), boardEnd = boardBegin;
    if (g.rowsBelow > 0 && !boardSegTemplates.empty()) {
      const int64_t blockEnd = std::min<int64_t>(n, (int64_t)(b + 1) * kOuterWidth);
      SrcDesc sr{};
      sr.off = g.diagOff + blockEnd * n + blk.col0;
      sr.lda = (int32_t)n;
      sr.K = blk.width;
      sr.nRest = (int32_t)(n - blockEnd);
      sr.rowsBelow = (int32_t)(sr.nRest + g.rowsBelow);
      sr.lumpRowBase = lumpRowBase;
      plan.srcs.push_back(sr);
      const int32_t srcIdx = (int32_t)plan.srcs.size() - 1;
      for (SegDesc sg : boardSegTemplates) {
        sg.src = srcIdx;
        sg.q0 += sr.nRest;
        plan.segs.push_back(sg);
        const int32_t seg = (int32_t)plan.segs.size() - 1;
        for (int32_t cT = sg.q0; cT < sg.q0 + sg.m; cT += kTile) {
          for (int32_t rT = cT; rT < sr.rowsBelow; rT += kTile) plan.updTasks.push_back(UpdTask{seg, rT, cT, 0});
        }
        const double R2 = double(sr.rowsBelow - sg.q0), m = double(sg.m);
        plan.updElems += m * R2 - m * (m - 1) / 2;
        plan.updFlops += 2.0 * sr.K * (m * R2 - m * (m - 1) / 2);
      }
      boardEnd = (int64_t)plan.updTasks.size();
      xcdOrder(boardBegin, boardEnd);
    }
    if (std::getenv("BSP_DL_DUMP")) {
      fprintf(stderr, "fork %d: pool %zu due %zu opt %zu kept %zu share %.0f dueCost %.0f tCost %.0f total %.0f launched %.0f\n", b,
              pool.size() + due.size() + opt.size(), due.size(), opt.size(), pool.size(), share, dueCost, tCost, costTotal, costLaunched);
    }
    const auto dueRange = emitUnits(due);
    xcdOrder(dueRange.first, dueRange.second);
    const auto optRange = emitUnits(opt);
    xcdOrder(optRange.first, optRange.second);
    const bool anyBulk = dueRange.second > dueRange.first || optRange.second > optRange.first || boardEnd > boardBegin;
    // streams: 0 = execution stream (chain, hand-over), 1 = due stream (block solve, due tiles),
    // 2 = optional stream (optional tiles, board targets)
    const bool anyDue = dueRange.second > dueRange.first;
    const bool anyOpt = optRange.second > optRange.first || boardEnd > boardBegin;
    int32_t evCH = -1, evH1 = -1, evT = -1;
    if (hasT || anyDue || anyOpt) {
      evCH = numEvents++;
      op(kDlRecord, 0, evCH);
      plan.numForkLevels++;
    }
    if (hasT || anyDue) op(kDlWait, 1, evCH);
    if (hasT) {
      DlOp& o = op(kDlTrsmBlock, 1, b);
      o.rowBegin = (int32_t)tBegin;
      o.rowEnd = (int32_t)tEnd;
      countBlockTrsm(b, tEnd - tBegin);
    }
    if (anyOpt && (hasT || b > 0)) {  // everything the due stream has solved so far
      evT = numEvents++;
      op(kDlRecord, 1, evT);
    }
    if (hasHand) {
      countBlockTrsm(b, h1End - h1Begin);
      if (h1Ridden) {
        evH1 = evCH;  // (solved inside the block's last step)
      } else {
        if (evDuePrev >= 0) op(kDlWait, 0, evDuePrev);
        DlOp& o = op(kDlTrsmBlock, 0, b);
        o.rowBegin = (int32_t)h1Begin;
        o.rowEnd = (int32_t)h1End;
        if (anyDue || anyOpt) {
          evH1 = numEvents++;
          op(kDlRecord, 0, evH1);
        }
      }
      dl.blocks[b].h2Stage = rawValid ? 1 : 0;
      op(kDlHandUpd, 0, b);
    }
    if (anyDue) {
      if (evH1 >= 0 && evH1 != evCH) op(kDlWait, 1, evH1);
      // optional launches that may have written these targets: those of forks <= b - group - 2 (at the
      // last fork, where everything left is due: all of them)
      for (int32_t h = lastFork ? b - 1 : b - group - 2; h >= 0; h--) {
        if (optDone[h] >= 0) {
          op(kDlWait, 1, optDone[h]);
          break;
        }
      }
      DlOp& o = op(kDlBulk, 1, b);
      o.taskBegin = dueRange.first;
      o.taskEnd = dueRange.second;
      o.due = 1;
    }
    if (hasT || anyDue) {
      evDuePrev = numEvents++;
      op(kDlRecord, 1, evDuePrev);
    }
    if (anyOpt) {
      op(kDlWait, 2, evCH);
      if (evT >= 0) op(kDlWait, 2, evT);
      if (evH1 >= 0 && evH1 != evCH) op(kDlWait, 2, evH1);
      if (optRange.second > optRange.first) {
        DlOp& o = op(kDlBulk, 2, b);
        o.taskBegin = optRange.first;
        o.taskEnd = optRange.second;
      }
      if (boardEnd > boardBegin) {
        DlOp& o = op(kDlBulk, 2, b);
        o.taskBegin = boardBegin;
        o.taskEnd = boardEnd;
      }
      optDone[b] = numEvents++;
      op(kDlRecord, 2, optDone[b]);
    }
  }
  BASPACHO_CHECK(pool.empty());
  for (int32_t st = 1; st <= 2; st++) {  // join
    const int32_t ev = numEvents++;
    op(kDlRecord, st, ev);
    op(kDlWait, 0, ev);
  }
  dl.numEvents = numEvents;
  plan.maxChainRows = std::max<int64_t>(plan.maxChainRows, dl.maxWindowRows);
  plan.hasDeferred = plan.hasDeferred || plan.deferredFlops > 0;
  return dl;
}


}  // namespace

HipPlanOptions HipPlanOptions::fromEnv() {
  HipPlanOptions o;
  if (const char* e = std::getenv("BSP_DUE_STREAM")) o.dueStream = e[0] != '0';
  if (const char* e = std::getenv("BSP_DENSE_LUMP")) o.denseLump = e[0] != '0';
  o.planTiming = std::getenv("BSP_TIMING") != nullptr;
  if (const char* e = std::getenv("BSP_GATHER_MAX_PAIRS")) o.gatherMaxPairs = std::max(8, atoi(e));
  if (const char* e = std::getenv("BSP_BULK_AHEAD")) o.bulkAhead = std::atof(e);
  if (const char* e = std::getenv("BSP_DL_GROUP")) o.dlGroup = std::max(1, atoi(e));
  return o;
}

HipPlanHost buildHipPlan(const CoalescedBlockMatrixSkel& sk, const vector<int64_t>& elimRangesIn,
                         int64_t startLump, int64_t upToLump, const HipPlanOptions& opts) {
  HipPlanHost plan;
  plan.opts = opts;
  plan.startLump = startLump;
  plan.upToLump = upToLump;
  const int64_t nLumps = sk.numLumps();
  const int64_t denseFrom = elimRangesIn.empty() ? 0 : elimRangesIn.back();
  BASPACHO_CHECK_LT(sk.order(), (int64_t)INT32_MAX);
  // lookahead schedule (addPanels): assumed rate of the bulk update beside the chain, and the share
  // of the next block's estimated chain time handed to the side stream as optional work
  constexpr double kBulkFlopsPerUs = 33e6;
  const double bulkAhead = opts.bulkAhead;

  vector<vector<PanelBuild>> levelBuckets;       // dense levels
  auto bucketAt = [](vector<vector<PanelBuild>>& buckets, size_t lvl) -> vector<PanelBuild>& {
    if (buckets.size() <= lvl) buckets.resize(lvl + 1);
    return buckets[lvl];
  };

  // per-panel segment ranges (segments of one panel are contiguous in plan.segs)
  vector<int64_t> panelSegBegin, panelSegEnd;

  // ---- helper: cut a lump into outer blocks and panels; returns number of panels.
  // Segments of a panel: the remaining columns of its outer block (source = the panel, K = nb);
  // the last panel of an outer block also carries the segments of the block-wide source
  // (K = block width): rest of the lump and, if requested, every board of the lump column.
  auto addPanels = [&](int64_t l, const LumpCols& g, int32_t lumpRowBase, bool withBoards,
                       const vector<SegDesc>& boardSegTemplates, bool noSegs = false) {
    int32_t count = 0;
    const int64_t n = g.width;
    vector<int64_t> pendingFrom;  // per column block of this lump (lookahead schedule, see below)
    for (int64_t blockStart = 0; blockStart < n; blockStart += kOuterWidth) {
      const int64_t blockEnd = std::min<int64_t>(n, blockStart + kOuterWidth);
      for (int64_t c0 = blockStart; c0 < blockEnd; c0 += kPanelWidth, count++) {
        const int32_t nb = (int32_t)std::min<int64_t>(kPanelWidth, blockEnd - c0);
        PanelDesc pd;
        pd.diagOff = g.diagOff + c0 * n + c0;
        pd.lda = (int32_t)n;
        pd.nb = nb;
        pd.nRest = (int32_t)(n - c0 - nb);
        pd.rowsBelow = (int32_t)(pd.nRest + g.rowsBelow);
        pd.lumpRowBase = lumpRowBase;
        pd.lump = (int32_t)l;
        pd.vecOff = (int32_t)(sk.lumpStart[l] + c0);
        pd.pad = 0;
        plan.panels.push_back(pd);
        plan.potrfFlops += double(nb) * nb * nb / 3.0;
        plan.trsmFlops += double(pd.rowsBelow) * nb * nb;
        panelSegBegin.push_back((int64_t)plan.segs.size());
        if (noSegs) {  // (a dense-lump plan carries the lump's updates: addDenseLump)
          panelSegEnd.push_back((int64_t)plan.segs.size());
          continue;
        }
        const int64_t innerCols = blockEnd - c0 - nb;
        if (innerCols > 0) {
          SrcDesc sr{};
          sr.off = pd.diagOff + (int64_t)nb * n;
          sr.lda = (int32_t)n;
          sr.K = nb;
          sr.rowsBelow = pd.rowsBelow;
          sr.nRest = pd.nRest;
          sr.lumpRowBase = lumpRowBase;
          plan.srcs.push_back(sr);
          SegDesc s{};
          s.src = (int32_t)plan.srcs.size() - 1;
          s.kind = kSegIntra;
          s.q0 = 0;
          s.m = (int32_t)innerCols;
          s.tgtBase = g.diagOff + (c0 + nb) * n + (c0 + nb);
          s.tgtStride = (int32_t)n;
          plan.segs.push_back(s);
        }
        if (c0 + nb == blockEnd) {  // the outer block is complete
          SrcDesc sr{};
          sr.off = g.diagOff + blockEnd * n + blockStart;
          sr.lda = (int32_t)n;
          sr.K = (int32_t)(blockEnd - blockStart);
          sr.nRest = (int32_t)(n - blockEnd);
          sr.rowsBelow = (int32_t)(sr.nRest + g.rowsBelow);
          sr.lumpRowBase = lumpRowBase;
          if (sr.rowsBelow > 0) {
            plan.srcs.push_back(sr);
            const int32_t srcIdx = (int32_t)plan.srcs.size() - 1;
            if (sr.nRest > 0) {
              // "now": the next outer block's columns, by this block alone (rank-256), on the
              // execution stream
              SegDesc s{};
              s.src = srcIdx;
              s.kind = kSegIntra;
              s.outer = 1;
              s.lump = (int32_t)l;
              s.q0 = 0;
              s.m = (int32_t)std::min<int64_t>(sr.nRest, kOuterWidth);
              s.tgtBase = g.diagOff + blockEnd * n + blockEnd;
              s.tgtStride = (int32_t)n;
              plan.segs.push_back(s);
              // Columns further right go to the lookahead (side) stream, in DEADLINE order rather
              // than source order: a column block c only has to be up to date when block c-1's
              // "now" update reaches it, so what this block (and earlier ones) still owe to c is
              // applied as late as the side stream's load allows -- and then in ONE pass over
              // all pending source blocks (rank 256 x pending: the target tile is read and
              // written once).  pendingFrom[c] = first source block not yet applied to c.
              //   * first launch (outer = 2): c = b + 2, the columns the next block's "now" update
              //     touches (the execution stream waits for this launch);
              //   * optional (outer = 3): c = b + 3, b + 4 ... while the estimated time stays
              //     within the next block's chain (plan-time estimate; events enforce order).
              const int64_t b = blockStart / kOuterWidth;
              const int64_t numBlocks = (n + kOuterWidth - 1) / kOuterWidth;
              if ((int64_t)pendingFrom.size() < numBlocks) pendingFrom.assign(numBlocks, 0);
              auto unitFlops = [&](int64_t c) {
                const double m = double(std::min<int64_t>(kOuterWidth, n - c * kOuterWidth));
                const double R = double(n - c * kOuterWidth + g.rowsBelow);
                return 2.0 * double(blockEnd - pendingFrom[c] * kOuterWidth) * (m * R - m * (m - 1) / 2);
              };
              // one unit per pending source block (rank 256 each: tiles of one launch then take
              // about the same time -- a single pass of rank 256 x pending would leave the launch
              // waiting for its few longest tiles); several units on one target in one launch
              // accumulate with atomics (sd.pad = 1)
              // DUE STREAM (default; BSP_DUE_STREAM=0: one side stream as before): the due units
              // (c = b + 2) run on a stream of their own beside the optional ones instead of in
              // front of them.  Units of different launches that may then overlap on one column
              // block accumulate with atomics: every due unit, and the optional units that reach
              // the column block the NEXT block's due units go to (c = b + 3).
              const bool dueStream = opts.dueStream;
              auto pushUnit = [&](int64_t c, int32_t outerKind) {
                // bit 0: several units on this target in one launch; bit 1: the target may be met
                // by a launch of the OTHER side stream (due-stream mode; the kernels take a mask, so
                // a call that orders the launches on one stream ignores this bit)
                const int32_t multi = (pendingFrom[c] < b ? 1 : 0) |
                                      ((dueStream && (outerKind == 2 || c == b + 3)) ? 2 : 0);
                for (int64_t sb = pendingFrom[c]; sb <= b; sb++) {
                  SrcDesc fs = sr;
                  fs.off = g.diagOff + blockEnd * n + sb * kOuterWidth;
                  fs.K = (int32_t)(std::min<int64_t>(blockEnd, (sb + 1) * kOuterWidth) - sb * kOuterWidth);
                  plan.srcs.push_back(fs);
                  SegDesc u = s;
                  u.src = (int32_t)plan.srcs.size() - 1;
                  u.outer = outerKind;
                  u.q0 = (int32_t)(c * kOuterWidth - blockEnd);
                  u.m = (int32_t)std::min<int64_t>(kOuterWidth, n - c * kOuterWidth);
                  u.pad = multi;
                  plan.segs.push_back(u);
                }
                pendingFrom[c] = b + 1;
              };
              // (0.6 of the block's chain time: measured flat between 0.45 and 1.0, rounds 2-3; the
              //  BSP_BULK_AHEAD switch now belongs to the dense-lump schedule)
              double budgetUs = 0.6 * (118.0 + 0.012 * double(sr.rowsBelow));
              // (topping the first launch up to a full round of workgroups with the nearest
              //  optional targets was measured slower: the execution stream waits on it)
              int64_t c = b + 2;
              if (c < numBlocks) {
                budgetUs -= unitFlops(c) / kBulkFlopsPerUs;
                pushUnit(c++, 2);
              }
              for (; c < numBlocks && budgetUs > 0; c++) {
                budgetUs -= unitFlops(c) / kBulkFlopsPerUs;
                pushUnit(c, 3);
              }
            }
            if (withBoards) {
              for (SegDesc s : boardSegTemplates) {
                s.src = srcIdx;
                s.q0 += sr.nRest;
                plan.segs.push_back(s);
              }
            }
          }
        }
        panelSegEnd.push_back((int64_t)plan.segs.size());
      }
    }
    return count;
  };

  // ---- helper: per-row lookup arrays of the below-diagonal rows of a lump
  auto appendLumpRows = [&](int64_t, const LumpCols& g) {
    const int64_t belowChain0 = g.chain0 + g.diagChains;
    for (int64_t c = belowChain0; c < g.chain0 + g.nChains; c++) {
      const int64_t span = sk.chainRowSpan[c];
      const int64_t rows = sk.spanStart[span + 1] - sk.spanStart[span];
      for (int64_t i = 0; i < rows; i++) {
        plan.rowChain.push_back((int32_t)(c - belowChain0));
        plan.rowLocal.push_back((int32_t)i);
        plan.rowColOff.push_back((int32_t)(sk.spanOffsetInLump[span] + i));
        plan.rowGlobal.push_back((int32_t)(sk.spanStart[span] + i));
      }
    }
    BASPACHO_CHECK_LT((int64_t)plan.rowChain.size(), (int64_t)INT32_MAX);
  };

  // ---- sparse-elimination ranges inside [startLump, upToLump)
  vector<vector<vector<PanelBuild>>> elimBigBuckets;
  for (size_t r = 0; r + 1 < elimRangesIn.size(); r++) {
    const int64_t rb = elimRangesIn[r], re = elimRangesIn[r + 1];
    if (re > upToLump) break;
    if (rb < startLump) continue;
    ElimRangePlan er;
    er.lumpBegin = rb;
    er.lumpEnd = re;
    er.chainBegin = sk.chainColPtr[rb];
    er.chainEnd = sk.chainColPtr[re];
    er.chainLumpOff = (int64_t)plan.elimChainLump.size();
    er.maxWidth = 0;
    er.descBegin = (int64_t)plan.elimLumpDesc.size();
    vector<vector<PanelBuild>> big;
    for (int64_t l = rb; l < re; l++) {
      LumpCols g = lumpCols(sk, l);
      er.maxWidth = std::max<int32_t>(er.maxWidth, (int32_t)g.width);
      plan.elimLumpDesc.push_back({g.diagOff, (int32_t)g.width, (int32_t)g.rowsBelow});
      for (int64_t c = 0; c < g.nChains; c++) plan.elimChainLump.push_back((int32_t)l);
      {
        // pair updates of this column: for chains i<=j below the diagonal, |sj| x |si| elements
        // (lower triangle only when i==j)
        double rowsAfter = 0, chainsAfter = 0, pairElems = 0, operandElems = 0;
        for (int64_t c = g.chain0 + g.nChains - 1; c >= g.chain0 + g.diagChains; c--) {
          const int64_t span = sk.chainRowSpan[c];
          const double sz = double(sk.spanStart[span + 1] - sk.spanStart[span]);
          pairElems += sz * (sz + 1) / 2 + rowsAfter * sz;
          // both source blocks of every pair (i <= j): (s_i + s_j) * width values
          operandElems += double(g.width) * (2 * sz + chainsAfter * sz + rowsAfter);
          rowsAfter += sz;
          chainsAfter += 1;
        }
        plan.elimPairOperandElems += operandElems;
        plan.elimPairElems += pairElems;
        plan.elimPairFlops += 2.0 * g.width * pairElems;
        plan.elimColElems += double(g.width) * (g.width + g.rowsBelow);
      }
      plan.flops += double(g.width) * g.width * g.width / 3.0 +
                    double(g.rowsBelow) * g.width * g.width +
                    double(g.rowsBelow) * g.rowsBelow * g.width;
      if (g.width > kElimSmallMax) {
        const int32_t first = (int32_t)plan.panels.size();
        const int32_t rowBase = (int32_t)plan.rowChain.size();
        appendLumpRows(l, g);
        int32_t n = addPanels(l, g, rowBase, /*withBoards=*/false, {});
        for (int32_t j = 0; j < n; j++) bucketAt(big, j).push_back({first + j, j});
      }
    }
    elimBigBuckets.push_back(std::move(big));
    buildElimGather(sk, plan, er);
    plan.elimRanges.push_back(std::move(er));
  }
  // ---- dense lumps
  const int64_t denseBegin = std::max(startLump, denseFrom);
  vector<int32_t> lastLevelOfLump(nLumps, -1);
  // pre-pass: levels and panel counts of every dense lump; a lump of several outer blocks whose
  // panels are alone in their levels runs as a DenseLumpPlan
  vector<char> isDenseLump(nLumps, 0);
  if (opts.denseLump) {
    vector<int32_t> lastLvl(nLumps, -1), firstLvl(nLumps, 0), nPan(nLumps, 0), perLevel;
    for (int64_t l = denseBegin; l < upToLump; l++) {
      const int64_t n = sk.lumpStart[l + 1] - sk.lumpStart[l];
      int32_t np = 0;
      for (int64_t bs = 0; bs < n; bs += kOuterWidth) {
        np += (int32_t)((std::min<int64_t>(kOuterWidth, n - bs) + kPanelWidth - 1) / kPanelWidth);
      }
      int32_t level = 0;
      for (int64_t q = sk.boardRowPtr[l]; q < sk.boardRowPtr[l + 1] - 1; q++) {
        const int64_t s = sk.boardColLump[q];
        if (s >= denseBegin && s < l) level = std::max(level, lastLvl[s] + 1);
      }
      firstLvl[l] = level;
      nPan[l] = np;
      lastLvl[l] = level + np - 1;
      if ((int64_t)perLevel.size() < level + np) perLevel.resize(level + np, 0);
      for (int32_t j = 0; j < np; j++) perLevel[level + j]++;
    }
    for (int64_t l = denseBegin; l < upToLump; l++) {
      const int64_t n = sk.lumpStart[l + 1] - sk.lumpStart[l];
      bool alone = n > kOuterWidth;
      for (int32_t j = 0; alone && j < nPan[l]; j++) alone = perLevel[firstLvl[l] + j] == 1;
      isDenseLump[l] = alone;
    }
  }
  std::map<int32_t, int32_t> dlFirstPanel, dlOtherPanel;
  for (int64_t l = denseBegin; l < upToLump; l++) {
    LumpCols g = lumpCols(sk, l);
    plan.flops += double(g.width) * g.width * g.width / 3.0 +
                  double(g.rowsBelow) * g.width * g.width +
                  double(g.rowsBelow) * g.rowsBelow * g.width;

    // per-row lookup arrays of the below-diagonal rows
    const int32_t lumpRowBase = (int32_t)plan.rowChain.size();
    const int64_t belowChain0 = g.chain0 + g.diagChains;
    appendLumpRows(l, g);

    // one segment template per off-diagonal board (q0 relative to the first chain row)
    vector<SegDesc> boardSegs;
    const int64_t b0 = sk.boardColPtr[l], bEnd = sk.boardColPtr[l + 1] - 1;  // bEnd = sentinel
    const int64_t rowsAboveBelow = sk.chainRowsTillEnd[belowChain0 - 1];
    for (int64_t b = b0 + 1; b < bEnd; b++) {
      const int64_t t = sk.boardRowLump[b];
      const int64_t chFirst = sk.boardChainColOrd[b], chNext = sk.boardChainColOrd[b + 1];
      SegDesc s{};
      s.kind = kSegBoard;
      s.q0 = (int32_t)(sk.chainRowsTillEnd[g.chain0 + chFirst - 1] - rowsAboveBelow);
      s.m = (int32_t)(sk.chainRowsTillEnd[g.chain0 + chNext - 1] -
                      sk.chainRowsTillEnd[g.chain0 + chFirst - 1]);
      s.tgtStride = (int32_t)(sk.lumpStart[t + 1] - sk.lumpStart[t]);
      s.firstChainOrd = (int32_t)(chFirst - g.diagChains);
      s.chainTabPtr = (int64_t)plan.chainOffTab.size();
      s.tgtBase = t;  // temporarily: the target lump (replaced below by 0)
      // offsets, inside column t, of every chain of this column from the board onwards
      int64_t tc = sk.chainColPtr[t];
      const int64_t tcEnd = sk.chainColPtr[t + 1];
      for (int64_t c = g.chain0 + chFirst; c < g.chain0 + g.nChains; c++) {
        const int64_t span = sk.chainRowSpan[c];
        while (tc < tcEnd && sk.chainRowSpan[tc] < span) tc++;
        BASPACHO_CHECK(tc < tcEnd && sk.chainRowSpan[tc] == span);  // fill property
        plan.chainOffTab.push_back(sk.chainData[tc]);
      }
      boardSegs.push_back(s);
    }

    // level of the first panel: after every planned source column of block-row l
    int32_t level = 0;
    for (int64_t q = sk.boardRowPtr[l]; q < sk.boardRowPtr[l + 1] - 1; q++) {
      const int64_t s = sk.boardColLump[q];
      if (s >= denseBegin && s < l) level = std::max(level, lastLevelOfLump[s] + 1);
    }
    const int32_t first = (int32_t)plan.panels.size();
    const int32_t n = addPanels(l, g, lumpRowBase, /*withBoards=*/true, boardSegs, isDenseLump[l]);
    for (int32_t j = 0; j < n; j++) bucketAt(levelBuckets, level + j).push_back({first + j, level + j});
    lastLevelOfLump[l] = level + n - 1;
    if (isDenseLump[l]) {
      const int32_t id = (int32_t)plan.denseLumps.size();
      plan.denseLumps.push_back(buildDenseLump(sk, plan, l, g, lumpRowBase, boardSegs, bulkAhead, opts.dlGroup));
      dlFirstPanel[first] = id;
      for (int32_t j = 1; j < n; j++) dlOtherPanel[first + j] = id;
    }
  }

  // ---- emit task lists level by level
  auto emitLevels = [&](const vector<vector<PanelBuild>>& buckets, vector<LevelRange>& out) {
    std::map<int32_t, int64_t> lastDeferredLevel;  // lump -> level index that deferred tiles
    std::map<int32_t, std::vector<int64_t>> blockForks;  // lump -> its block-boundary fork levels
    for (const auto& bucket : buckets) {
      LevelRange lr;
      lr.panelBegin = (int64_t)plan.levelPanels.size();
      lr.trsmBegin = (int64_t)plan.trsmTasks.size();
      lr.updBegin = (int64_t)plan.updTasks.size();
      lr.waitDefLevel = -1;
      const int64_t levelIdx = (int64_t)out.size();
      // single-panel level followed by a single-panel level of the same lump?
      const size_t bi = (size_t)(&bucket - &buckets[0]);
      const bool chain = bucket.size() == 1 && bi + 1 < buckets.size() &&
                         buckets[bi + 1].size() == 1 &&
                         plan.panels[buckets[bi + 1][0].panel].lump == plan.panels[bucket[0].panel].lump;
      vector<UpdTask> deferred, deferredLate;  // due / optional
      int32_t nowSegs = 0, nowSeg = -1;  // segments with non-deferred 64x64 tiles in this level
      bool nowPlain = true;              // ... all intra-lump, non-atomic, untouched order
      // how many panels of this level hit each target lump
      std::map<int64_t, int> hits;
      for (const auto& pb : bucket) {
        for (int64_t s = panelSegBegin[pb.panel]; s < panelSegEnd[pb.panel]; s++) {
          if (plan.segs[s].kind == kSegBoard) hits[plan.segs[s].tgtBase]++;
        }
      }
      for (const auto& pb : bucket) {
        const PanelDesc& pd = plan.panels[pb.panel];
        plan.levelPanels.push_back(pb.panel);
        for (int32_t r = 0; r < pd.rowsBelow; r += kTile) plan.trsmTasks.push_back({pb.panel, r});
        for (int64_t s = panelSegBegin[pb.panel]; s < panelSegEnd[pb.panel]; s++) {
          const SegDesc& sd = plan.segs[s];
          const SrcDesc& sr = plan.srcs[sd.src];
          // (bit 1: only when launches of two side streams can meet, see pushUnit)
          const int32_t atomic = ((sd.kind == kSegBoard && hits[sd.tgtBase] > 1) || (sd.pad & 1) ? 1 : 0) |
                                 (sd.pad & 2);
          if (sd.outer == 1) {
            // this block-wide update touches columns that the previous block's deferred tiles
            // of the same lump also touch: they must have completed
            auto it = lastDeferredLevel.find(sd.lump);
            if (it != lastDeferredLevel.end()) lr.waitDefLevel = std::max(lr.waitDefLevel, it->second);
          }
          bool anyDeferred = false;
          const int32_t step = kTile;
          const bool defer = sd.outer >= 2;  // lookahead units: every tile goes to the side stream
          if (defer) {
            // ROW-tile-major: the (up to four) column tiles of one row tile are neighbours in the
            // list, so the XCD that gets this stretch of the list (xcdOrder) fetches the row
            // operand A_i once for all of them and keeps the unit's few column operands B_j (<= 0.5
            // MB) in its L2.  Column-major order handed the column tiles of a row to four different
            // XCDs, i.e. every 128 KB row operand crossed the fabric four times.
            const bool late = sd.outer == 3;  // (2: due at the block boundary)
            vector<UpdTask>& dst = late ? deferredLate : deferred;
            for (int32_t rT = sd.q0; rT < sr.rowsBelow; rT += step) {
              for (int32_t cT = sd.q0; cT < sd.q0 + sd.m && cT <= rT; cT += step) {
                // (the top-left tile of a column block is its tile (0,0): the chain may be
                //  applying early rank-64 updates to it at the same time, LevelRange::extraDiag)
                const int32_t a = atomic | ((rT == sd.q0 && cT == sd.q0) ? 1 : 0);
                dst.push_back(UpdTask{(int32_t)s, rT, cT, a});
              }
            }
            anyDeferred = true;
          } else {
            for (int32_t cT = sd.q0; cT < sd.q0 + sd.m; cT += step) {
              const int32_t a = atomic;
              for (int32_t rT = cT; rT < sr.rowsBelow; rT += step) {
                plan.updTasks.push_back(UpdTask{(int32_t)s, rT, cT, a});
                if (nowSeg != (int32_t)s) nowSegs++;
                nowSeg = (int32_t)s;
                if (sd.kind != kSegIntra || atomic) nowPlain = false;
              }
            }
          }
          if (anyDeferred) {
            // due units must not overtake the optional units forked two block boundaries ago
            auto& forks = blockForks[sd.lump];
            if (forks.empty() || forks.back() != levelIdx) {
              if (forks.size() >= 2) lr.optWaitLevel = std::max(lr.optWaitLevel, forks[forks.size() - 2]);
              forks.push_back(levelIdx);
            }
            lastDeferredLevel[sd.lump] = levelIdx;
          }
          const double R = double(sr.rowsBelow - sd.q0), m = double(sd.m);
          plan.updElems += m * R - m * (m - 1) / 2;
          plan.updFlops += 2.0 * sr.K * (m * R - m * (m - 1) / 2);
          if (anyDeferred) plan.deferredFlops += 2.0 * sr.K * (m * R - m * (m - 1) / 2);
        }
      }
      lr.panelEnd = (int64_t)plan.levelPanels.size();
      lr.trsmEnd = (int64_t)plan.trsmTasks.size();
      lr.updEnd = (int64_t)plan.updTasks.size();
      lr.defBegin = lr.updEnd;
      plan.updTasks.insert(plan.updTasks.end(), deferred.begin(), deferred.end());
      lr.defMid = (int64_t)plan.updTasks.size();
      plan.updTasks.insert(plan.updTasks.end(), deferredLate.begin(), deferredLate.end());
      lr.defEnd = (int64_t)plan.updTasks.size();
      // XCD-aware order: workgroup b lands on XCD b % 8 (observed dispatch; each XCD has its own
      // L2), so hand every XCD a CONTIGUOUS run of the tile list (neighbouring tiles share
      // operand rows) instead of every 8th tile.  Pure permutation: speed only.
      auto xcdOrder = [&](int64_t begin, int64_t end) {
        const int64_t cnt = end - begin;
        if (cnt < 64) return;
        vector<UpdTask> tmp(plan.updTasks.begin() + begin, plan.updTasks.begin() + end);
        const int64_t base = cnt / 8, extra = cnt % 8;
        int64_t chunkStart[9];
        chunkStart[0] = 0;
        for (int x = 0; x < 8; x++) chunkStart[x + 1] = chunkStart[x] + base + (x < extra ? 1 : 0);
        for (int64_t p = 0; p < cnt; p++) {
          const int64_t x = p % 8, k = p / 8;
          // positions p with p%8 == x receive chunk x in order (chunk sizes match by construction)
          plan.updTasks[begin + p] = tmp[chunkStart[x] + k];
        }
      };
      if (lr.defEnd > lr.defBegin) {
        plan.hasDeferred = true;
        plan.numForkLevels++;
      }
      if (bucket.size() == 1) {
        auto itF = dlFirstPanel.find(bucket[0].panel);
        if (itF != dlFirstPanel.end()) {
          lr.dl = itF->second;
          plan.numLaunches += (int64_t)plan.denseLumps[itF->second].ops.size();
        } else if (dlOtherPanel.count(bucket[0].panel)) {
          lr.dl = -2;
        }
      }
      if (bucket.size() == 1) {
        lr.directPanel = bucket[0].panel;
        if (nowSegs == 1 && nowPlain) {
          lr.directSeg = nowSeg;
          const SegDesc& sd = plan.segs[nowSeg];
          const SrcDesc& sr = plan.srcs[sd.src];
          const double R = double(sr.rowsBelow - sd.q0);
          const double m = double(sd.outer ? std::min<int32_t>(sd.m, kOuterWidth) : sd.m);
          plan.updFlopsDirect += 2.0 * sr.K * (m * R - m * (m - 1) / 2);
        }
        if (chain && lr.directSeg >= 0) {
          const SegDesc& sd = plan.segs[lr.directSeg];
          const PanelDesc& next = plan.panels[buckets[bi + 1][0].panel];
          {
            // intra-block step (rank-nb, its own block's remaining columns) with a full 64-column
            // tile (0,0) of a following block inside the lump
            const PanelDesc& pd0 = plan.panels[bucket[0].panel];
            const SrcDesc& sr0 = plan.srcs[sd.src];
            if (!sd.outer && sd.kind == kSegIntra && sd.q0 == 0 && sr0.K == pd0.nb &&
                pd0.nb == kPanelWidth && pd0.nRest - sd.m >= kTile) {
              lr.extraDiag = 1;
            }
          }
          // (a narrower next panel does not fill tile 0: its rows below the panel would be missed)
          if (sd.q0 == 0 && sd.rowMin == 0 && sd.tgtBase == next.diagOff &&
              sd.tgtStride == next.lda) {
            lr.rawNext = 1;
            plan.maxChainRows = std::max<int64_t>(plan.maxChainRows, next.rowsBelow);
          }
          if (sd.q0 == 0 && sd.rowMin == 0 && sd.tgtBase == next.diagOff && next.nb == kTile &&
              sd.tgtStride == next.lda && lr.updEnd - lr.updBegin >= 2) {
            lr.fuseNext = 1;
            const SrcDesc& sr = plan.srcs[sd.src];
            const int32_t nbLast = plan.panels[bucket[0].panel].nb;
            if (sd.outer && sr.K > nbLast) lr.splitK = sr.K - nbLast;
          }
        }
      }
      xcdOrder(lr.updBegin, lr.updEnd);
      xcdOrder(lr.defBegin, lr.defMid);
      xcdOrder(lr.defMid, lr.defEnd);
      plan.maxPanelsInLevel = std::max<int64_t>(plan.maxPanelsInLevel, lr.panelEnd - lr.panelBegin);
      plan.numLaunches += 1 + (lr.trsmEnd > lr.trsmBegin) + (lr.updEnd > lr.updBegin) +
                          (lr.defEnd > lr.defBegin);
      out.push_back(lr);
    }
    // which trsm / potrf run inside another launch (measurement only)
    for (size_t li = 0; li < out.size(); li++) {
      const LevelRange& lr = out[li];
      const bool directUpd = lr.directPanel >= 0 && lr.directSeg >= 0 && lr.updEnd > lr.updBegin;
      const bool nextDirect = li + 1 < out.size() && out[li + 1].directPanel >= 0;
      if (directUpd && lr.fuseNext && nextDirect) {
        const double nb = plan.panels[out[li + 1].directPanel].nb;
        plan.potrfFlopsFused += nb * nb * nb / 3.0;
      }
      if (li > 0 && directUpd && lr.trsmEnd > lr.trsmBegin) {
        const LevelRange& pr = out[li - 1];
        const bool staged = pr.directPanel >= 0 && pr.directSeg >= 0 && pr.updEnd > pr.updBegin &&
                            pr.rawNext;
        const SegDesc& sd = plan.segs[lr.directSeg];
        const PanelDesc& pd = plan.panels[lr.directPanel];
        const SrcDesc& sr = plan.srcs[sd.src];
        const bool intraStep = !sd.outer && sr.K == pd.nb;
        const bool blockLast = sd.outer == 1 && sr.K > pd.nb && (sr.K - pd.nb) % kTile == 0 &&
                               sr.rowsBelow == pd.rowsBelow && sr.lda == pd.lda;
        if (staged && (intraStep || blockLast)) {
          plan.trsmFlopsMerged += double(pd.rowsBelow) * pd.nb * pd.nb;
        }
      }
    }
  };
  for (size_t r = 0; r < plan.elimRanges.size(); r++) {
    emitLevels(elimBigBuckets[r], plan.elimRanges[r].bigLevels);
    plan.numLaunches += 2;
  }
  emitLevels(levelBuckets, plan.levels);

  // board segments carried their target lump in tgtBase only for the atomic analysis
  for (auto& s : plan.segs) {
    if (s.kind == kSegBoard) s.tgtBase = 0;
  }
  BASPACHO_CHECK_LT((int64_t)plan.segs.size(), (int64_t)INT32_MAX);
  if (std::getenv("BSP_PLAN_TILE_STATS")) {
    // 16x16x4 MFMA steps the update tiles execute (whole 64 x 64 tiles, the strictly upper wave of
    // a diagonal tile skipped) against those that touch a wanted entry
    double run = 0, need = 0, needWave = 0, elems = 0;
    for (const UpdTask& t : plan.updTasks) {
      const SegDesc& sd = plan.segs[t.seg];
      const SrcDesc& sr = plan.srcs[sd.src];
      const int kSteps = (sr.K + 3) / 4, segEnd = sd.q0 + sd.m;
      const bool diag = t.rowTile == t.colTile;
      run += (diag ? 12.0 : 16.0) * kSteps;
      bool live[4][4];
      for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) {
          const int r0 = std::max<int>(t.rowTile + 16 * i, sd.rowMin), r1 = std::min(t.rowTile + 16 * i + 16, sr.rowsBelow);
          const int c0 = t.colTile + 16 * j, c1 = std::min(t.colTile + 16 * j + 16, segEnd);
          live[i][j] = r0 < r1 && c0 < c1 && r1 - 1 >= c0;
          if (live[i][j]) need += kSteps;
          for (int r = r0; r < r1; r++) {
            for (int c = c0; c < c1; c++) elems += r >= c;
          }
        }
      }
      for (int wi = 0; wi < 2; wi++) {
        for (int wj = 0; wj < 2; wj++) {
          if (live[2 * wi][2 * wj] || live[2 * wi][2 * wj + 1] || live[2 * wi + 1][2 * wj] || live[2 * wi + 1][2 * wj + 1]) needWave += 4.0 * kSteps;
        }
      }
    }
    fprintf(stderr, "update tiles: %zu, MFMA steps run %.3g, needed (16x16 granular) %.3g = %.1f %%, (32x32 wave granular) %.1f %%, fill of the run steps %.1f %%\n",
            plan.updTasks.size(), run, need, 100 * need / run, 100 * needWave / run, 100 * elems / (run * 256 / 1.0) * 1.0);
  }
  return plan;
}

// Symbolic replay of a dense-lump schedule: see hip_plan.h.
std::string verifyDenseLump(const HipPlanHost& plan, const DenseLumpPlan& dl) {
  const int64_t n = dl.n, R = dl.rowsTotal;
  const int32_t P = (int32_t)dl.steps.size();                 // 64-column panels
  const int32_t Td = (int32_t)((n + kTile - 1) / kTile);      // row tiles of the diagonal region
  const int32_t Tb = (int32_t)((R - n + kTile - 1) / kTile);  // row tiles below it
  const int32_t TT = Td + Tb;
  if (P != Td) return "panels and diagonal row tiles differ";
  auto rowTileOf = [&](int64_t rho) -> int32_t {
    return rho < n ? (int32_t)(rho / kTile) : Td + (int32_t)((rho - n) / kTile);
  };
  auto at = [&](int32_t t, int32_t k) { return (size_t)t * P + k; };
  vector<vector<uint8_t>> applied((size_t)TT * P);  // per (row tile, panel): count per source panel
  for (int32_t t = 0; t < TT; t++) {
    for (int32_t k = 0; k < P && (t >= Td || k <= t); k++) applied[at(t, k)].assign(k, 0);
  }
  vector<uint8_t> solved((size_t)TT * P, 0);
  // happens-before bookkeeping: per tile, the last writer / reader of each stream
  constexpr int NS = 3;
  struct Acc { int32_t w[NS] = {-1, -1, -1}, r[NS] = {-1, -1, -1}; };
  vector<Acc> acc((size_t)TT * P);
  // known[s][o]: last op of stream o that is ordered before stream s's next op (vector clocks)
  int32_t known[NS][NS];
  for (auto& kr : known) {
    for (int32_t& v : kr) v = -1;
  }
  struct Clock { int32_t v[NS]; };
  vector<Clock> eventClock(dl.numEvents, Clock{{-1, -1, -1}});
  vector<int32_t> eventStream(dl.numEvents, -1);
  std::string err;
  auto fail = [&](const std::string& what, int32_t opIdx) {
    if (err.empty()) err = what + " (op " + std::to_string(opIdx) + ")";
  };
  auto touch = [&](int32_t t, int32_t k, bool write, int32_t st, int32_t idx) {
    Acc& a = acc[at(t, k)];
    for (int32_t o = 0; o < NS; o++) {
      if (o == st) continue;
      if (a.w[o] > known[st][o]) fail("unordered access after a write of stream " + std::to_string(o) + " at tile " + std::to_string(t) + "," + std::to_string(k), idx);
      if (write && a.r[o] > known[st][o]) fail("unordered write after a read of stream " + std::to_string(o) + " at tile " + std::to_string(t) + "," + std::to_string(k), idx);
    }
    (write ? a.w[st] : a.r[st]) = idx;
  };
  auto allApplied = [&](int32_t t, int32_t k, int32_t upTo) {
    const auto& v = applied[at(t, k)];
    for (int32_t p = 0; p < upTo; p++) {
      if (v[p] != 1) return false;
    }
    return true;
  };
  auto apply = [&](int32_t t, int32_t k, int32_t p0, int32_t p1, int32_t idx) {
    if (solved[at(t, k)]) fail("update of a solved tile " + std::to_string(t) + "," + std::to_string(k), idx);
    for (int32_t p = p0; p < p1; p++) {
      if (!solved[at(t, p)] || !solved[at(k, p)]) fail("update from unsolved rows, tile " + std::to_string(t) + "," + std::to_string(k) + " source " + std::to_string(p), idx);
      if (++applied[at(t, k)][p] != 1) fail("source applied twice, tile " + std::to_string(t) + "," + std::to_string(k) + " source " + std::to_string(p), idx);
    }
  };
  auto potrf = [&](int32_t k, int32_t st, int32_t idx) {
    if (!allApplied(k, k, k)) fail("potrf of an incomplete diagonal tile " + std::to_string(k), idx);
    if (solved[at(k, k)]) fail("second potrf of panel " + std::to_string(k), idx);
    solved[at(k, k)] = 1;
    touch(k, k, true, st, idx);
  };
  auto windowEndTile = [&](const DlStep& s, int32_t k) { return k + 1 + (s.pd.rowsBelow + kTile - 1) / kTile; };
  auto trsmPanelWindow = [&](int32_t k, int32_t st, int32_t idx) {
    const DlStep& s = dl.steps[k];
    if (!solved[at(k, k)]) fail("trsm before the potrf of panel " + std::to_string(k), idx);
    for (int32_t t = k + 1; t < windowEndTile(s, k); t++) {
      if (!allApplied(t, k, k)) fail("trsm of an incomplete tile " + std::to_string(t) + "," + std::to_string(k), idx);
      if (solved[at(t, k)]) fail("tile solved twice " + std::to_string(t) + "," + std::to_string(k), idx);
      solved[at(t, k)] = 1;
      touch(t, k, true, st, idx);
    }
  };
  auto updWindow = [&](int32_t k, bool fused, int32_t st, int32_t idx) {
    const DlStep& s = dl.steps[k];
    const int32_t tEnd = windowEndTile(s, k);
    for (int32_t t = k + 1; t < tEnd; t++) {
      for (int32_t c = k + 1; c <= t; c++) {
        apply(t, c, k, k + 1, idx);
        touch(t, c, true, st, idx);
      }
    }
    if (fused) potrf(k + 1, st, idx);
  };
  auto trsmBlockRows = [&](int32_t blockIdx, int32_t rowBegin, int32_t rowEnd, int32_t st, int32_t idx) {
        const DlBlock& b = dl.blocks[blockIdx];
        const int32_t k0 = b.slot0, k1 = k0 + (b.width + kPanelWidth - 1) / kPanelWidth;
        for (int32_t k = k0; k < k1; k++) {
          for (int32_t c = k0; c <= k; c++) {
            if (c == k && !solved[at(k, k)]) fail("trsmBlock before the block's potrf", idx);
            if (c < k && !solved[at(k, c)]) fail("trsmBlock before the block is solved", idx);
            touch(k, c, false, st, idx);
          }
        }
        if (rowBegin % kTile != 0 && rowBegin != n) fail("trsmBlock rows not tile aligned", idx);
        for (int64_t rho = rowBegin; rho < rowEnd; rho += kTile) {
          const int32_t t = rowTileOf(rho);
          for (int32_t k = k0; k < k1; k++) {
            if (!allApplied(t, k, k0)) fail("trsmBlock of an incomplete tile " + std::to_string(t) + "," + std::to_string(k), idx);
            for (int32_t p = k0; p < k; p++) {  // (applied inside the kernel)
              if (++applied[at(t, k)][p] != 1) fail("source applied twice inside trsmBlock", idx);
            }
            if (solved[at(t, k)]) fail("tile solved twice " + std::to_string(t) + "," + std::to_string(k), idx);
            solved[at(t, k)] = 1;
            touch(t, k, true, st, idx);
          }
          if (rho < n && rho + kTile > n && rowEnd > n) {
            // (the ragged last tile of the diagonal region: the rows below start a tile of their own)
            rho = n - kTile;
          }
        }
  };
  int32_t idx = 0;
  for (const DlOp& o : dl.ops) {
    const int32_t st = o.stream;
    switch (o.kind) {
      case kDlPotrf: potrf(o.a, st, idx); break;
      case kDlTrsmPanel: trsmPanelWindow(o.a, st, idx); break;
      case kDlStep:
        if (o.rowEnd > o.rowBegin) trsmBlockRows(dl.steps[o.a].block, o.rowBegin, o.rowEnd, st, idx);
        trsmPanelWindow(o.a, st, idx);
        updWindow(o.a, dl.steps[o.a].fuse != 0, st, idx);
        break;
      case kDlStepUpd: updWindow(o.a, dl.steps[o.a].fuse != 0, st, idx); break;
      case kDlTrsmBlock: trsmBlockRows(o.a, o.rowBegin, o.rowEnd, st, idx); break;
      case kDlHandUpd: {
        const DlBlock& b = dl.blocks[o.a];
        const int32_t k0 = b.slot0, k1 = k0 + kOuterWidth / kPanelWidth;
        const int32_t t0 = (b.col0 + kOuterWidth) / kTile;  // first row tile of the next block
        int32_t tiles = 0;
        for (int32_t q = b.h2RowTile0; q < b.h2Src.rowsBelow; q += kTile) {
          const int32_t t = t0 + q / kTile;
          for (int32_t c = t0; c <= t; c++) {
            for (int32_t p = k0; p < k1; p++) touch(t, p, false, st, idx), touch(c, p, false, st, idx);
            apply(t, c, k0, k1, idx);
            touch(t, c, true, st, idx);
            tiles++;
          }
        }
        if (tiles != b.h2Tiles) fail("hand-over tile count", idx);
        break;
      }
      case kDlBulk: {
        for (int64_t q = o.taskBegin; q < o.taskEnd; q++) {
          const UpdTask& ut = plan.updTasks[q];
          const SegDesc& sd = plan.segs[ut.seg];
          const SrcDesc& sr = plan.srcs[sd.src];
          const int64_t rel = sr.off - dl.diagOff;
          const int64_t row0 = rel / n, colS = rel % n;  // first row below the source / its first column
          const int32_t k0 = (int32_t)(colS / kPanelWidth), k1 = k0 + (sr.K + kPanelWidth - 1) / kPanelWidth;
          const int64_t rho = row0 + ut.rowTile;
          const int32_t t = rowTileOf(rho);
          if (sd.kind == kSegBoard) {  // rows below the lump x rows below the lump: sources must be solved
            const int32_t t2 = rowTileOf(row0 + ut.colTile);
            for (int32_t p = k0; p < k1; p++) {
              if (!solved[at(t, p)] || !solved[at(t2, p)]) fail("board update from unsolved rows", idx);
              touch(t, p, false, st, idx);
              touch(t2, p, false, st, idx);
            }
            continue;
          }
          const int32_t c = (int32_t)((row0 + ut.colTile) / kTile);
          if (rho < n && rho % kTile != 0) fail("bulk tile not aligned", idx);
          if (rho >= n && (rho - n) % kTile != 0) fail("bulk tile below the lump not aligned", idx);
          for (int32_t p = k0; p < k1; p++) touch(t, p, false, st, idx), touch(c, p, false, st, idx);
          apply(t, c, k0, k1, idx);
          touch(t, c, true, st, idx);
        }
        break;
      }
      case kDlRecord: {
        Clock c;
        for (int32_t q = 0; q < NS; q++) c.v[q] = q == st ? idx : known[st][q];
        eventClock[o.a] = c;
        eventStream[o.a] = st;
        break;
      }
      case kDlWait: {
        if (eventStream[o.a] < 0) {
          fail("wait for an event that was not recorded", idx);
          break;
        }
        for (int32_t q = 0; q < NS; q++) {
          if (q != st) known[st][q] = std::max(known[st][q], eventClock[o.a].v[q]);
        }
        break;
      }
      default: fail("unknown op", idx);
    }
    if (!err.empty()) return err;
    idx++;
  }
  for (int32_t t = 0; t < TT; t++) {
    for (int32_t k = 0; k < P && (t >= Td || k <= t); k++) {
      if (!solved[at(t, k)]) return "tile never solved " + std::to_string(t) + "," + std::to_string(k);
      if (!allApplied(t, k, k)) return "tile misses a source " + std::to_string(t) + "," + std::to_string(k);
    }
  }
  return "";
}

HipPlanHost buildDenseOpPlan(int64_t n, int64_t k, int64_t offA, bool potrfOnly, int64_t vecOff,
                             int64_t ldaIn) {
  HipPlanHost plan;
  const int64_t ld = ldaIn > 0 ? ldaIn : n;  // row stride of the block and of the rows below it
  BASPACHO_CHECK_GE(ld, n);
  BASPACHO_CHECK_LT(ld, (int64_t)INT32_MAX);
  BASPACHO_CHECK_LT(n + k, (int64_t)INT32_MAX);
  const int64_t rowsB = potrfOnly ? 0 : k;
  for (int64_t blockStart = 0; blockStart < n; blockStart += kOuterWidth) {
    const int64_t blockEnd = std::min<int64_t>(n, blockStart + kOuterWidth);
    for (int64_t c0 = blockStart; c0 < blockEnd; c0 += kPanelWidth) {
      const int32_t nb = (int32_t)std::min<int64_t>(kPanelWidth, blockEnd - c0);
      LevelRange lr{};
      lr.waitDefLevel = -1;
      PanelDesc pd{};
      pd.diagOff = offA + c0 * ld + c0;
      pd.lda = (int32_t)ld;
      pd.nb = nb;
      pd.nRest = (int32_t)(n - c0 - nb);
      pd.rowsBelow = (int32_t)(pd.nRest + rowsB);
      pd.vecOff = (int32_t)(vecOff + c0);
      const int32_t pIdx = (int32_t)plan.panels.size();
      plan.panels.push_back(pd);
      const int32_t rowMin = potrfOnly ? 0 : pd.nRest;  // trsm: only the k rows are touched
      lr.panelBegin = (int64_t)plan.levelPanels.size();
      if (potrfOnly) plan.levelPanels.push_back(pIdx);
      lr.panelEnd = (int64_t)plan.levelPanels.size();
      lr.trsmBegin = (int64_t)plan.trsmTasks.size();
      for (int32_t r = rowMin; r < pd.rowsBelow; r += kTile) plan.trsmTasks.push_back({pIdx, r});
      lr.trsmEnd = (int64_t)plan.trsmTasks.size();
      lr.updBegin = (int64_t)plan.updTasks.size();
      auto addSeg = [&](const SrcDesc& sr, int64_t cols, int64_t tgtBase) {
        plan.srcs.push_back(sr);
        SegDesc sd{};
        sd.src = (int32_t)plan.srcs.size() - 1;
        sd.kind = kSegIntra;
        sd.q0 = 0;
        sd.m = (int32_t)cols;
        sd.tgtBase = tgtBase;
        sd.tgtStride = (int32_t)ld;
        sd.rowMin = potrfOnly ? 0 : sr.nRest;
        plan.segs.push_back(sd);
        const int32_t s = (int32_t)plan.segs.size() - 1;
        for (int32_t cT = 0; cT < sd.m; cT += kTile) {
          for (int32_t rT = cT; rT < sr.rowsBelow; rT += kTile) {
            if (rT + kTile > sd.rowMin) plan.updTasks.push_back({s, rT, cT, 0});
          }
        }
      };
      const int64_t innerCols = blockEnd - c0 - nb;
      if (innerCols > 0 && pd.rowsBelow > 0) {
        SrcDesc sr{};
        sr.off = pd.diagOff + (int64_t)nb * ld;
        sr.lda = (int32_t)ld;
        sr.K = nb;
        sr.rowsBelow = pd.rowsBelow;
        sr.nRest = pd.nRest;
        addSeg(sr, innerCols, offA + (c0 + nb) * ld + (c0 + nb));
      }
      if (c0 + nb == blockEnd && n - blockEnd > 0) {
        SrcDesc sr{};
        sr.off = offA + blockEnd * ld + blockStart;
        sr.lda = (int32_t)ld;
        sr.K = (int32_t)(blockEnd - blockStart);
        sr.nRest = (int32_t)(n - blockEnd);
        sr.rowsBelow = (int32_t)(sr.nRest + rowsB);
        addSeg(sr, n - blockEnd, offA + blockEnd * ld + blockEnd);
      }
      lr.updEnd = (int64_t)plan.updTasks.size();
      lr.defBegin = lr.defMid = lr.defEnd = lr.updEnd;
      plan.levels.push_back(lr);
    }
  }
  return plan;
}

SolveGatherPlan buildSolveGather(const CoalescedBlockMatrixSkel& sk, const HipPlanHost& plan) {
  SolveGatherPlan out;
  const int64_t nSpans = (int64_t)sk.spanStart.size() - 1;
  vector<int64_t> count(nSpans + 1);
  for (const ElimRangePlan& er : plan.elimRanges) {
    const int64_t itemBegin = (int64_t)out.items.size();
    const int64_t entryBase = (int64_t)out.entries.size();
    // counting sort of the below-diagonal chains of the small lumps by row span
    std::fill(count.begin(), count.end(), 0);
    auto forEachChain = [&](auto&& fn) {
      for (int64_t l = er.lumpBegin; l < er.lumpEnd; l++) {
        const int64_t n = sk.lumpStart[l + 1] - sk.lumpStart[l];
        if (n > kElimSmallMax) continue;
        const int64_t c0 = sk.chainColPtr[l], cEnd = sk.chainColPtr[l + 1];
        const int64_t diagCh = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
        for (int64_t c = c0 + diagCh; c < cEnd; c++) fn(l, n, c);
      }
    };
    forEachChain([&](int64_t, int64_t, int64_t c) { count[sk.chainRowSpan[c] + 1]++; });
    for (int64_t s = 0; s < nSpans; s++) count[s + 1] += count[s];
    const int64_t total = count[nSpans];
    BASPACHO_CHECK_LT(entryBase + total, (int64_t)1 << 31);
    out.entries.resize(entryBase + total);
    vector<int64_t> fill(count.begin(), count.end() - 1);
    forEachChain([&](int64_t l, int64_t n, int64_t c) {
      out.entries[entryBase + fill[sk.chainRowSpan[c]]++] = {sk.chainData[c],
                                                             (int32_t)sk.lumpStart[l], (int32_t)n};
    });
    for (int64_t s = 0; s < nSpans; s++) {
      for (int64_t b = count[s]; b < count[s + 1]; b += 256) {
        const int64_t bEnd = std::min(b + 256, count[s + 1]);
        int32_t maxN = 0;
        for (int64_t q = b; q < bEnd; q++) maxN = std::max(maxN, out.entries[entryBase + q].n);
        out.items.push_back({(int32_t)(entryBase + b), (int32_t)(entryBase + bEnd),
                             (int32_t)sk.spanStart[s],
                             (int32_t)(sk.spanStart[s + 1] - sk.spanStart[s]), maxN, 0});
      }
    }
    out.rangeItems.emplace_back(itemBegin, (int64_t)out.items.size());
    // lump-major lists for the backward pass
    out.rangeLumpDesc.push_back((int64_t)out.lumpDescs.size());
    for (int64_t l = er.lumpBegin; l < er.lumpEnd; l++) {
      const int64_t n = sk.lumpStart[l + 1] - sk.lumpStart[l];
      const int64_t c0 = sk.chainColPtr[l], cEnd = sk.chainColPtr[l + 1];
      const int64_t diagCh = sk.boardChainColOrd[sk.boardColPtr[l] + 1];
      SolveLumpDesc d{};
      d.diagOff = sk.chainData[c0];
      d.blockBegin = (int32_t)out.lumpBlocks.size();
      if (n <= kElimSmallMax) {
        for (int64_t c = c0 + diagCh; c < cEnd; c++) {
          const int64_t span = sk.chainRowSpan[c];
          out.lumpBlocks.push_back({sk.chainData[c], (int32_t)sk.spanStart[span],
                                    (int32_t)(sk.spanStart[span + 1] - sk.spanStart[span])});
        }
      }
      d.blockEnd = (int32_t)out.lumpBlocks.size();
      d.xOff = (int32_t)sk.lumpStart[l];
      d.n = (int32_t)n;
      out.lumpDescs.push_back(d);
    }
    BASPACHO_CHECK_LT((int64_t)out.lumpBlocks.size(), (int64_t)1 << 31);
  }
  return out;
}

}  // namespace BaSpaCho
