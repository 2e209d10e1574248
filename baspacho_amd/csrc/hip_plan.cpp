#include "hip_plan.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <atomic>
#include <map>
#include <thread>

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

void buildElimGather(const CoalescedBlockMatrixSkel& sk, HipPlanHost& plan, ElimRangePlan& er,
                     int64_t soleDenseLump) {
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
  // GATHER OVERLAP: group the MFMA items by chunk of target column blocks (stable: row order inside a
  // chunk stays what it was), when every target lies in the one dense lump of the plan
  vector<std::pair<int64_t, int64_t>> orderRanges;  // item ranges (relative to itemBegin) to XCD-order
  {
    const int64_t nLarge = er.itemEnd - er.itemBegin;
    bool chunked = false;
    if (plan.opts.gatherOverlap && soleDenseLump >= 0 && nLarge >= 4096 && er.tinyEnd == er.tinyBegin &&
        er.ldsEnd == er.ldsBegin) {
      const int64_t T = soleDenseLump;
      const int64_t n = sk.lumpStart[T + 1] - sk.lumpStart[T];
      const int64_t nBlk = (n + kOuterWidth - 1) / kOuterWidth;
      const int64_t base = sk.chainData[sk.chainColPtr[T]];  // data offset of the lump column
      const int64_t rowsT = n + lumpCols(sk, T).rowsBelow;
      bool ok = nBlk >= plan.opts.overlapMinBlocks && n * n < (int64_t(1) << 40);
      vector<int32_t> chunkOf((size_t)nBlk, 0);
      int32_t nChunks = 0;
      if (ok) {
        const int32_t first = std::max(1, plan.opts.overlapFirst), step = std::max(1, plan.opts.overlapStep);
        for (int64_t cb = 0; cb < nBlk; cb++) {
          chunkOf[cb] = cb < first ? 0 : 1 + (int32_t)((cb - first) / step);
        }
        nChunks = chunkOf[nBlk - 1] + 1;
        vector<int32_t> itemChunk((size_t)nLarge);
        for (int64_t k = 0; k < nLarge && ok; k++) {
          const ElimGatherItem& it = plan.elimItems[er.itemBegin + k];
          const int64_t rel = it.tgtOff - base;
          if (rel < 0 || it.tgtStride != n || rel >= rowsT * n) {
            ok = false;
            break;
          }
          itemChunk[k] = chunkOf[(rel % n) / kOuterWidth];
        }
        if (ok && nChunks >= 2) {
          vector<ElimGatherItem> tmp(plan.elimItems.begin() + er.itemBegin, plan.elimItems.begin() + er.itemEnd);
          vector<int64_t> tagTmp = itemRowTag;
          vector<int64_t> ptr((size_t)nChunks + 1, 0);
          for (int64_t k = 0; k < nLarge; k++) ptr[itemChunk[k] + 1]++;
          for (int32_t c = 0; c < nChunks; c++) ptr[c + 1] += ptr[c];
          vector<int64_t> cursor(ptr.begin(), ptr.end() - 1);
          for (int64_t k = 0; k < nLarge; k++) {
            const int64_t d = cursor[itemChunk[k]]++;
            plan.elimItems[er.itemBegin + d] = tmp[k];
            itemRowTag[d] = tagTmp[k];
          }
          er.chunkItemPtr.resize((size_t)nChunks + 1);
          for (int32_t c = 0; c <= nChunks; c++) er.chunkItemPtr[c] = er.itemBegin + ptr[c];
          er.chunkOfColBlock = chunkOf;
          er.overlapLump = T;
          for (int32_t c = 0; c < nChunks; c++) orderRanges.emplace_back(ptr[c], ptr[c + 1]);
          chunked = true;
        }
      }
    }
    if (!chunked) orderRanges.emplace_back(0, nLarge);
  }
  // XCD-aware order (speed only).  A workgroup takes 4 consecutive items and
  // workgroup b runs on XCD b % 8, each XCD with its own 4 MB L2.  All items of one target ROW (same
  // sj) read the same B_j source blocks, so a row is handed to ONE XCD (row r -> XCD r % 8, rows
  // balance the load statistically) instead of being sprayed over all eight L2s (measured L2 hit
  // rate 26 %).
  for (const auto& rng : orderRanges) {
  const int64_t g0 = rng.first, g1 = rng.second;
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
  }
  lap("sort + emit items");
}

struct PanelBuild {
  int32_t panel;
  int32_t level;
};

}  // namespace

// (the product's switches arrive through HipBackendOptions; these are developer aids)
void HipPlanOptions::applyDeveloperEnv() {
  planTiming = std::getenv("BSP_TIMING") != nullptr;
  if (const char* e = std::getenv("BSP_GATHER_OVERLAP_FIRST")) overlapFirst = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("BSP_GATHER_OVERLAP_STEP")) overlapStep = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("BSP_SOLVE_SORT_WINDOW")) solveSortWindow = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("BSP_TAIL_MIN_BLOCKS")) tailMinBlocks = std::max(2, std::atoi(e));
  if (const char* e = std::getenv("BSP_TAIL_NARROW_MIN")) tailNarrowMin = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("BSP_TAIL_WHOLE")) tailWholeNarrow = e[0] != '0';
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
  // first column of a lump's persistent tail (-1: none): the last tailBlocks outer blocks of a lump with
  // nothing below it that is at least tailMinBlocks blocks wide; of a NARROW one (tailNarrowMin), all but
  // the first block
  auto tailFromOf = [&](int64_t n, int64_t rowsBelow) -> int64_t {
    const int64_t numBlocks = (n + kOuterWidth - 1) / kOuterWidth;
    if (opts.tailBlocks <= 0 || rowsBelow != 0) return -1;
    if (numBlocks >= opts.tailMinBlocks) {
      // (at most 32 blocks = 128 panels: tile (q, q-2) waits for a word of spine q, whose ticket is one
      //  column group -- at most 128 roles -- later; the 512 roles the GPU holds always include it)
      return kOuterWidth * std::max<int64_t>(1, numBlocks - std::min(opts.tailBlocks, 32));
    }
    if (opts.tailNarrowMin >= 2 && numBlocks >= opts.tailNarrowMin && n - kOuterWidth > 5 * kPanelWidth) {
      return kOuterWidth;  // (at least six panels of tail; one matrix only, see hip_plan.h)
    }
    return -1;
  };
  auto panelsOf = [](int64_t n) -> int32_t {
    int32_t c = 0;
    for (int64_t b = 0; b < n; b += kOuterWidth) {
      c += (int32_t)((std::min<int64_t>(n, b + kOuterWidth) - b + kPanelWidth - 1) / kPanelWidth);
    }
    return c;
  };
  auto addPanels = [&](int64_t l, const LumpCols& g, int32_t lumpRowBase, bool withBoards,
                       const vector<SegDesc>& boardSegTemplates, int tailMode) {
    int32_t count = 0;
    const int64_t n = g.width;
    vector<int64_t> pendingFrom;  // per column block of this lump (lookahead schedule, see below)
    // PERSISTENT TAIL: columns from tailFrom on belong to one launch of hip_tail_kernel.h -- their
    // panels exist (the solves walk them) but carry no segments, and the block before them hands ALL
    // its pending lookahead units over at once
    // (tailMode 2: a narrow root lump that follows other levels -- the WHOLE lump is the tail)
    const int64_t tailFrom = tailMode == 2 ? 0 : (tailMode == 1 ? tailFromOf(n, g.rowsBelow) : -1);
    if (tailFrom >= 0) plan.hasTail = true;
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
        pd.pad = (tailFrom >= 0 && c0 >= tailFrom) ? 1 : 0;
        plan.panels.push_back(pd);
        plan.potrfFlops += double(nb) * nb * nb / 3.0;
        plan.trsmFlops += double(pd.rowsBelow) * nb * nb;
        panelSegBegin.push_back((int64_t)plan.segs.size());
        if (pd.pad) {  // inside the tail: no segments
          panelSegEnd.push_back((int64_t)plan.segs.size());
          const double m = double(n - c0 - nb);
          plan.tailUpdFlops += 2.0 * nb * (m * (m + 1) / 2);
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
              double budgetUs = bulkAhead * (118.0 + 0.012 * double(sr.rowsBelow));
              // (topping the first launch up to a full round of workgroups with the nearest
              //  optional targets was measured slower: the execution stream waits on it)
              int64_t c = b + 2;
              if (tailFrom >= 0 && blockEnd == tailFrom) {
                // the block before a persistent tail: everything still pending, as due units
                for (; c < numBlocks; c++) pushUnit(c, 2);
              }
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
        int32_t n = addPanels(l, g, rowBase, /*withBoards=*/false, {}, /*tailMode=*/0);
        for (int32_t j = 0; j < n; j++) bucketAt(big, j).push_back({first + j, j});
      }
    }
    elimBigBuckets.push_back(std::move(big));
    {
      // (gather overlap: the dense part of this plan is exactly one lump, and it is the last range)
      const int64_t denseBegin0 = std::max(startLump, denseFrom);
      const bool lastRange = re == denseFrom;
      buildElimGather(sk, plan, er, (lastRange && upToLump - denseBegin0 == 1) ? denseBegin0 : -1);
    }
    plan.elimRanges.push_back(std::move(er));
  }
  // ---- dense lumps
  const int64_t denseBegin = std::max(startLump, denseFrom);
  vector<int32_t> lastLevelOfLump(nLumps, -1);
  // A tail level is ONE launch that factors nothing but the tail (emitLevels): a lump may only hand
  // columns to the tail if each of those panels is ALONE in its level.  Forests with several wide roots
  // and deep side branches (MERI, block-diagonal problems) put other lumps' panels on the same levels;
  // levels are known before any panel is built, so: one dry pass over the level numbers.
  vector<char> tailOk(nLumps, 0);
  {
    vector<int32_t> lastLv(nLumps, -1), firstLv(nLumps, 0), occupancy;
    for (int64_t l = denseBegin; l < upToLump; l++) {
      int32_t level = 0;
      for (int64_t q = sk.boardRowPtr[l]; q < sk.boardRowPtr[l + 1] - 1; q++) {
        const int64_t s = sk.boardColLump[q];
        if (s >= denseBegin && s < l) level = std::max(level, lastLv[s] + 1);
      }
      const int32_t n = panelsOf(sk.lumpStart[l + 1] - sk.lumpStart[l]);
      firstLv[l] = level;
      lastLv[l] = level + n - 1;
      if ((int64_t)occupancy.size() < level + n) occupancy.resize(level + n, 0);
      for (int32_t j = 0; j < n; j++) occupancy[level + j]++;
    }
    for (int64_t l = denseBegin; l < upToLump; l++) {
      const LumpCols g = lumpCols(sk, l);
      const int64_t tf = tailFromOf(g.width, g.rowsBelow);
      if (tf < 0) continue;
      bool alone = true;
      for (int32_t j = (int32_t)(tf / kOuterWidth) * (kOuterWidth / kPanelWidth); j <= lastLv[l] - firstLv[l]; j++) {
        alone = alone && occupancy[firstLv[l] + j] == 1;
      }
      tailOk[l] = alone;
      // narrow root lump with levels before it: its first outer block too, when those panels are alone as
      // well (the level before the tail exists: it takes the flushDue mark; GRID 82x82: 16 instead of 12 panels)
      if (alone && opts.tailWholeNarrow && firstLv[l] > 0 &&
          (g.width + kOuterWidth - 1) / kOuterWidth < opts.tailMinBlocks) {
        bool all = true;
        for (int32_t j = 0; j <= lastLv[l] - firstLv[l]; j++) all = all && occupancy[firstLv[l] + j] == 1;
        if (all) tailOk[l] = 2;
      }
    }
  }
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
    const int32_t n = addPanels(l, g, lumpRowBase, /*withBoards=*/true, boardSegs, (int)tailOk[l]);
    for (int32_t j = 0; j < n; j++) bucketAt(levelBuckets, level + j).push_back({first + j, level + j});
    lastLevelOfLump[l] = level + n - 1;
  }

  // ---- emit task lists level by level
  auto emitLevels = [&](const vector<vector<PanelBuild>>& buckets, vector<LevelRange>& out) {
    std::map<int32_t, int64_t> lastDeferredLevel;  // lump -> level index that deferred tiles
    std::map<int32_t, std::vector<int64_t>> blockForks;  // lump -> its block-boundary fork levels
    int64_t tailHead = -1;
    for (const auto& bucket : buckets) {
      LevelRange lr;
      lr.panelBegin = (int64_t)plan.levelPanels.size();
      lr.trsmBegin = (int64_t)plan.trsmTasks.size();
      lr.updBegin = (int64_t)plan.updTasks.size();
      lr.waitDefLevel = -1;
      const int64_t levelIdx = (int64_t)out.size();
      // single-panel level followed by a single-panel level of the same lump?
      const size_t bi = (size_t)(&bucket - &buckets[0]);
      // (a persistent tail takes the chain over: the level before it fuses / stages nothing for it)
      const bool chain = bucket.size() == 1 && bi + 1 < buckets.size() &&
                         buckets[bi + 1].size() == 1 &&
                         plan.panels[buckets[bi + 1][0].panel].lump == plan.panels[bucket[0].panel].lump &&
                         plan.panels[buckets[bi + 1][0].panel].pad == 0;
      for (const auto& pb : bucket) {  // (tailOk above: a tail panel never shares its level)
        BASPACHO_CHECK(bucket.size() == 1 || plan.panels[pb.panel].pad == 0);
      }
      if (bucket.size() == 1 && plan.panels[bucket[0].panel].pad == 1) {
        const bool first = out.empty() || out.back().tail == 0;
        // (the panel and its row tiles stay listed: the solves walk a tail level like any other
        //  one-panel level; factor() skips it)
        const PanelDesc& tp = plan.panels[bucket[0].panel];
        plan.levelPanels.push_back(bucket[0].panel);
        for (int32_t r = 0; r < tp.rowsBelow; r += kTile) plan.trsmTasks.push_back({bucket[0].panel, r});
        lr.panelEnd = (int64_t)plan.levelPanels.size();
        lr.trsmEnd = (int64_t)plan.trsmTasks.size();
        lr.updEnd = lr.defBegin = lr.defMid = lr.defEnd = lr.updBegin;
        lr.directPanel = bucket[0].panel;
        lr.tail = first ? 1 : 2;
        if (first) {
          plan.numLaunches += 1;
          tailHead = (int64_t)out.size();
        }
        out.push_back(lr);
        out[tailHead].tailPanels++;
        continue;
      }
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
    for (size_t li = 1; li < out.size(); li++) {
      if (out[li].tail == 1) out[li - 1].flushDue = 1;
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
  // gather overlap: which chunk every dense launch has to wait for (every level must be a one-panel
  // level of the target lump; otherwise the chunks simply run one after the other before the dense part)
  for (ElimRangePlan& er : plan.elimRanges) {
    if (er.chunkItemPtr.empty()) continue;
    bool ok = !plan.levels.empty();
    for (const LevelRange& lr : plan.levels) {
      ok = ok && lr.directPanel >= 0 && plan.panels[lr.directPanel].lump == er.overlapLump;
    }
    if (!ok) {
      er.overlapLump = -1;
      continue;
    }
    const int64_t nBlk = (int64_t)er.chunkOfColBlock.size();
    for (LevelRange& lr : plan.levels) {
      const PanelDesc& pd = plan.panels[lr.directPanel];
      const int64_t b = (pd.lda - pd.nRest - pd.nb) / kOuterWidth;
      lr.gatherNow = er.chunkOfColBlock[std::min(b + 1, nBlk - 1)];
      auto maxChunk = [&](int64_t begin, int64_t end) {
        int32_t m = -1;
        for (int64_t t = begin; t < end; t++) {
          // (lookahead units: tile columns count from the end of the source block)
          const int64_t cb = std::min(nBlk - 1, (kOuterWidth * (b + 1) + plan.updTasks[t].colTile) / kOuterWidth);
          m = std::max(m, er.chunkOfColBlock[cb]);
        }
        return m;
      };
      lr.gatherDue = maxChunk(lr.defBegin, lr.defMid);
      lr.gatherOpt = maxChunk(lr.defMid, lr.defEnd);
    }
  }

  // board segments carried their target lump in tgtBase only for the atomic analysis
  for (auto& s : plan.segs) {
    if (s.kind == kSegBoard) s.tgtBase = 0;
  }
  BASPACHO_CHECK_LT((int64_t)plan.segs.size(), (int64_t)INT32_MAX);
  return plan;
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
    int64_t commonN = -1, spanBelow = nSpans;
    for (int64_t l = er.lumpBegin; l < er.lumpEnd; l++) {
      const int64_t n = sk.lumpStart[l + 1] - sk.lumpStart[l];
      commonN = (commonN < 0 || commonN == n) ? n : 0;
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
          spanBelow = std::min(spanBelow, span);
        }
      }
      d.blockEnd = (int32_t)out.lumpBlocks.size();
      d.xOff = (int32_t)sk.lumpStart[l];
      d.n = (int32_t)n;
      out.lumpDescs.push_back(d);
    }
    out.rangeWide.push_back({(int32_t)(commonN >= 1 && commonN <= 4 ? commonN : 0), spanBelow,
                             sk.spanStart[spanBelow]});
    {
      // The backward kernels give a lump 16 lanes and a wave four lumps: a wave runs as long as its
      // longest lump, and track lengths have a heavy tail (BAL-871: mean 5.6 blocks, 6 % of the points
      // beyond 20; PMC: twice the wave loads the blocks need).  Nothing depends on the order of the
      // descriptors, so the 16 lumps of a workgroup are sorted by block count: its waves get lumps of
      // similar length, the workgroup still reads one contiguous piece of L.
      // (tried: windows of 1024 lumps -- solveLt 0.56 -> 0.73 ms with one right-hand side, 1.49 -> 2.21
      //  with ten: the lumps of a wave then sit 100 KB apart and share neither lines nor rows)
      const int64_t kWindow = plan.opts.solveSortWindow;
      auto first = out.lumpDescs.begin() + out.rangeLumpDesc.back();
      const int64_t cnt = out.lumpDescs.end() - first;
      for (int64_t w0 = 0; kWindow > 1 && w0 < cnt; w0 += kWindow) {
        std::stable_sort(first + w0, first + std::min(cnt, w0 + kWindow),
                         [](const SolveLumpDesc& a, const SolveLumpDesc& b) {
                           return a.blockEnd - a.blockBegin > b.blockEnd - b.blockBegin;
                         });
      }
    }
    BASPACHO_CHECK_LT((int64_t)out.lumpBlocks.size(), (int64_t)1 << 31);
  }
  return out;
}

}  // namespace BaSpaCho
