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

// overlapLump >= 0: the caller would like to overlap this range's update with the factorization of
// that (single, wide) dense lump; granted -- er.overlapLump set, items grouped by target column
// block -- when every target lies in it and every item is an MFMA item.
void buildElimGather(const CoalescedBlockMatrixSkel& sk, HipPlanHost& plan, ElimRangePlan& er,
                     int64_t overlapLump = -1) {
  er.useGather = false;
  er.overlapLump = -1;
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
  // PACKED OPERANDS (ElimRangePlan::packRows): lumps of width <= 4 (the descriptor-driven factor
  // kernel writes the copy), every below block with the same number of rows h, 16 < h * h and
  // h <= 16 (every item is then an MFMA item), h * width <= one slot.
  er.packRows = 0;
  std::vector<int32_t> packSlotOfLump;
  {
    const bool want = plan.opts.elimPack && !plan.opts.gatherRowForm;
    bool ok = want && er.lumpEnd > er.lumpBegin;
    int64_t h = -1, slots = 0;
    for (int64_t l = er.lumpBegin; ok && l < er.lumpEnd; l++) {
      LumpCols g = lumpCols(sk, l);
      ok = g.width <= 4;
      packSlotOfLump.push_back((int32_t)slots);
      for (int64_t c = g.chain0 + g.diagChains; ok && c < g.chain0 + g.nChains; c++) {
        const int64_t sp = sk.chainRowSpan[c];
        const int64_t rows = sk.spanStart[sp + 1] - sk.spanStart[sp];
        if (h < 0) h = rows;
        ok = rows == h && h * g.width <= kElimPackSlot;
        slots++;
      }
    }
    ok = ok && h > 0 && h <= 16 && h * h > 16 && slots * kElimPackSlot < (int64_t(1) << 31);
    if (ok) {
      er.packRows = (int32_t)h;
      er.packSlots = slots;
    } else {
      packSlotOfLump.clear();
    }
  }
  // (every exit that leaves the range without gather items must leave it unpacked too)
  struct Unpack {
    ElimRangePlan& er;
    ~Unpack() {
      if (!er.useGather || er.useRowForm) er.packRows = 0;
    }
  } unpackOnFailure{er};
  // enumerate(f): f(targetChain, si, width, offJ, offI) for every pair; false if unsupported
  auto enumerate = [&](auto&& f) -> bool {
    for (int64_t l = er.lumpBegin; l < er.lumpEnd; l++) {
      LumpCols g = lumpCols(sk, l);
      if (g.width > 255) return false;
      const int64_t cBegin = g.chain0 + g.diagChains, cEnd = g.chain0 + g.nChains;
      const int64_t slot0 = er.packRows ? packSlotOfLump[l - er.lumpBegin] : 0;
      auto srcOff = [&](int64_t c) {
        return er.packRows ? (slot0 + (c - cBegin)) * kElimPackSlot : sk.chainData[c];
      };
      for (int64_t i = cBegin; i < cEnd; i++) {
        const int64_t si = sk.chainRowSpan[i];
        const int64_t siSize = sk.spanStart[si + 1] - sk.spanStart[si];
        mapTarget(sk.spanToLump[si]);
        if (sk.spanToLump[si] != overlapLump) overlapLump = -1;
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
  // ---- ROW FORM (opt-in, BSP_GATHER_ROW_FORM=1; every block must fit a 16x16 MFMA tile): one
  //      workgroup per row of targets, pairs ordered by source column.  Measured on BAL-871:
  //      2.2 ms against 1.76 ms for the item form below -- it fetches 21 % fewer bytes (5.5 GB
  //      instead of 7.0 GB) but back-to-back loads of neighbouring blocks each miss on the cache
  //      line they share, and 16 waves per CU (LDS is full of accumulators) hide less latency
  //      than the 32 of the item form.
  const bool rowFormEnabled = plan.opts.gatherRowForm;
  if (rowFormEnabled) {
    bool ok = true;
    vector<ElimRowItem> rowItems;
    vector<ElimRowSlot> rowSlots;
    vector<uint32_t> oj, oi;
    vector<uint16_t> sl;
    oj.reserve((size_t)nPairs);
    oi.reserve((size_t)nPairs);
    sl.reserve((size_t)nPairs);
    double targetElems = 0;
    int32_t maxLds = 0, maxSlots = 0;
    vector<int32_t> sis;
    sortBuckets(sorted, bucketPtr, [](const Pair& x, const Pair& y) {
      return x.offJ != y.offJ ? x.offJ < y.offJ : x.si < y.si;
    });
    for (int64_t c = 0; c < nChainsTot && ok; c++) {
      const int64_t b = bucketPtr[c], e = bucketPtr[c + 1];
      if (b == e) continue;
      const int64_t sj = sk.chainRowSpan[c];
      const int64_t rows = sk.spanStart[sj + 1] - sk.spanStart[sj];
      const int32_t width = sorted[b].width;
      if (rows > 16 || width > 16) {
        ok = false;
        break;
      }
      sis.clear();
      for (int64_t k = b; k < e; k++) sis.push_back(sorted[k].si);
      std::sort(sis.begin(), sis.end());
      sis.erase(std::unique(sis.begin(), sis.end()), sis.end());
      if (sis.size() > 60000) ok = false;
      // a row whose accumulators do not fit LDS is cut into parts by target column (disjoint
      // targets, each part with the pairs of its own targets only)
      vector<int32_t> partOfSlot(sis.size()), slotInPart(sis.size());
      vector<int32_t> partFirstSlot;  // index into sis of every part's first slot
      {
        int32_t used = 0;
        for (size_t q = 0; q < sis.size(); q++) {
          const int64_t cols = sk.spanStart[sis[q] + 1] - sk.spanStart[sis[q]];
          if (cols > 16) ok = false;
          const int32_t need = (int32_t)(rows * cols);
          if (partFirstSlot.empty() || used + need > kRowFormMaxLdsElems) {
            partFirstSlot.push_back((int32_t)q);
            used = 0;
          }
          partOfSlot[q] = (int32_t)partFirstSlot.size() - 1;
          slotInPart[q] = (int32_t)q - partFirstSlot.back();
          used += need;
        }
      }
      const size_t nParts = partFirstSlot.size();
      vector<vector<int64_t>> partPairs(nParts);
      for (int64_t k = b; k < e && ok; k++) {
        if (sorted[k].width != width) ok = false;
        const size_t q = std::lower_bound(sis.begin(), sis.end(), sorted[k].si) - sis.begin();
        partPairs[partOfSlot[q]].push_back(k);
      }
      for (size_t part = 0; part < nParts && ok; part++) {
        const size_t q0 = partFirstSlot[part];
        const size_t q1 = part + 1 < nParts ? (size_t)partFirstSlot[part + 1] : sis.size();
        ElimRowItem it{};
        it.pairBegin = (int32_t)(plan.elimPairOffJ.size() + oj.size());
        it.slotBegin = (int32_t)(plan.elimRowSlots.size() + rowSlots.size());
        int32_t ldsOff = 0;
        for (size_t q = q0; q < q1; q++) {
          const int32_t si = sis[q];
          const int64_t cols = sk.spanStart[si + 1] - sk.spanStart[si];
          const int64_t t = sk.spanToLump[si];
          ElimRowSlot sd{};
          sd.tgtOff = sk.chainData[c] + sk.spanOffsetInLump[si];
          sd.tgtStride = (int32_t)(sk.lumpStart[t + 1] - sk.lumpStart[t]);
          sd.ldsOff = ldsOff;
          sd.cols = (int16_t)cols;
          sd.flags = (int16_t)(sj == si ? 2 : 0);
          rowSlots.push_back(sd);
          ldsOff += (int32_t)(rows * cols);
          targetElems += sj == si ? double(rows) * (rows + 1) / 2 : double(rows) * cols;
        }
        for (int64_t k : partPairs[part]) {
          const size_t q = std::lower_bound(sis.begin(), sis.end(), sorted[k].si) - sis.begin();
          oj.push_back(sorted[k].offJ);
          oi.push_back(sorted[k].offI);
          sl.push_back((uint16_t)slotInPart[q]);
        }
        it.pairEnd = (int32_t)(plan.elimPairOffJ.size() + oj.size());
        it.slotEnd = (int32_t)(plan.elimRowSlots.size() + rowSlots.size());
        it.ldsElems = ldsOff;
        it.rows = (int16_t)rows;
        it.n = (int16_t)width;
        // long rows are cut by source-column range so that no workgroup runs much longer than
        // the others (the longest row of BAL-871 holds 1.5x the per-CU average of the pairs)
        const int32_t nPairsIt = it.pairEnd - it.pairBegin;
        const int32_t cuts = (nPairsIt + kRowFormMaxPairs - 1) / kRowFormMaxPairs;
        if (cuts > 1) {
          const int32_t first = it.pairBegin;
          for (int32_t q = 0; q < cuts; q++) {
            ElimRowItem piece = it;
            piece.pairBegin = first + (int32_t)((int64_t)nPairsIt * q / cuts);
            piece.pairEnd = first + (int32_t)((int64_t)nPairsIt * (q + 1) / cuts);
            piece.shared = 1;
            rowItems.push_back(piece);
          }
        } else {
          rowItems.push_back(it);
        }
        maxLds = std::max(maxLds, ldsOff);
        maxSlots = std::max(maxSlots, (int32_t)(q1 - q0));
      }
    }
    if (ok) {
      // longest rows first: with one workgroup per CU the tail of the launch is the last rows
      std::stable_sort(rowItems.begin(), rowItems.end(), [](const ElimRowItem& x, const ElimRowItem& y) {
        return x.pairEnd - x.pairBegin > y.pairEnd - y.pairBegin;
      });
      // slotIdx stream is padded to the pair streams (they are shared with the item form)
      plan.elimPairSlot.resize(plan.elimPairOffJ.size(), 0);
      plan.elimPairOffJ.insert(plan.elimPairOffJ.end(), oj.begin(), oj.end());
      plan.elimPairOffI.insert(plan.elimPairOffI.end(), oi.begin(), oi.end());
      plan.elimPairSlot.insert(plan.elimPairSlot.end(), sl.begin(), sl.end());
      er.rowBegin = (int64_t)plan.elimRows.size();
      plan.elimRows.insert(plan.elimRows.end(), rowItems.begin(), rowItems.end());
      er.rowEnd = (int64_t)plan.elimRows.size();
      plan.elimRowSlots.insert(plan.elimRowSlots.end(), rowSlots.begin(), rowSlots.end());
      const int32_t accElems = (maxLds + 3) & ~3;
      er.rowLdsBytes = accElems * 8 + maxSlots * (int32_t)sizeof(ElimRowSlot);
      er.rowLdsBytesF32 = accElems * 4 + maxSlots * (int32_t)sizeof(ElimRowSlot);
      er.useRowForm = true;
      er.useGather = true;
      er.itemBegin = er.itemEnd = er.tinyBegin = er.tiny9End = er.tinyEnd = er.ldsBegin = er.ldsEnd =
          (int64_t)plan.elimItems.size();
      plan.elimTargetElems += targetElems;
      lap("row form");
      return;
    }
  }
  plan.elimPairOffJ.reserve(plan.elimPairOffJ.size() + (size_t)nPairs);
  plan.elimPairOffI.reserve(plan.elimPairOffI.size() + (size_t)nPairs);
  er.itemBegin = (int64_t)plan.elimItems.size();
  const int64_t maxPairs = std::max<int64_t>(8, plan.opts.gatherMaxPairs);
  vector<int64_t> itemRowTag;  // target chain of every emitted item
  vector<int32_t> itemChunk;   // source-data chunk of every emitted item
  vector<int32_t> itemColBlock;  // outer block of the target column inside its lump
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
        itemChunk.push_back((int32_t)chunkOfU);
        itemColBlock.push_back((int32_t)(sk.spanOffsetInLump[si] / kOuterWidth));
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
  if (er.packRows) {
    er.packSlotOff = (int64_t)plan.elimPackSlot.size();
    plan.elimPackSlot.insert(plan.elimPackSlot.end(), packSlotOfLump.begin(), packSlotOfLump.end());
  }
  // items whose target block has at most 16 elements (e.g. 3x3 blocks of automatically detected
  // ranges) go to the kernel that packs four items per wave: move them behind the others
  {
    vector<ElimGatherItem> large, tiny, tiny9, wide;
    vector<int64_t> tagL;
    vector<int32_t> cbL;
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
        cbL.push_back(itemColBlock[k - er.itemBegin]);
      }
    }
    auto dst = plan.elimItems.begin() + er.itemBegin;
    dst = std::copy(large.begin(), large.end(), dst);
    dst = std::copy(tiny9.begin(), tiny9.end(), dst);
    dst = std::copy(tiny.begin(), tiny.end(), dst);
    std::copy(wide.begin(), wide.end(), dst);
    itemRowTag = tagL;
    itemColBlock = cbL;
    itemChunk.assign(large.size(), 0);
    er.itemEnd = er.itemBegin + (int64_t)large.size();
    er.tinyBegin = er.itemEnd;
    er.tiny9End = er.tinyBegin + (int64_t)tiny9.size();
    er.tinyEnd = er.tiny9End + (int64_t)tiny.size();
    er.ldsBegin = er.tinyEnd;
    er.ldsEnd = er.ldsBegin + (int64_t)wide.size();
  }
  // OVERLAP groups: items by outer block of the target column, a few blocks per group -- early
  // groups small (the dense chain waits for the first one), later ones larger
  vector<int64_t> groupBounds = {0, er.itemEnd - er.itemBegin};  // relative item indices
  if (overlapLump >= 0 && er.tinyEnd == er.tinyBegin && er.ldsEnd == er.ldsBegin &&
      er.itemEnd - er.itemBegin >= 4096) {
    const int64_t nItems = er.itemEnd - er.itemBegin;
    const int64_t width = sk.lumpStart[overlapLump + 1] - sk.lumpStart[overlapLump];
    const int32_t numBlocks = (int32_t)((width + kOuterWidth - 1) / kOuterWidth);
    vector<int32_t> colBounds = {0};
    for (int32_t step = 2; colBounds.back() < numBlocks; step += (colBounds.size() % 2 == 0)) {
      colBounds.push_back(std::min<int32_t>(numBlocks, colBounds.back() + step));
    }
    auto groupOf = [&](int32_t cb) {
      return (int32_t)(std::upper_bound(colBounds.begin(), colBounds.end(), cb) - colBounds.begin()) - 1;
    };
    vector<int64_t> order(nItems);
    for (int64_t k = 0; k < nItems; k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
      return groupOf(itemColBlock[x]) < groupOf(itemColBlock[y]);
    });
    vector<ElimGatherItem> tmp(plan.elimItems.begin() + er.itemBegin, plan.elimItems.begin() + er.itemEnd);
    vector<int64_t> tag2(nItems);
    const int32_t nGroups = (int32_t)colBounds.size() - 1;
    groupBounds.assign(nGroups + 1, nItems);
    er.groupPairs.assign(nGroups, 0.0);
    int32_t cur = -1;
    for (int64_t k = 0; k < nItems; k++) {
      const int32_t gq = groupOf(itemColBlock[order[k]]);
      while (cur < gq) groupBounds[++cur] = k;
      plan.elimItems[er.itemBegin + k] = tmp[order[k]];
      tag2[k] = itemRowTag[order[k]];
      er.groupPairs[gq] += double(tmp[order[k]].pairEnd - tmp[order[k]].pairBegin);
    }
    while (cur < nGroups) groupBounds[++cur] = nItems;
    itemRowTag = tag2;
    er.overlapLump = overlapLump;
    er.groupColBlock = colBounds;
    er.groupItem.resize(nGroups + 1);
    for (int32_t q = 0; q <= nGroups; q++) er.groupItem[q] = er.itemBegin + groupBounds[q];
  }
  // XCD-aware order (speed only), inside every group.  A workgroup takes 4 consecutive items and
  // workgroup b runs on XCD b % 8, each XCD with its own 4 MB L2.  All items of one target ROW (same
  // sj) read the same B_j source blocks, so a row is handed to ONE XCD (row r -> XCD r % 8, rows
  // balance the load statistically) instead of being sprayed over all eight L2s (measured L2 hit
  // rate 26 %).
  for (size_t gq = 0; gq + 1 < groupBounds.size(); gq++) {
    const int64_t g0 = groupBounds[gq], g1 = groupBounds[gq + 1];
    const int64_t nItems = g1 - g0;
    if (nItems < 512) continue;
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
    if (plan.opts.gatherReverse) {
      for (auto& v : perXcd) std::reverse(v.begin(), v.end());
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

}  // namespace

HipPlanOptions HipPlanOptions::fromEnv() {
  HipPlanOptions o;
  auto on = [](const char* name, bool dflt) {
    const char* e = std::getenv(name);
    return e ? e[0] != '0' : dflt;
  };
  auto optIn = [](const char* name) {
    const char* e = std::getenv(name);
    return e && e[0] == '1';
  };
  o.dueStream = on("BSP_DUE_STREAM", true);
  o.earlyDue = optIn("BSP_EARLY_DUE") && o.dueStream;
  o.dueSplit = optIn("BSP_DUE_SPLIT") && o.dueStream;
  o.bulkRowMajor = on("BSP_BULK_ROW_MAJOR", true);
  o.elimPack = optIn("BSP_ELIM_PACK");
  o.gatherRowForm = optIn("BSP_GATHER_ROW_FORM");
  o.elimOverlap = optIn("BSP_ELIM_OVERLAP");
  o.planTiming = std::getenv("BSP_TIMING") != nullptr;
  o.dropElimUpdate = optIn("BSP_FAULT_DROP_ELIM_UPDATE");
  o.nowSplit = optIn("BSP_NOW_SPLIT");
  o.chainWindow = optIn("BSP_CHAIN_WINDOW");
  o.gatherReverse = optIn("BSP_GATHER_REVERSE");
  if (o.chainWindow) o.nowSplit = false;
  if (const char* e = std::getenv("BSP_GATHER_MAX_PAIRS")) o.gatherMaxPairs = std::max(8, atoi(e));
  if (const char* e = std::getenv("BSP_BULK_AHEAD")) o.bulkAhead = std::atof(e);
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
  // overlapped elimination (set after the elimination ranges are planned, used by addPanels)
  const ElimRangePlan* ov = nullptr;
  vector<int64_t> optionalLimit;  // per outer block b of the overlap lump

  // ---- helper: cut a lump into outer blocks and panels; returns number of panels.
  // Segments of a panel: the remaining columns of its outer block (source = the panel, K = nb);
  // the last panel of an outer block also carries the segments of the block-wide source
  // (K = block width): rest of the lump and, if requested, every board of the lump column.
  auto addPanels = [&](int64_t l, const LumpCols& g, int32_t lumpRowBase, bool withBoards,
                       const vector<SegDesc>& boardSegTemplates) {
    int32_t count = 0;
    const int64_t n = g.width;
    vector<int64_t> pendingFrom;  // per column block of this lump (lookahead schedule, see below)
    // (opt-in: measured 7.26-7.40 ms against 7.15-7.21 on BAL-871 -- the K = 192 + 64 split
    //  costs the side streams more than the earlier start gives back)
    const bool earlyDue = opts.earlyDue;
    for (int64_t blockStart = 0; blockStart < n; blockStart += kOuterWidth) {
      const int64_t blockEnd = std::min<int64_t>(n, blockStart + kOuterWidth);
      int64_t earlyDueCols = 0;  // leading columns of this block whose due unit (c = b + 2) went early
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
        // CHAIN WINDOW (opts.chainWindow): a panel of a lump with further outer blocks updates, with
        // its own rank nb, not only the rest of its block but the whole NEXT block as well, and the
        // block-wide rank-256 "now" update (390 tiles with 256 source columns from memory on the
        // execution stream, ~50 us each beside a saturated bulk stream) does not exist: every launch
        // of the chain is made of light tiles.  The next block's columns are shared with the
        // lookahead units that are due at the end of this block: both sides use atomics there.
        const bool window = opts.chainWindow && n > blockEnd && nb == kPanelWidth &&
                            blockEnd - blockStart == kOuterWidth;
        const int64_t ownCols = blockEnd - c0 - nb;
        const int64_t innerCols = ownCols + (window ? std::min<int64_t>(kOuterWidth, n - blockEnd) : 0);
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
          if (window) {
            s.lump = (int32_t)l;
            s.firstChainOrd = (int32_t)ownCols;
            s.pad = 4 | (ownCols == 0 ? 8 : 0);
          }
          plan.segs.push_back(s);
        }
        // EARLY DUE: the due unit of this block (to column block b + 2) is what the chain waits for
        // at the end of the NEXT block, and it is launched beside a saturated GPU.  Its first 192
        // source columns are final one chain step before the block is complete: they go now, and
        // only the last panel's 64 columns are left for the block boundary (both accumulate with
        // atomics; same stream, in order).
        if (earlyDue && blockEnd - blockStart == kOuterWidth && c0 + nb == blockEnd - kPanelWidth &&
            nb == kPanelWidth) {
          const int64_t b = blockStart / kOuterWidth;
          const int64_t numBlocks = (n + kOuterWidth - 1) / kOuterWidth;
          const int64_t c = b + 2;
          if (c < numBlocks && (int64_t)pendingFrom.size() == numBlocks && pendingFrom[c] == b) {
            SrcDesc fs{};
            fs.off = g.diagOff + blockEnd * n + blockStart;
            fs.lda = (int32_t)n;
            fs.K = (int32_t)(kOuterWidth - kPanelWidth);
            fs.nRest = (int32_t)(n - blockEnd);
            fs.rowsBelow = (int32_t)(fs.nRest + g.rowsBelow);
            fs.lumpRowBase = lumpRowBase;
            plan.srcs.push_back(fs);
            SegDesc u{};
            u.src = (int32_t)plan.srcs.size() - 1;
            u.kind = kSegIntra;
            u.outer = 4;
            u.lump = (int32_t)l;
            u.q0 = (int32_t)(c * kOuterWidth - blockEnd);
            u.m = (int32_t)std::min<int64_t>(kOuterWidth, n - c * kOuterWidth);
            u.tgtBase = g.diagOff + blockEnd * n + blockEnd;
            u.tgtStride = (int32_t)n;
            u.pad = 2;
            plan.segs.push_back(u);
            plan.segColBlock.resize(plan.segs.size(), -1);
            plan.segColBlock.back() = (int32_t)c;
            earlyDueCols = fs.K;
          }
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
              // (CHAIN WINDOW: the block's four panels have each applied their rank 64 to these
              //  columns already; `s` only serves as the template of the lookahead units below)
              const bool windowBlock = opts.chainWindow && blockEnd - blockStart == kOuterWidth;
              if (!windowBlock) plan.segs.push_back(s);
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
                // (CHAIN WINDOW: the chain's own tiles reach column block c from block c - 1's first
                //  step on, i.e. they meet the due units and the optional units to c = b + 3
                //  whatever the streams are: plain bit 0, honoured by every launch mode)
                const int32_t multi = (pendingFrom[c] < b ? 1 : 0) |
                                      ((dueStream && (outerKind == 2 || c == b + 3)) ? 2 : 0) |
                                      ((windowBlock && (outerKind == 2 || c == b + 3)) ? 1 : 0);
                for (int64_t sb = pendingFrom[c]; sb <= b; sb++) {
                  SrcDesc fs = sr;
                  fs.off = g.diagOff + blockEnd * n + sb * kOuterWidth;
                  fs.K = (int32_t)(std::min<int64_t>(blockEnd, (sb + 1) * kOuterWidth) - sb * kOuterWidth);
                  if (sb == b && c == b + 2 && earlyDueCols > 0) {
                    // (the block's leading columns went ahead one chain step earlier: EARLY DUE)
                    fs.off += earlyDueCols;
                    fs.K -= (int32_t)earlyDueCols;
                  }
                  plan.srcs.push_back(fs);
                  SegDesc u = s;
                  u.src = (int32_t)plan.srcs.size() - 1;
                  u.outer = outerKind;
                  u.q0 = (int32_t)(c * kOuterWidth - blockEnd);
                  u.m = (int32_t)std::min<int64_t>(kOuterWidth, n - c * kOuterWidth);
                  u.pad = multi;
                  plan.segs.push_back(u);
                  plan.segColBlock.resize(plan.segs.size(), -1);
                  plan.segColBlock.back() = (int32_t)c;
                }
                pendingFrom[c] = b + 1;
              };
              double budgetUs = bulkAhead * (118.0 + 0.012 * double(sr.rowsBelow));
              // (topping the first launch up to a full round of workgroups with the nearest
              //  optional targets was measured slower: the execution stream waits on it)
              int64_t c = b + 2;
              if (c < numBlocks) {
                budgetUs -= unitFlops(c) / kBulkFlopsPerUs;
                pushUnit(c++, 2);
              }
              const int64_t cLimit = (ov && l == ov->overlapLump && b < (int64_t)optionalLimit.size())
                                         ? optionalLimit[b] : numBlocks;
              for (; c < numBlocks && budgetUs > 0 && c <= cLimit; c++) {
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
    // candidate for overlapping this range's update with the dense phase: it is the last range
    // before the dense part, which is ONE wide lump inside the planned interval, and the range has
    // no wide lumps of its own (buildElimGather checks that every target lies in that lump)
    int64_t overlapLump = -1;
    {
      const int64_t denseBegin0 = std::max(startLump, denseFrom);
      // (opt-in: measured on BAL-871 the update takes 2.4x as long beside the dense kernels and
      //  slows the chain by as much as it saves, 7.85 against 7.63 ms -- DESIGN.md "tried")
      const bool enabled = opts.elimOverlap;
      if (enabled && re == denseFrom && denseBegin0 == denseFrom && upToLump == denseFrom + 1 &&
          upToLump <= nLumps && elimBigBuckets.back().empty() &&
          sk.lumpStart[denseFrom + 1] - sk.lumpStart[denseFrom] >= 6 * kOuterWidth) {
        overlapLump = denseFrom;
      }
    }
    buildElimGather(sk, plan, er, overlapLump);
    plan.elimRanges.push_back(std::move(er));
  }
  // overlapped elimination: which column blocks of the dense lump may receive OPTIONAL lookahead
  // units forked at block b -- only those whose gather group is expected to be complete by then
  // (plan-time estimate; events enforce the order, this only keeps the side stream from stalling)
  ov = (!plan.elimRanges.empty() && plan.elimRanges.back().overlapLump >= 0)
           ? &plan.elimRanges.back() : nullptr;
  if (ov) {
    constexpr double kGatherPairsPerUs = 5.5e3;  // beside the dense kernels (8.9e3 alone)
    const LumpCols g = lumpCols(sk, ov->overlapLump);
    const int64_t numBlocks = (g.width + kOuterWidth - 1) / kOuterWidth;
    vector<double> groupDone(ov->groupPairs.size());
    double t = 0;
    for (size_t q = 0; q < groupDone.size(); q++) groupDone[q] = (t += ov->groupPairs[q] / kGatherPairsPerUs);
    double tBlock = groupDone[ov->groupOfColBlock(std::min<int64_t>(1, numBlocks - 1))];
    optionalLimit.assign(numBlocks, numBlocks);
    for (int64_t b = 0; b < numBlocks; b++) {
      const double rows = double(g.width - (b + 1) * kOuterWidth + g.rowsBelow);
      tBlock += 118.0 + 0.012 * std::max(rows, 0.0);  // units forked at block b start about here
      int64_t lim = b + 2;
      while (lim + 1 < numBlocks && groupDone[ov->groupOfColBlock(lim + 1)] <= tBlock) lim++;
      optionalLimit[b] = lim;
    }
  }

  // ---- dense lumps
  const int64_t denseBegin = std::max(startLump, denseFrom);
  vector<int32_t> lastLevelOfLump(nLumps, -1);
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
    const int32_t n = addPanels(l, g, lumpRowBase, /*withBoards=*/true, boardSegs);
    for (int32_t j = 0; j < n; j++) bucketAt(levelBuckets, level + j).push_back({first + j, level + j});
    lastLevelOfLump[l] = level + n - 1;
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
      vector<UpdTask> deferred, deferred1, deferredLate;  // due (first column tile) / due (rest) / optional
      // (opt-in: measured 7.32 against 7.23 ms on BAL-871 -- the extra small launch per block
      //  costs more than the shorter wait returns)
      const bool dueSplit = opts.dueSplit;
      int32_t maxCbMid = -1, maxCbLate = -1;  // furthest target column block of the deferred units
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
          if (sd.outer == 1 || (sd.pad & 8)) {
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
            const bool late = sd.outer == 3;  // (2: due at the block boundary, 4: early due)
            const bool split = dueSplit && sd.outer == 2;
            auto dstOf = [&](int32_t cT) -> vector<UpdTask>& {
              return late ? deferredLate : (split && cT != sd.q0 ? deferred1 : deferred);
            };
            const bool rowMajor = opts.bulkRowMajor;
            if (rowMajor) {
              for (int32_t rT = sd.q0; rT < sr.rowsBelow; rT += step) {
                for (int32_t cT = sd.q0; cT < sd.q0 + sd.m && cT <= rT; cT += step) {
                  // (the top-left tile of a column block is its tile (0,0): the chain may be
                  //  applying early rank-64 updates to it at the same time, LevelRange::extraDiag)
                  const int32_t a = atomic | ((rT == sd.q0 && cT == sd.q0) ? 1 : 0);
                  dstOf(cT).push_back(UpdTask{(int32_t)s, rT, cT, a});
                }
              }
            } else {
              for (int32_t cT = sd.q0; cT < sd.q0 + sd.m; cT += step) {
                for (int32_t rT = cT; rT < sr.rowsBelow; rT += step) {
                  const int32_t a = atomic | ((rT == sd.q0 && cT == sd.q0) ? 1 : 0);
                  dstOf(cT).push_back(UpdTask{(int32_t)s, rT, cT, a});
                }
              }
            }
            anyDeferred = true;
            const int32_t cb = s < (int64_t)plan.segColBlock.size() ? plan.segColBlock[s] : -1;
            (late ? maxCbLate : maxCbMid) = std::max(late ? maxCbLate : maxCbMid, cb);
          } else {
            // (CHAIN WINDOW segment: the tiles in the next block's columns may meet lookahead units;
            //  a block-last panel's level has waited for them.  The task-list kernels go by the flag,
            //  the direct chain kernels by SegDesc::firstChainOrd)
            const bool win = (sd.pad & 4) != 0 && !(sd.pad & 8);
            for (int32_t cT = sd.q0; cT < sd.q0 + sd.m; cT += step) {
              const int32_t a = atomic | ((win && cT >= sd.q0 + sd.firstChainOrd) ? 1 : 0);
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
            if (sd.outer == 4) {  // early due (one step before the boundary): not a fork of its own
              if (forks.size() >= 2) lr.optWaitLevel = std::max(lr.optWaitLevel, forks[forks.size() - 2]);
            } else {
              if (forks.empty() || forks.back() != levelIdx) {
                if (forks.size() >= 2) lr.optWaitLevel = std::max(lr.optWaitLevel, forks[forks.size() - 2]);
                forks.push_back(levelIdx);
              }
              lastDeferredLevel[sd.lump] = levelIdx;
            }
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
      lr.defMid0 = (int64_t)plan.updTasks.size();
      plan.updTasks.insert(plan.updTasks.end(), deferred1.begin(), deferred1.end());
      lr.defMid = (int64_t)plan.updTasks.size();
      plan.updTasks.insert(plan.updTasks.end(), deferredLate.begin(), deferredLate.end());
      lr.defEnd = (int64_t)plan.updTasks.size();
      lr.soonBegin = lr.soonMid = lr.soonEnd = lr.defEnd;
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
        // first panel of an outer block: the column tiles 1..3 of this block received their due
        // units from the fork two block boundaries back (DUE SPLIT)
        const PanelDesc& pd = plan.panels[bucket[0].panel];
        auto it = blockForks.find(pd.lump);
        if ((pd.lda - pd.nRest - pd.nb) % kOuterWidth == 0 && it != blockForks.end() && it->second.size() >= 2) {
          const int64_t f = it->second[it->second.size() - 2];
          if (out[f].defMid > out[f].defMid0) lr.waitDue1Level = f;
        }
      }
      if (ov && bucket.size() == 1 && plan.panels[bucket[0].panel].lump == ov->overlapLump) {
        // overlapped elimination: the kernels of outer block b touch column blocks b (panel steps)
        // and b + 1 (the block-wide "now" update); a deferred launch touches up to its furthest unit
        const PanelDesc& pd = plan.panels[bucket[0].panel];
        const int64_t c0 = pd.lda - pd.nRest - pd.nb;
        const int64_t numBlocks = (pd.lda + kOuterWidth - 1) / kOuterWidth;
        if (c0 % kOuterWidth == 0) {
          lr.waitGather = ov->groupOfColBlock(std::min<int64_t>(c0 / kOuterWidth + 1, numBlocks - 1));
        }
        if (maxCbMid >= 0) lr.defWaitGatherMid = ov->groupOfColBlock(maxCbMid);
        if (std::max(maxCbMid, maxCbLate) >= 0) {
          lr.defWaitGatherEnd = ov->groupOfColBlock(std::max(maxCbMid, maxCbLate));
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
            // (not with a CHAIN WINDOW segment: the next block's tile (0,0) is inside the window)
            if (!sd.outer && sd.kind == kSegIntra && sd.q0 == 0 && sr0.K == pd0.nb &&
                pd0.nb == kPanelWidth && pd0.nRest - sd.m >= kTile && !(sd.pad & 4)) {
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
      // NOW SPLIT (LevelRange::nowHeadTiles): block-last step of a chain with a full next block
      // (four column tiles) and at least that block's first two steps in this chain
      if (opts.nowSplit && chain && lr.directSeg >= 0 && lr.fuseNext &&
          bi + 2 < buckets.size() && buckets[bi + 2].size() == 1 &&
          plan.panels[buckets[bi + 2][0].panel].lump == plan.panels[bucket[0].panel].lump) {
        const SegDesc& sd = plan.segs[lr.directSeg];
        const SrcDesc& sr = plan.srcs[sd.src];
        const PanelDesc& pd = plan.panels[bucket[0].panel];
        if (sd.outer == 1 && sd.kind == kSegIntra && sd.m == kOuterWidth && sr.K == kOuterWidth &&
            sd.rowMin == 0 && pd.nb == kPanelWidth && sr.rowsBelow - sd.q0 >= 2 * kOuterWidth) {
          int32_t head = 0;
          for (int32_t cT = sd.q0; cT < sd.q0 + 2 * kTile; cT += kTile) {
            head += (sr.rowsBelow - cT + kTile - 1) / kTile;
          }
          lr.nowHeadTiles = head;
          for (int32_t cT = sd.q0 + 2 * kTile; cT < sd.q0 + sd.m; cT += kTile) {
            if (cT == sd.q0 + 3 * kTile) lr.soonMid = (int64_t)plan.updTasks.size();
            for (int32_t rT = cT; rT < sr.rowsBelow; rT += kTile) {
              plan.updTasks.push_back(UpdTask{lr.directSeg, rT, cT, 1});
            }
          }
          lr.soonEnd = (int64_t)plan.updTasks.size();
        }
      }
      xcdOrder(lr.updBegin, lr.updEnd);
      xcdOrder(lr.defBegin, lr.defMid0);
      xcdOrder(lr.defMid0, lr.defMid);
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
      lr.defBegin = lr.defMid0 = lr.defMid = lr.defEnd = lr.updEnd;
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
