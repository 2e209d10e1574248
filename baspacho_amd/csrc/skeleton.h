// The factor skeleton ("plan"): index arrays describing a symmetric block matrix whose columns
// have been coalesced into lumps (supernodes).
//   span  = one user parameter block          lump  = consecutive spans factored together
//   chain = (row span) x (lump columns)       board = (all row spans of one row lump) x (lump)
// The numeric data of one lump column is ONE dense row-major matrix (all chain rows x lumpSize);
// its first board is the full square diagonal block.
// Names and array semantics are those of baspacho/baspacho/CoalescedBlockMatrix.h:38-111
// (constructor CoalescedBlockMatrix.cpp:17-122), which the reference's unit test pins with
// literal arrays (tests/CoalescedBlockMatrixTest.cpp:48-112).
#pragma once

#include <cstdint>
#include <vector>

#include "accessor.h"
#include "bsp_utils.h"

namespace BaSpaCho {

constexpr int64_t kInvalid = -1;

struct CoalescedBlockMatrixSkel {
  CoalescedBlockMatrixSkel(const std::vector<int64_t>& spanStart,
                           const std::vector<int64_t>& lumpToSpan,
                           const std::vector<int64_t>& colPtr, const std::vector<int64_t>& rowInd);

  // dense (order x order, COLUMN-major like the reference's Eigen default) copy of the data;
  // lower half only unless fillUpperHalf; startSpanIndex (lump boundary) selects the
  // bottom-right corner
  template <typename T>
  void densify(std::vector<T>& dense, int64_t& denseOrder, const T* data,
               bool fillUpperHalf = false, int64_t startSpanIndex = 0) const;

  // data.diagonal = data.diagonal * (1 + alpha) + beta
  template <typename T>
  void damp(T* data, int64_t dataLen, T alpha, T beta) const;
  template <typename T>
  void damp(std::vector<T>& data, T alpha, T beta) const {
    damp(data.data(), (int64_t)data.size(), alpha, beta);
  }

  int64_t numSpans() const { return (int64_t)spanStart.size() - 1; }
  int64_t numLumps() const { return (int64_t)lumpStart.size() - 1; }
  int64_t order() const { return spanStart.back(); }
  int64_t dataSize() const { return chainData.back(); }
  int64_t spanVectorOffset(int64_t span) const { return spanStart[span]; }
  int64_t spanMatrixOffset(int64_t span) const {
    BASPACHO_CHECK_EQ(spanOffsetInLump[span], 0);
    return chainData[chainColPtr[spanToLump[span]]];
  }

  CoalescedAccessor accessor() const {
    CoalescedAccessor acc;
    acc.init(spanStart.data(), spanToLump.data(), lumpStart.data(), spanOffsetInLump.data(),
             chainColPtr.data(), chainRowSpan.data(), chainData.data());
    return acc;
  }

  std::vector<int64_t> spanStart;         // (with final el)
  std::vector<int64_t> spanToLump;        // (with final el)
  std::vector<int64_t> lumpStart;         // (with final el)
  std::vector<int64_t> lumpToSpan;        // (with final el)
  std::vector<int64_t> spanOffsetInLump;  // (with final el)

  // per-chain data, column-ordered
  std::vector<int64_t> chainColPtr;       // first chain of each lump column (with end)
  std::vector<int64_t> chainRowSpan;      // row span of the chain
  std::vector<int64_t> chainData;         // numeric data offset (with end = dataSize)
  std::vector<int64_t> chainRowsTillEnd;  // rows of the column up to and including this chain

  // per-board data, column-ordered; every column ends with a sentinel board
  std::vector<int64_t> boardColPtr;       // first board of each column (with end)
  std::vector<int64_t> boardRowLump;      // row lump (sentinel = kInvalid)
  std::vector<int64_t> boardChainColOrd;  // ordinal of the board's first chain (sentinel = #chains)

  // per-board data, row-ordered (no sentinels)
  std::vector<int64_t> boardRowPtr;   // first board of each row lump (with end)
  std::vector<int64_t> boardColLump;  // column lump of the board
  std::vector<int64_t> boardColOrd;   // ordinal of the board inside its column
};

}  // namespace BaSpaCho
