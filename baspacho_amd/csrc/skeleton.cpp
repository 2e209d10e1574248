#include "skeleton.h"

namespace BaSpaCho {

using std::vector;

CoalescedBlockMatrixSkel::CoalescedBlockMatrixSkel(const vector<int64_t>& spanStart_,
                                                   const vector<int64_t>& lumpToSpan_,
                                                   const vector<int64_t>& colPtr,
                                                   const vector<int64_t>& rowInd)
    : spanStart(spanStart_), lumpToSpan(lumpToSpan_) {
  BASPACHO_CHECK_GE((int64_t)lumpToSpan.size(), 1);
  BASPACHO_CHECK_GE(spanStart.size(), lumpToSpan.size());
  BASPACHO_CHECK_EQ((int64_t)spanStart.size() - 1, lumpToSpan.back());
  BASPACHO_CHECK_EQ(colPtr.size(), lumpToSpan.size());
  BASPACHO_CHECK(isStrictlyIncreasing(spanStart, 0, spanStart.size()));
  BASPACHO_CHECK(isStrictlyIncreasing(lumpToSpan, 0, lumpToSpan.size()));

  const int64_t nSpans = (int64_t)spanStart.size() - 1;
  const int64_t nLumps = (int64_t)lumpToSpan.size() - 1;

  // span <-> lump maps
  spanToLump.assign(nSpans + 1, nLumps);
  lumpStart.assign(nLumps + 1, spanStart[nSpans]);
  spanOffsetInLump.assign(nSpans + 1, 0);
  for (int64_t l = 0; l < nLumps; l++) {
    lumpStart[l] = spanStart[lumpToSpan[l]];
    for (int64_t s = lumpToSpan[l]; s < lumpToSpan[l + 1]; s++) {
      spanToLump[s] = l;
      spanOffsetInLump[s] = spanStart[s] - lumpStart[l];
    }
  }

  // column-ordered chains and boards
  chainColPtr.assign(nLumps + 1, 0);
  boardColPtr.assign(nLumps + 1, 0);
  int64_t dataCursor = 0;
  for (int64_t l = 0; l < nLumps; l++) {
    const int64_t first = colPtr[l], last = colPtr[l + 1];
    const int64_t nDiagSpans = lumpToSpan[l + 1] - lumpToSpan[l];
    const int64_t width = lumpStart[l + 1] - lumpStart[l];
    BASPACHO_CHECK(isStrictlyIncreasing(rowInd, first, last));
    // the column must open with the complete diagonal block
    BASPACHO_CHECK_GE(last - first, nDiagSpans);
    BASPACHO_CHECK_EQ(rowInd[first], lumpToSpan[l]);
    BASPACHO_CHECK_EQ(rowInd[first + nDiagSpans - 1], lumpToSpan[l + 1] - 1);

    chainColPtr[l] = (int64_t)chainRowSpan.size();
    boardColPtr[l] = (int64_t)boardRowLump.size();
    int64_t rowsSoFar = 0, openRowLump = kInvalid;
    for (int64_t q = first; q < last; q++) {
      const int64_t span = rowInd[q];
      const int64_t spanRows = spanStart[span + 1] - spanStart[span];
      if (spanToLump[span] != openRowLump) {  // a new board starts here
        openRowLump = spanToLump[span];
        boardRowLump.push_back(openRowLump);
        boardChainColOrd.push_back(q - first);
      }
      chainRowSpan.push_back(span);
      chainData.push_back(dataCursor);
      dataCursor += spanRows * width;
      rowsSoFar += spanRows;
      chainRowsTillEnd.push_back(rowsSoFar);
    }
    boardRowLump.push_back(kInvalid);  // sentinel
    boardChainColOrd.push_back(last - first);
  }
  chainColPtr[nLumps] = (int64_t)chainRowSpan.size();
  boardColPtr[nLumps] = (int64_t)boardRowLump.size();
  chainData.push_back(dataCursor);

  // row-ordered view of the (non-sentinel) boards
  boardRowPtr.assign(nLumps + 1, 0);
  for (int64_t l = 0; l < nLumps; l++) {
    for (int64_t b = boardColPtr[l]; b + 1 < boardColPtr[l + 1]; b++) boardRowPtr[boardRowLump[b]]++;
  }
  const int64_t nBoards = cumSumVec(boardRowPtr);
  boardColLump.resize(nBoards);
  boardColOrd.resize(nBoards);
  vector<int64_t> cursor(boardRowPtr.begin(), boardRowPtr.end() - 1);
  for (int64_t l = 0; l < nLumps; l++) {
    for (int64_t b = boardColPtr[l]; b + 1 < boardColPtr[l + 1]; b++) {
      int64_t slot = cursor[boardRowLump[b]]++;
      boardColLump[slot] = l;
      boardColOrd[slot] = b - boardColPtr[l];
    }
  }
}

template <typename T>
void CoalescedBlockMatrixSkel::densify(vector<T>& dense, int64_t& denseOrder, const T* data,
                                       bool fillUpperHalf, int64_t startSpanIndex) const {
  BASPACHO_CHECK_GE(startSpanIndex, 0);
  BASPACHO_CHECK_LT(startSpanIndex, (int64_t)spanOffsetInLump.size());
  BASPACHO_CHECK_EQ(spanOffsetInLump[startSpanIndex], 0);
  const int64_t base = spanStart[startSpanIndex];
  const int64_t n = order() - base;
  denseOrder = n;
  dense.assign((size_t)(n * n), T(0));
  for (int64_t l = spanToLump[startSpanIndex]; l < numLumps(); l++) {
    const int64_t width = lumpStart[l + 1] - lumpStart[l];
    const int64_t col0 = lumpStart[l] - base;
    for (int64_t c = chainColPtr[l]; c < chainColPtr[l + 1]; c++) {
      const int64_t span = chainRowSpan[c];
      const int64_t row0 = spanStart[span] - base;
      const int64_t rows = spanStart[span + 1] - spanStart[span];
      const T* src = data + chainData[c];
      for (int64_t r = 0; r < rows; r++) {
        for (int64_t q = 0; q < width; q++) {
          dense[(size_t)((col0 + q) * n + row0 + r)] = src[r * width + q];  // column-major
        }
      }
    }
  }
  if (fillUpperHalf) {
    for (int64_t c = 0; c < n; c++) {
      for (int64_t r = c + 1; r < n; r++) dense[(size_t)(r * n + c)] = dense[(size_t)(c * n + r)];
    }
  }
}

template <typename T>
void CoalescedBlockMatrixSkel::damp(T* data, int64_t dataLen, T alpha, T beta) const {
  BASPACHO_CHECK_EQ(dataLen, dataSize());
  for (int64_t l = 0; l < numLumps(); l++) {
    const int64_t width = lumpStart[l + 1] - lumpStart[l];
    T* diag = data + chainData[chainColPtr[l]];
    for (int64_t i = 0; i < width; i++) {
      T& d = diag[i * (width + 1)];
      d = d * (T(1) + alpha) + beta;
    }
  }
}

template void CoalescedBlockMatrixSkel::densify<double>(vector<double>&, int64_t&, const double*,
                                                        bool, int64_t) const;
template void CoalescedBlockMatrixSkel::densify<float>(vector<float>&, int64_t&, const float*, bool,
                                                       int64_t) const;
template void CoalescedBlockMatrixSkel::damp<double>(double*, int64_t, double, double) const;
template void CoalescedBlockMatrixSkel::damp<float>(float*, int64_t, float, float) const;

}  // namespace BaSpaCho
