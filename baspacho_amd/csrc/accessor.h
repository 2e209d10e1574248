// Block accessors: how callers locate a block of the factor inside the flat numeric buffer.
// Plain structs of raw pointers (POD, no constructor) so they can be passed by value to a HIP
// kernel.  Same names/semantics as baspacho/baspacho/Accessor.h:18-200; the Eigen `block()`
// helpers are replaced by raw (pointer, stride) views since Eigen is not a dependency here.
#pragma once

#include <cstdint>
#include <tuple>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BASPACHO_HOST_DEVICE __host__ __device__
#else
#define BASPACHO_HOST_DEVICE
#endif

namespace BaSpaCho {

struct CoalescedAccessor {
  void init(const int64_t* spanStart_, const int64_t* spanToLump_, const int64_t* lumpStart_,
            const int64_t* spanOffsetInLump_, const int64_t* chainColPtr_,
            const int64_t* chainRowSpan_, const int64_t* chainData_) {
    spanStart = spanStart_;
    spanToLump = spanToLump_;
    lumpStart = lumpStart_;
    spanOffsetInLump = spanOffsetInLump_;
    chainColPtr = chainColPtr_;
    chainRowSpan = chainRowSpan_;
    chainData = chainData_;
  }

  BASPACHO_HOST_DEVICE int64_t paramSize(int64_t blockIndex) const {
    return spanStart[blockIndex + 1] - spanStart[blockIndex];
  }

  BASPACHO_HOST_DEVICE int64_t paramStart(int64_t blockIndex) const {
    return spanStart[blockIndex];
  }

  // (offset, row stride) of block (row,col), row >= col, inside the numeric data
  BASPACHO_HOST_DEVICE std::pair<int64_t, int64_t> blockOffset(int64_t rowBlockIndex,
                                                               int64_t colBlockIndex) const {
    int64_t lump = spanToLump[colBlockIndex];
    int64_t lumpSize = lumpStart[lump + 1] - lumpStart[lump];
    int64_t first = chainColPtr[lump], count = chainColPtr[lump + 1] - first;
    // chains of a column are sorted by row span: binary search
    int64_t lo = 0, hi = count;
    while (hi - lo > 1) {
      int64_t mid = lo + (hi - lo) / 2;
      if (chainRowSpan[first + mid] <= rowBlockIndex) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    return {chainData[first + lo] + spanOffsetInLump[colBlockIndex], lumpSize};
  }

  // true when block (row,col) exists in the structure
  BASPACHO_HOST_DEVICE bool hasBlock(int64_t rowBlockIndex, int64_t colBlockIndex) const {
    int64_t lump = spanToLump[colBlockIndex];
    int64_t first = chainColPtr[lump], count = chainColPtr[lump + 1] - first;
    int64_t lo = 0, hi = count;
    while (hi - lo > 1) {
      int64_t mid = lo + (hi - lo) / 2;
      if (chainRowSpan[first + mid] <= rowBlockIndex) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    return chainRowSpan[first + lo] == rowBlockIndex;
  }

  BASPACHO_HOST_DEVICE std::pair<int64_t, int64_t> diagBlockOffset(int64_t blockIndex) const {
    int64_t lump = spanToLump[blockIndex];
    int64_t lumpSize = lumpStart[lump + 1] - lumpStart[lump];
    return {chainData[chainColPtr[lump]] + spanOffsetInLump[blockIndex] * (lumpSize + 1),
            lumpSize};
  }

  const int64_t* spanStart;
  const int64_t* spanToLump;
  const int64_t* lumpStart;
  const int64_t* spanOffsetInLump;
  const int64_t* chainColPtr;
  const int64_t* chainRowSpan;
  const int64_t* chainData;
};

struct PermutedCoalescedAccessor {
  void init(const CoalescedAccessor& plainAcc_, const int64_t* permutation_) {
    plainAcc = plainAcc_;
    permutation = permutation_;
  }

  void init(const int64_t* spanStart_, const int64_t* spanToLump_, const int64_t* lumpStart_,
            const int64_t* spanOffsetInLump_, const int64_t* chainColPtr_,
            const int64_t* chainRowSpan_, const int64_t* chainData_,
            const int64_t* permutation_) {
    plainAcc.init(spanStart_, spanToLump_, lumpStart_, spanOffsetInLump_, chainColPtr_,
                  chainRowSpan_, chainData_);
    permutation = permutation_;
  }

  BASPACHO_HOST_DEVICE int64_t paramSize(int64_t blockIndex) const {
    return plainAcc.paramSize(permutation[blockIndex]);
  }

  BASPACHO_HOST_DEVICE int64_t paramStart(int64_t blockIndex) const {
    return plainAcc.paramStart(permutation[blockIndex]);
  }

  // (offset, stride, flipped): flipped means the stored block is the transpose of (row,col)
  BASPACHO_HOST_DEVICE std::tuple<int64_t, int64_t, bool> blockOffset(
      int64_t rowBlockIndex, int64_t colBlockIndex) const {
    int64_t pr = permutation[rowBlockIndex], pc = permutation[colBlockIndex];
    auto offStride = plainAcc.blockOffset(pr > pc ? pr : pc, pr > pc ? pc : pr);
    return std::make_tuple(offStride.first, offStride.second, pr < pc);
  }

  BASPACHO_HOST_DEVICE std::pair<int64_t, int64_t> diagBlockOffset(int64_t blockIndex) const {
    return plainAcc.diagBlockOffset(permutation[blockIndex]);
  }

  CoalescedAccessor plainAcc;
  const int64_t* permutation;
};

}  // namespace BaSpaCho
