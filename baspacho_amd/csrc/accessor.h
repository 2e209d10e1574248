// Block accessors: how callers locate a block of the factor inside the flat numeric buffer.
// Plain structs of raw pointers (POD, no constructor) so they can be passed by value to a HIP
// kernel.  Same names/semantics as baspacho/baspacho/Accessor.h:18-200.  Eigen is not a dependency
// here: `block()` / `diagBlock()` (Accessor.h:69-107,165-200) return a BlockView -- a strided
// (rows, cols, rowStride, colStride) view with operator()(i, j), setZero(), +=, -= and transpose(),
// the subset of Eigen::Map the reference's callers use (BaAtLargeOptimizer.cpp:119-129) -- with the
// same template arguments (compile-time sizes are checked against the parameter sizes) and the
// same flip handling for permuted accessors (a flipped block is a view with swapped strides).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BASPACHO_HOST_DEVICE __host__ __device__
#else
#define BASPACHO_HOST_DEVICE
#endif

namespace BaSpaCho {

constexpr int Dynamic = -1;  // Eigen::Dynamic's role in block<rowSize, colSize>()

// compile-time block sizes must match the parameter sizes (BASPACHO_CHECK_EQ in Accessor.h:76-81):
// an exception on the host, a trap inside a kernel
BASPACHO_HOST_DEVICE inline void checkBlockSize(int64_t wanted, int64_t actual) {
  if (wanted != Dynamic && wanted != actual) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_trap();
#else
    throw std::runtime_error("accessor: compile-time block size " + std::to_string(wanted) +
                             " != parameter size " + std::to_string(actual));
#endif
  }
}

// Strided view of a block of the numeric data (what the reference returns as
// Eigen::Map<Matrix<T, r, c, RowMajor>, 0, Stride<Dynamic, Dynamic>>, Accessor.h:69-107,165-200)
template <typename T>
struct BlockView {
  T* ptr;
  int64_t nRows, nCols, rowStride, colStride;
  BASPACHO_HOST_DEVICE int64_t rows() const { return nRows; }
  BASPACHO_HOST_DEVICE int64_t cols() const { return nCols; }
  BASPACHO_HOST_DEVICE T& operator()(int64_t i, int64_t j) const { return ptr[i * rowStride + j * colStride]; }
  BASPACHO_HOST_DEVICE BlockView transpose() const { return {ptr, nCols, nRows, colStride, rowStride}; }
  BASPACHO_HOST_DEVICE void setZero() const {
    for (int64_t i = 0; i < nRows; i++) {
      for (int64_t j = 0; j < nCols; j++) (*this)(i, j) = T(0);
    }
  }
  // element-wise accumulate from anything indexable as src(i, j) (another view, a lambda, ...)
  template <typename Src>
  BASPACHO_HOST_DEVICE const BlockView& operator+=(const Src& src) const {
    for (int64_t i = 0; i < nRows; i++) {
      for (int64_t j = 0; j < nCols; j++) (*this)(i, j) += src(i, j);
    }
    return *this;
  }
  template <typename Src>
  BASPACHO_HOST_DEVICE const BlockView& operator-=(const Src& src) const {
    for (int64_t i = 0; i < nRows; i++) {
      for (int64_t j = 0; j < nCols; j++) (*this)(i, j) -= src(i, j);
    }
    return *this;
  }
};

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

  // block reference, from the numeric data pointer (Accessor.h:69-87); row >= col
  template <int rowSize = Dynamic, int colSize = Dynamic, typename T>
  BASPACHO_HOST_DEVICE BlockView<T> block(T* data, int64_t rowBlockIndex, int64_t colBlockIndex) const {
    checkBlockSize(rowSize, paramSize(rowBlockIndex));
    checkBlockSize(colSize, paramSize(colBlockIndex));
    auto os = blockOffset(rowBlockIndex, colBlockIndex);
    return {data + os.first, rowSize != Dynamic ? rowSize : paramSize(rowBlockIndex),
            colSize != Dynamic ? colSize : paramSize(colBlockIndex), os.second, 1};
  }

  // diagonal block reference (Accessor.h:89-101)
  template <int size = Dynamic, typename T>
  BASPACHO_HOST_DEVICE BlockView<T> diagBlock(T* data, int64_t blockIndex) const {
    checkBlockSize(size, paramSize(blockIndex));
    auto os = diagBlockOffset(blockIndex);
    const int64_t n = size != Dynamic ? size : paramSize(blockIndex);
    return {data + os.first, n, n, os.second, 1};
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

  // block reference through the permutation (Accessor.h:165-185): a flipped block (stored as the
  // transpose) comes back as a view with swapped strides, so view(i, j) is always entry (i, j) of
  // block (rowBlockIndex, colBlockIndex)
  template <int rowSize = Dynamic, int colSize = Dynamic, typename T>
  BASPACHO_HOST_DEVICE BlockView<T> block(T* data, int64_t rowBlockIndex, int64_t colBlockIndex) const {
    checkBlockSize(rowSize, paramSize(rowBlockIndex));
    checkBlockSize(colSize, paramSize(colBlockIndex));
    auto osf = blockOffset(rowBlockIndex, colBlockIndex);
    const bool flip = std::get<2>(osf);
    return {data + std::get<0>(osf), rowSize != Dynamic ? rowSize : paramSize(rowBlockIndex),
            colSize != Dynamic ? colSize : paramSize(colBlockIndex), flip ? 1 : std::get<1>(osf),
            flip ? std::get<1>(osf) : 1};
  }

  template <int size = Dynamic, typename T>
  BASPACHO_HOST_DEVICE BlockView<T> diagBlock(T* data, int64_t blockIndex) const {
    checkBlockSize(size, paramSize(blockIndex));
    auto os = diagBlockOffset(blockIndex);
    const int64_t n = size != Dynamic ? size : paramSize(blockIndex);
    return {data + os.first, n, n, os.second, 1};
  }

  CoalescedAccessor plainAcc;
  const int64_t* permutation;
};

}  // namespace BaSpaCho
