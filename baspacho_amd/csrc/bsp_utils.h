// Small host utilities shared by the symbolic phase and the solver driver.
// Behavioural reference: baspacho/baspacho/Utils.h:123-196, Utils.cpp:70-105,
// DebugMacros.h:17-50 (precondition failures throw std::runtime_error).
#pragma once

#include <array>
#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace BaSpaCho {

[[noreturn]] void throwError(const char* file, int line, const std::string& msg);

#define BASPACHO_CHECK(cond)                                               \
  do {                                                                     \
    if (!(cond)) ::BaSpaCho::throwError(__FILE__, __LINE__, #cond);        \
  } while (0)

#define BASPACHO_CHECK_OP(a, op, b)                                        \
  do {                                                                     \
    auto&& bsp_va_ = (a);                                                  \
    auto&& bsp_vb_ = (b);                                                  \
    if (!(bsp_va_ op bsp_vb_)) {                                           \
      std::ostringstream bsp_os_;                                          \
      bsp_os_ << #a " " #op " " #b " (" << bsp_va_ << " vs " << bsp_vb_    \
              << ")";                                                      \
      ::BaSpaCho::throwError(__FILE__, __LINE__, bsp_os_.str());           \
    }                                                                      \
  } while (0)

#define BASPACHO_CHECK_EQ(a, b) BASPACHO_CHECK_OP(a, ==, b)
#define BASPACHO_CHECK_NE(a, b) BASPACHO_CHECK_OP(a, !=, b)
#define BASPACHO_CHECK_LE(a, b) BASPACHO_CHECK_OP(a, <=, b)
#define BASPACHO_CHECK_LT(a, b) BASPACHO_CHECK_OP(a, <, b)
#define BASPACHO_CHECK_GE(a, b) BASPACHO_CHECK_OP(a, >=, b)
#define BASPACHO_CHECK_GT(a, b) BASPACHO_CHECK_OP(a, >, b)
#define BASPACHO_CHECK_NOTNULL(p) BASPACHO_CHECK((p) != nullptr)

// largest index a in [0,size) with array[a] <= needle (array sorted, array[0] <= needle)
inline int64_t bisect(const int64_t* array, int64_t size, int64_t needle) {
  int64_t lo = 0, hi = size;
  while (hi - lo > 1) {
    int64_t mid = lo + (hi - lo) / 2;
    if (array[mid] <= needle) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  return lo;
}

template <typename T>
bool isStrictlyIncreasing(const std::vector<T>& v, size_t begin, size_t end) {
  for (size_t i = begin + 1; i < end; i++) {
    if (!(v[i - 1] < v[i])) return false;
  }
  return true;
}

template <typename T>
bool isWeaklyIncreasing(const std::vector<T>& v, size_t begin, size_t end) {
  for (size_t i = begin + 1; i < end; i++) {
    if (v[i] < v[i - 1]) return false;
  }
  return true;
}

// exclusive prefix sum in place; v has one trailing slot that receives the total
int64_t cumSumVec(std::vector<int64_t>& v);

// shift entries one slot to the right (undoing pointer advancing), v[downTo] = value
void rewindVec(std::vector<int64_t>& v, int64_t downTo = 0, int64_t value = 0);

// retv[p[i]] = i
std::vector<int64_t> inversePermutation(const std::vector<int64_t>& p);

// retv[i] = v[w[i]]
std::vector<int64_t> composePermutations(const std::vector<int64_t>& v,
                                         const std::vector<int64_t>& w);

// it[perm[i]] = w[i]
template <typename T, typename It>
void leftPermute(It it, const std::vector<int64_t>& perm, const std::vector<T>& w) {
  for (size_t i = 0; i < perm.size(); i++) it[perm[i]] = w[i];
}

template <typename It>
void shiftConcat(std::vector<int64_t>& target, int64_t shift, It first, It last) {
  for (; first != last; ++first) target.push_back(*first + shift);
}

std::string secondsToString(double secs, int precision = 3);

// Lightweight per-op statistics (reference: OpStat in Utils.h:48-121).  No sync hook
// here: the HIP backend never synchronises per op, timings come from HIP events.
struct OpStat {
  bool enabled = false;
  int64_t numRuns = 0;
  double totTime = 0, lastTime = 0, maxTime = 0;
  // per-call samples {size0, size1, size2, seconds} (the role of the reference's callBack hook,
  // Utils.h:100-119, which bench -Z uses to dump one CSV row per op, Bench.cpp:72-124)
  bool keepSamples = false;
  std::vector<std::array<double, 4>> samples;
  void reset() {
    numRuns = 0;
    totTime = lastTime = maxTime = 0;
    samples.clear();
  }
  void add(double t, double s0 = 0, double s1 = 0, double s2 = 0) {
    numRuns++;
    totTime += t;
    lastTime = t;
    if (t > maxTime) maxTime = t;
    if (keepSamples) samples.push_back({s0, s1, s2, t});
  }
  std::string toString() const;
};

}  // namespace BaSpaCho
