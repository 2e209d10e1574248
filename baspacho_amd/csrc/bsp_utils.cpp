#include "bsp_utils.h"
#include "backend_options.h"
#include <cstdlib>
#include <algorithm>

#include <chrono>
#include <cmath>
#include <ctime>
#include <iomanip>

namespace BaSpaCho {

void throwError(const char* file, int line, const std::string& msg) {
  using namespace std::chrono;
  auto now = system_clock::now();
  std::time_t tt = system_clock::to_time_t(now);
  struct tm lt;
  localtime_r(&tt, &lt);
  auto ms = duration_cast<milliseconds>(now.time_since_epoch()).count() % 1000;
  std::ostringstream os;
  os << "[" << std::put_time(&lt, "%T") << "." << std::setfill('0') << std::setw(3) << ms << " "
     << file << ":" << line << "] Check failed: " << msg;
  throw std::runtime_error(os.str());
}

int64_t cumSumVec(std::vector<int64_t>& v) {
  int64_t running = 0;
  for (size_t i = 0; i + 1 < v.size(); i++) {
    int64_t cur = v[i];
    v[i] = running;
    running += cur;
  }
  v.back() = running;
  return running;
}

void rewindVec(std::vector<int64_t>& v, int64_t downTo, int64_t value) {
  for (int64_t i = (int64_t)v.size() - 1; i > downTo; i--) v[i] = v[i - 1];
  v[downTo] = value;
}

std::vector<int64_t> inversePermutation(const std::vector<int64_t>& p) {
  std::vector<int64_t> inv(p.size());
  for (size_t i = 0; i < p.size(); i++) inv[p[i]] = (int64_t)i;
  return inv;
}

std::vector<int64_t> composePermutations(const std::vector<int64_t>& v,
                                         const std::vector<int64_t>& w) {
  BASPACHO_CHECK_EQ(v.size(), w.size());
  std::vector<int64_t> out(v.size());
  for (size_t i = 0; i < v.size(); i++) out[i] = v[w[i]];
  return out;
}

std::string secondsToString(double secs, int precision) {
  std::ostringstream os;
  os << std::fixed << std::setprecision(precision);
  if (secs < 1e-3) {
    os << secs * 1e6 << "us";
  } else if (secs < 1.0) {
    os << secs * 1e3 << "ms";
  } else {
    os << secs << "s";
  }
  return os.str();
}

std::string OpStat::toString() const {
  std::ostringstream os;
  os << "#=" << numRuns << ", time=" << secondsToString(totTime)
     << ", last=" << secondsToString(lastTime) << ", max=" << secondsToString(maxTime);
  return os.str();
}

// the ONLY place the schedule switches are read from the environment (backend_options.h)
void HipBackendOptions::applyEnv() {
  auto flag = [](const char* name, int32_t& dst, bool invert = false) {
    if (const char* e = std::getenv(name)) dst = ((e[0] != '0') != invert) ? 1 : 0;
  };
  auto num = [](const char* name, int32_t& dst, long lo) {
    if (const char* e = std::getenv(name)) dst = (int32_t)std::max<long>(lo, std::strtol(e, nullptr, 0));
  };
  auto real = [](const char* name, double& dst) {
    if (const char* e = std::getenv(name)) dst = std::atof(e);
  };
  flag("BSP_NO_LOOKAHEAD", lookahead, /*invert=*/true);
  flag("BSP_DUE_STREAM", dueStream);
  flag("BSP_SPLIT_K", splitK);
  num("BSP_GATHER_MAX_PAIRS", gatherMaxPairs, 8);
  flag("BSP_GATHER_OVERLAP", gatherOverlap);
  num("BSP_SUB_BATCH_MIN", subBatchMin, 0);
  num("BSP_SUB_BATCHES", subBatches, 2);
  num("BSP_TAIL_BLOCKS", tailBlocks, 0);
  if (std::getenv("BSP_LAZY_PLAN")) lazyPlan = 1;
  flag("BSP_BLOCK_SOLVE", blockSolve);
  flag("BSP_SOLVE_INV", solveInv);
  flag("BSP_SOLVE_SWEEP", solveSweep);
  num("BSP_SWEEP_MIN_WIDTH", sweepMinWidth, 1);
  flag("BSP_SOLVE_WIDE", solveWide);
  flag("BSP_CHAIN_CONTRACTION", chainContraction);
  if (std::getenv("BSP_DENSE_MERGE_OFF")) denseMerge = 0;
  num("BSP_EXPECTED_BATCH", expectedBatch, 1);
  real("BSP_LOOKAHEAD_MIN_GF", lookaheadMinGF);
  real("BSP_BULK_AHEAD", bulkAhead);
  real("BSP_LEVEL_COST_US", levelCostUs);
}

}  // namespace BaSpaCho
