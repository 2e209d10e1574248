#include "min_degree.h"

#include <algorithm>
#include <unordered_set>

#include "bsp_utils.h"

namespace BaSpaCho {

namespace {

enum NodeState : uint8_t { kVar, kElem, kMergedVar, kDeadElem };

struct QuotientGraph {
  int64_t n;
  std::vector<std::vector<int32_t>> adjVars;   // variable -> neighbouring variables
  std::vector<std::vector<int32_t>> adjElems;  // variable -> neighbouring elements
  std::vector<std::vector<int32_t>> elemVars;  // element  -> its variables (pattern L_e)
  std::vector<std::vector<int32_t>> members;   // principal variable -> variables merged into it
  std::vector<NodeState> state;
  std::vector<int64_t> nv;      // supervariable weight (0 when merged away)
  std::vector<int64_t> degree;  // approx external degree (vars) / weighted |L_e| (elements)
  std::vector<int64_t> w;       // scratch stamps for |L_e \ L_p|
  std::vector<int64_t> mark;    // scratch membership stamps
  int64_t wflg = 1, markStamp = 0;

  // degree buckets
  std::vector<int32_t> head, next, prev;
  int64_t minDeg = 0;

  explicit QuotientGraph(int64_t n_)
      : n(n_), adjVars(n_), adjElems(n_), elemVars(n_), members(n_), state(n_, kVar), nv(n_, 1),
        degree(n_, 0), w(n_, 0), mark(n_, 0), head(n_ + 1, -1), next(n_, -1), prev(n_, -1) {}

  void bucketInsert(int32_t i) {
    int64_t d = degree[i];
    next[i] = head[d];
    prev[i] = -1;
    if (head[d] >= 0) prev[head[d]] = i;
    head[d] = i;
    if (d < minDeg) minDeg = d;
  }
  void bucketRemove(int32_t i) {
    int64_t d = degree[i];
    if (prev[i] >= 0) {
      next[prev[i]] = next[i];
    } else {
      head[d] = next[i];
    }
    if (next[i] >= 0) prev[next[i]] = prev[i];
    next[i] = prev[i] = -1;
  }
};

}  // namespace

// Chain contraction.  A node with at most two neighbours is a safe pivot for minimum degree (its
// elimination adds at most one edge), but min-degree proper takes such nodes one END of a chain at
// a time and hands back an elimination tree that is a path: one dependent step per node, which on
// this device is one dependent launch (or in-kernel column step) per node.  Taking a maximal
// INDEPENDENT set of them per round instead (odd-even / cyclic reduction on a chain) gives rounds
// that halve the chain: every round is a set of independent small lumps, i.e. a sparse-elimination
// range (elimination_tree.cpp, computeSparseElimRanges), and the depth is logarithmic.  The price is
// the fill edge between the two neighbours of every pivot (2x the blocks of a block-tridiagonal
// matrix).  Rounds stop when the set is too small to become a range.  Returns the pivots in order
// and leaves the contracted graph in adj (dead nodes have alive[i] == 0).
static std::vector<int64_t> contractChains(std::vector<std::vector<int32_t>>& adj,
                                           std::vector<uint8_t>& alive, int64_t minRound) {
  // Adjacency lists keep their dead entries (skipped on the way, dropped by the caller) and live
  // degrees are counted separately, so that a hub with 10^5 leaves costs its leaves nothing; the
  // edge set for "is the fill edge new" is built on first use only (graphs without such pivots --
  // grids, dense camera blocks -- never pay for it).
  const int64_t n = (int64_t)adj.size();
  std::vector<int64_t> order;
  std::vector<int32_t> deg(n);
  for (int64_t v = 0; v < n; v++) deg[v] = (int32_t)adj[v].size();
  std::vector<int64_t> blockedAt(n, -1);
  struct Pick { int32_t v, a, b; };  // pivot and its (up to two) live neighbours, -1 = none
  std::vector<Pick> picked;
  std::unordered_set<uint64_t> edges;
  bool haveEdges = false;
  auto key = [](int32_t x, int32_t y) {
    return (uint64_t)(uint32_t)std::min(x, y) << 32 | (uint32_t)std::max(x, y);
  };
  // candidates = live nodes with at most two live neighbours, kept as a worklist: a round costs its
  // candidates, not a scan of all n nodes (50 parallel strips that release one pivot each per round
  // would otherwise take n / 50 rounds of n steps)
  std::vector<int32_t> cand, nextCand;
  std::vector<uint8_t> inCand(n, 0);
  for (int64_t v = 0; v < n; v++) {
    if (alive[v] && deg[v] <= 2) {
      cand.push_back((int32_t)v);
      inCand[v] = 1;
    }
  }
  for (int64_t round = 0;; round++) {
    picked.clear();
    for (int32_t v : cand) {
      if (!alive[v] || deg[v] > 2 || blockedAt[v] == round) continue;
      Pick p{(int32_t)v, -1, -1};
      for (int32_t u : adj[v]) {
        if (!alive[u]) continue;
        (p.a < 0 ? p.a : p.b) = u;
      }
      picked.push_back(p);
      if (p.a >= 0) blockedAt[p.a] = round;
      if (p.b >= 0) blockedAt[p.b] = round;
    }
    if ((int64_t)picked.size() < minRound) break;
    if (!haveEdges) {
      for (int64_t v = 0; v < n; v++) {
        for (int32_t u : adj[v]) {
          if (u > v) edges.insert(key((int32_t)v, u));
        }
      }
      haveEdges = true;
    }
    for (const Pick& p : picked) {
      alive[p.v] = 0;
      if (p.a >= 0) deg[p.a]--;
      if (p.b >= 0) deg[p.b]--;
      if (p.a >= 0 && p.b >= 0 && edges.insert(key(p.a, p.b)).second) {
        adj[p.a].push_back(p.b);
        adj[p.b].push_back(p.a);
        deg[p.a]++;
        deg[p.b]++;
      }
      std::vector<int32_t>().swap(adj[p.v]);
      order.push_back(p.v);
    }
    // next round's candidates: this round's survivors plus the neighbours whose degree just dropped
    nextCand.clear();
    for (int32_t v : cand) {
      if (alive[v] && deg[v] <= 2) {
        nextCand.push_back(v);
      } else {
        inCand[v] = 0;
      }
    }
    for (const Pick& p : picked) {
      for (int32_t u : {p.a, p.b}) {
        if (u >= 0 && alive[u] && deg[u] <= 2 && !inCand[u]) {
          inCand[u] = 1;
          nextCand.push_back(u);
        }
      }
    }
    // (the scan order of a round decides which of two adjacent candidates is taken: keep it
    //  ascending, as the full scan was, so that the ordering -- and every plan -- is unchanged)
    std::sort(nextCand.begin(), nextCand.end());
    cand.swap(nextCand);
  }
  return order;
}

static std::vector<int64_t> minimumDegreeCore(const std::vector<int64_t>& ptrs,
                                              const std::vector<int64_t>& inds);

std::vector<int64_t> minimumDegreeOrdering(const std::vector<int64_t>& ptrs,
                                           const std::vector<int64_t>& inds,
                                           int64_t chainContractionMinRound) {
  const int64_t n = (int64_t)ptrs.size() - 1;
  if (n <= 0) return {};
  BASPACHO_CHECK_LT(n, (int64_t)INT32_MAX);
  if (chainContractionMinRound <= 0) return minimumDegreeCore(ptrs, inds);

  std::vector<std::vector<int32_t>> adj(n);
  for (int64_t i = 0; i < n; i++) {
    for (int64_t k = ptrs[i]; k < ptrs[i + 1]; k++) {
      const int64_t j = inds[k];
      BASPACHO_CHECK_LT(j, n);
      if (j == i) continue;
      adj[i].push_back((int32_t)j);
      adj[j].push_back((int32_t)i);
    }
  }
  for (auto& a : adj) {
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
  }
  std::vector<uint8_t> alive(n, 1);
  std::vector<int64_t> perm = contractChains(adj, alive, chainContractionMinRound);
  if (perm.empty()) return minimumDegreeCore(ptrs, inds);

  // min-degree on what is left (with the fill edges of the contraction)
  std::vector<int64_t> newId(n, -1), oldId;
  for (int64_t i = 0; i < n; i++) {
    if (alive[i]) {
      newId[i] = (int64_t)oldId.size();
      oldId.push_back(i);
    }
  }
  std::vector<int64_t> rPtrs(1, 0), rInds;
  for (int64_t i : oldId) {
    for (int32_t j : adj[i]) {
      if (j < i && alive[j]) rInds.push_back(newId[j]);  // lower triangle is enough, the core symmetrises
    }
    rPtrs.push_back((int64_t)rInds.size());
  }
  for (int64_t q : minimumDegreeCore(rPtrs, rInds)) perm.push_back(oldId[q]);
  BASPACHO_CHECK_EQ((int64_t)perm.size(), n);
  return perm;
}

static std::vector<int64_t> minimumDegreeCore(const std::vector<int64_t>& ptrs,
                                              const std::vector<int64_t>& inds) {
  const int64_t n = (int64_t)ptrs.size() - 1;
  std::vector<int64_t> perm;
  perm.reserve(n);
  if (n <= 0) return perm;
  BASPACHO_CHECK_LT(n, (int64_t)INT32_MAX);

  QuotientGraph g(n);

  // symmetrised adjacency without the diagonal, duplicates removed
  for (int64_t i = 0; i < n; i++) {
    for (int64_t k = ptrs[i]; k < ptrs[i + 1]; k++) {
      int64_t j = inds[k];
      BASPACHO_CHECK_LT(j, n);
      if (j == i) continue;
      g.adjVars[i].push_back((int32_t)j);
      g.adjVars[j].push_back((int32_t)i);
    }
  }
  for (int64_t i = 0; i < n; i++) {
    auto& a = g.adjVars[i];
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
    g.degree[i] = (int64_t)a.size();
  }
  g.minDeg = n;
  for (int64_t i = n - 1; i >= 0; i--) g.bucketInsert((int32_t)i);

  int64_t numEliminated = 0;
  std::vector<int32_t> Lp, hashBucketHead(n, -1), hashNext(n, -1);
  std::vector<int64_t> hashOf(n, 0);
  std::vector<int32_t> touchedHashes;

  auto emit = [&](int32_t p) {
    perm.push_back(p);
    // iterative expansion of merged variables
    std::vector<int32_t> stack(g.members[p].rbegin(), g.members[p].rend());
    while (!stack.empty()) {
      int32_t v = stack.back();
      stack.pop_back();
      perm.push_back(v);
      for (auto it = g.members[v].rbegin(); it != g.members[v].rend(); ++it) stack.push_back(*it);
    }
  };

  while (numEliminated < n) {
    // ---- select pivot of minimum approximate degree
    while (g.minDeg < n && g.head[g.minDeg] < 0) g.minDeg++;
    BASPACHO_CHECK_LT(g.minDeg, n + 1);
    const int32_t p = g.head[g.minDeg];
    BASPACHO_CHECK_GE(p, 0);
    g.bucketRemove(p);

    // ---- form the pattern L_p of the new element
    Lp.clear();
    const int64_t stamp = ++g.markStamp;
    g.mark[p] = stamp;
    int64_t weightLp = 0;
    auto addVar = [&](int32_t v) {
      if (g.state[v] != kVar || g.nv[v] == 0 || g.mark[v] == stamp) return;
      g.mark[v] = stamp;
      Lp.push_back(v);
      weightLp += g.nv[v];
    };
    for (int32_t v : g.adjVars[p]) addVar(v);
    for (int32_t e : g.adjElems[p]) {
      if (g.state[e] != kElem) continue;
      for (int32_t v : g.elemVars[e]) addVar(v);
      g.state[e] = kDeadElem;  // absorbed into p
      std::vector<int32_t>().swap(g.elemVars[e]);
    }
    std::vector<int32_t>().swap(g.adjVars[p]);
    std::vector<int32_t>().swap(g.adjElems[p]);
    g.state[p] = kElem;
    numEliminated += g.nv[p];

    // ---- pass 1: w[e] - wflg = weighted |L_e \ L_p| for every element touching L_p
    if (g.wflg > (int64_t(1) << 60)) {
      std::fill(g.w.begin(), g.w.end(), 0);
      g.wflg = 1;
    }
    const int64_t wflg = g.wflg;
    for (int32_t i : Lp) {
      g.bucketRemove(i);
      for (int32_t e : g.adjElems[i]) {
        if (g.state[e] != kElem) continue;
        if (g.w[e] < wflg) g.w[e] = g.degree[e] + wflg;
        g.w[e] -= g.nv[i];
      }
    }

    // ---- pass 2: prune lists, approximate degrees, hashes, mass elimination
    touchedHashes.clear();
    size_t keepLp = 0;
    for (size_t idx = 0; idx < Lp.size(); idx++) {
      const int32_t i = Lp[idx];
      int64_t deg = 0;
      uint64_t hash = 0;
      auto& ae = g.adjElems[i];
      size_t ke = 0;
      for (int32_t e : ae) {
        if (g.state[e] != kElem) continue;
        int64_t ext = g.w[e] - wflg;
        if (ext > 0) {
          deg += ext;
          hash += (uint64_t)e;
          ae[ke++] = e;
        } else {
          // aggressive absorption: L_e is contained in L_p
          g.state[e] = kDeadElem;
          std::vector<int32_t>().swap(g.elemVars[e]);
        }
      }
      ae.resize(ke);
      auto& av = g.adjVars[i];
      size_t kv = 0;
      for (int32_t v : av) {
        if (g.state[v] != kVar || g.nv[v] == 0 || g.mark[v] == stamp) continue;
        deg += g.nv[v];
        hash += (uint64_t)v;
        av[kv++] = v;
      }
      av.resize(kv);

      if (ke == 0 && kv == 0) {
        // mass elimination: i is adjacent to nothing but L_p, eliminate it with p
        g.members[p].push_back(i);
        g.nv[p] += g.nv[i];
        numEliminated += g.nv[i];
        weightLp -= g.nv[i];
        g.nv[i] = 0;
        g.state[i] = kMergedVar;
        std::vector<int32_t>().swap(ae);
        std::vector<int32_t>().swap(av);
        continue;
      }
      ae.push_back(p);
      hash += (uint64_t)p;
      g.degree[i] = std::min(g.degree[i], deg);  // combined with |L_p| below
      hashOf[i] = (int64_t)(hash % (uint64_t)n);
      Lp[keepLp++] = i;
    }
    Lp.resize(keepLp);

    // ---- supervariable detection among L_p (indistinguishable nodes)
    for (int32_t i : Lp) {
      int64_t h = hashOf[i];
      if (hashBucketHead[h] < 0) touchedHashes.push_back((int32_t)h);
      hashNext[i] = hashBucketHead[h];
      hashBucketHead[h] = i;
    }
    for (int32_t h : touchedHashes) {
      for (int32_t i = hashBucketHead[h]; i >= 0; i = hashNext[i]) {
        if (g.nv[i] == 0) continue;
        const auto& aei = g.adjElems[i];
        const auto& avi = g.adjVars[i];
        const int64_t st = ++g.markStamp;
        // note: markStamp values used here never equal `stamp` of L_p again, which is fine
        // because the L_p membership test is no longer needed after pass 2.
        for (int32_t e : aei) g.w[e] = -st;  // negative stamps never collide with wflg values
        for (int32_t v : avi) g.mark[v] = st;
        int32_t prevJ = i;
        for (int32_t j = hashNext[i]; j >= 0; j = hashNext[j]) {
          bool same = g.nv[j] != 0 && g.adjElems[j].size() == aei.size() &&
                      g.adjVars[j].size() == avi.size();
          if (same) {
            for (int32_t e : g.adjElems[j]) {
              if (g.w[e] != -st) {
                same = false;
                break;
              }
            }
          }
          if (same) {
            for (int32_t v : g.adjVars[j]) {
              if (g.mark[v] != st) {
                same = false;
                break;
              }
            }
          }
          if (same) {
            g.members[i].push_back(j);
            g.nv[i] += g.nv[j];
            g.nv[j] = 0;
            g.state[j] = kMergedVar;
            std::vector<int32_t>().swap(g.adjElems[j]);
            std::vector<int32_t>().swap(g.adjVars[j]);
            hashNext[prevJ] = hashNext[j];  // unlink j
          } else {
            prevJ = j;
          }
        }
        // restore w for the elements we stamped (they must look "untouched" next round)
        for (int32_t e : aei) g.w[e] = 0;
      }
      hashBucketHead[h] = -1;
    }

    // ---- finalise: degrees, bucket insertion, element pattern
    auto& pattern = g.elemVars[p];
    pattern.clear();
    int64_t patternWeight = 0;
    for (int32_t i : Lp) {
      if (g.nv[i] == 0) continue;
      pattern.push_back(i);
      patternWeight += g.nv[i];
    }
    g.degree[p] = patternWeight;
    const int64_t remaining = n - numEliminated;
    for (int32_t i : pattern) {
      int64_t d = g.degree[i] + patternWeight - g.nv[i];
      d = std::min(d, remaining - g.nv[i]);
      g.degree[i] = std::max<int64_t>(d, 0);
      g.bucketInsert(i);
    }
    // all w stamps of this round are below the next flag
    g.wflg += 2 * n + 2;

    emit(p);
  }

  BASPACHO_CHECK_EQ((int64_t)perm.size(), n);
  return perm;
}

}  // namespace BaSpaCho
