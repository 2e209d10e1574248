// Fill-reducing ordering: approximate minimum degree on a quotient graph.
// The reference delegates this step to a third-party dependency that is not vendored in
// /root/reference (Eigen 3.4.0 AMDOrdering, call site baspacho/baspacho/SparseStructure.cpp:313-331,
// or SuiteSparse amd_l_order, :297-309).  This file is an independent implementation of the
// published algorithm (Amestoy, Davis, Duff, "An approximate minimum degree ordering algorithm",
// SIAM J. Matrix Anal. Appl. 17(4), 1996): element absorption, approximate external degrees,
// mass elimination, hash-based supervariable detection and aggressive absorption.  Any valid
// permutation yields a correct factor; quality is pinned by the reference's bound
// (tests/SparseStructureTest.cpp:117-152).
#pragma once

#include <cstdint>
#include <vector>

namespace BaSpaCho {

// ptrs/inds: any (half or full) pattern of a symmetric matrix; it is symmetrised internally.
// returns perm with perm[k] = original index eliminated k-th.
// chainContractionMinRound > 0: before min-degree, rounds of independent pivots with at most two
// neighbours (odd-even reduction of chain-like parts, see min_degree.cpp) as long as a round
// holds at least that many pivots; 0 = plain min-degree.
std::vector<int64_t> minimumDegreeOrdering(const std::vector<int64_t>& ptrs,
                                           const std::vector<int64_t>& inds,
                                           int64_t chainContractionMinRound = 0);

}  // namespace BaSpaCho
