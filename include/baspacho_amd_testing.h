/* TESTING entry points of libbaspacho_amd.so -- NOT part of the drop-in boundary.  Kept out of
 * include/baspacho_amd.h (the header a caller installs) so that no product code links against a hook
 * that deliberately breaks factor(); tests/ and baspacho_amd/__init__.py (Solver._testSetFault) are
 * the only users. */
#ifndef BASPACHO_AMD_TESTING_H
#define BASPACHO_AMD_TESTING_H

#include "baspacho_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fault injection: kind 1 = factor() skips the sparse-elimination update (Solver.cpp:190-196,
 * doElimination's update half), so the factor is WRONG -- tests/test_full_size_gpu.py checks that the
 * full-size parity checks then fail; 0 = off.  Per solver, not reachable through the environment;
 * the library prints a warning to stderr the first time an injected fault takes effect. */
int bsp_test_set_fault(bsp_solver* s, int32_t kind);

#ifdef __cplusplus
}
#endif

#endif
