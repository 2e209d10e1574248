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
/* kind 2 = the spine workgroup of block 1 of every persistent solve sweep (csrc/hip_sweep_kernels.h)
 * never publishes its x, and the sweeps' watchdog is set to 50 ms: the launch must end by itself and
 * the next solve must report the failed call (tests/test_sweep_gpu.py). */

/* Developer aid: with BSP_SWEEP_TRACE=1 in the environment when the solver is created, every spine
 * workgroup of a persistent solve sweep stamps {start, operands on chip, inputs arrived, x published}
 * (wall clock, 100 MHz); this reads the stamps of the LAST sweep, 4 values per block. */
int bsp_test_read_sweep_trace(bsp_solver* s, long long* out, int32_t max_blocks, int32_t* n_blocks);

#ifdef __cplusplus
}
#endif

#endif
