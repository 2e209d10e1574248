"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's algorithm for the `factor()` / `solve()` hot path
(facebookresearch/baspacho v1).  Nothing here is part of the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this package.
The product (`baspacho_amd/`) never imports it and fails loudly without its HIP library.

Parity status: PINNED against the reference's own known-answer tests
  * literal skeleton arrays + densify matrices  (tests/CoalescedBlockMatrixTest.cpp:48-180)
  * literal transpose / symmetric-permutation   (tests/SparseStructureTest.cpp:20-63)
  * dense-Cholesky oracle protocol              (tests/FactorTest.cpp:43-107: data -> damp ->
    densify -> dense LLT vs factor -> densify, lower triangle, 1e-10 / 1e-8)
The reference itself cannot be compiled here (un-vendored Eigen/dispenso), so the
dense oracle is numpy.linalg.cholesky, as the reference's tests use Eigen::LLT.

Modules
  skel.py       skeleton constructor, densify, damp        (CoalescedBlockMatrix.cpp:17-187)
  structure.py  block-pattern helpers + naive fill          (SparseStructure.cpp, TestingUtils.cpp)
  ref_factor.c  numeric factor/solve, plain loops           (Solver.cpp:42-397, MatOpsRef.cpp,
                                                             MatOpsCpuBase.h, MathUtils.h)
  blas_factor.c same driver on OpenBLAS (dlopen)            (MatOpsFast.cpp:83-361)  -> cpu_baseline
  cref.py       ctypes loader/wrappers for the two C files
"""
