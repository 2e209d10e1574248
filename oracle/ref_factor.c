/* ORACLE (test infrastructure only): numeric factor()/solve() of the reference, restated as
   plain C loops over the skeleton arrays (see ref_factor_impl.inc for file:line citations).
   Built by oracle/cref.py with `gcc -O2 -shared -fPIC`.  Never linked into the product. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "orc_skel.h"

/* bisect (Utils.h:155-166) */
static int64_t orc_bisect(const int64_t* a, int64_t size, int64_t needle) {
  int64_t lo = 0, hi = size;
  while (hi - lo > 1) {
    int64_t m = (lo + hi) / 2;
    if (needle >= a[m]) lo = m; else hi = m;
  }
  return lo;
}

#define REAL double
#define FN(name) name##_f64
#include "ref_factor_impl.inc"
#undef REAL
#undef FN

#define REAL float
#define FN(name) name##_f32
#include "ref_factor_impl.inc"
#undef REAL
#undef FN

/* the scalar kernels on their own, for the MathUtilsTest analogue (tests/MathUtilsTest.cpp:21-75):
   A row-major n x n with stride lda; v a row vector of n */
void orc_cholesky_f64(double* A, int64_t lda, int64_t n) { chol_f64(A, lda, n); }
void orc_solve_upper_t_f64(const double* L, int64_t lda, int64_t n, double* v) { solve_row_f64(L, lda, n, v); }
void orc_solve_upper_f64(const double* L, int64_t lda, int64_t n, double* v) { solve_row_t_f64(L, lda, n, v); }

/* Size-independent parity probe (test/measurement helper, no reference counterpart):
   out[0] = || L (L^T x) - A x ||_2, out[1] = || A x ||_2, with A (symmetric, lower triangle
   stored) and L applied as block-sparse operators laid out by the skeleton.  Strictly-upper
   entries of the square diagonal blocks are ignored (CoalescedBlockMatrix.h:23-37). */
int orc_probe_residual_f64(const orc_skel* sk, const double* A, const double* L, const double* x,
                           double* out) {
  int64_t n = sk->spanStart[sk->numSpans];
  double* ax = (double*)calloc((size_t)n, sizeof(double));
  double* y = (double*)calloc((size_t)n, sizeof(double));
  double* z = (double*)calloc((size_t)n, sizeof(double));
  if (!ax || !y || !z) return -2;
  for (int pass = 0; pass < 2; pass++) {
    for (int64_t l = 0; l < sk->numLumps; l++) {
      int64_t w = sk->lumpStart[l + 1] - sk->lumpStart[l], gc0 = sk->lumpStart[l];
      for (int64_t c = sk->chainColPtr[l]; c < sk->chainColPtr[l + 1]; c++) {
        int64_t span = sk->chainRowSpan[c];
        int64_t gr0 = sk->spanStart[span], rows = sk->spanStart[span + 1] - gr0;
        const double* a = A + sk->chainData[c];
        const double* f = L + sk->chainData[c];
        for (int64_t r = 0; r < rows; r++) {
          int64_t gr = gr0 + r;
          int64_t qEnd = gr - gc0 + 1 < w ? gr - gc0 + 1 : w; /* columns with gc <= gr */
          if (pass == 0) {
            double accA = 0, xr = x[gr];
            for (int64_t q = 0; q < qEnd; q++) {
              double av = a[r * w + q];
              accA += av * x[gc0 + q];
              if (gc0 + q != gr) ax[gc0 + q] += av * xr;
              y[gc0 + q] += f[r * w + q] * xr;
            }
            ax[gr] += accA;
          } else {
            double accZ = 0;
            for (int64_t q = 0; q < qEnd; q++) accZ += f[r * w + q] * y[gc0 + q];
            z[gr] += accZ;
          }
        }
      }
    }
  }
  double d = 0, na = 0;
  for (int64_t i = 0; i < n; i++) {
    d += (z[i] - ax[i]) * (z[i] - ax[i]);
    na += ax[i] * ax[i];
  }
  out[0] = sqrt(d);
  out[1] = sqrt(na);
  free(ax);
  free(y);
  free(z);
  return 0;
}

/* Test-input helper (no reference counterpart): out[i] = sum_{j != i} |A_ij| of the symmetric
   matrix stored by the skeleton (lower triangle; strictly-upper entries of the square diagonal
   blocks ignored).  tests/test_full_size_gpu.py uses it to build matrices that are diagonally
   dominant by a few percent only -- SPD, but with off-diagonal mass as large as the diagonal, so
   that a wrong update cannot hide behind the damping. */
int orc_abs_row_sums_f64(const orc_skel* sk, const double* A, double* out) {
  int64_t n = sk->spanStart[sk->numSpans];
  for (int64_t i = 0; i < n; i++) out[i] = 0;
  for (int64_t l = 0; l < sk->numLumps; l++) {
    int64_t w = sk->lumpStart[l + 1] - sk->lumpStart[l], gc0 = sk->lumpStart[l];
    for (int64_t c = sk->chainColPtr[l]; c < sk->chainColPtr[l + 1]; c++) {
      int64_t span = sk->chainRowSpan[c];
      int64_t gr0 = sk->spanStart[span], rows = sk->spanStart[span + 1] - gr0;
      const double* a = A + sk->chainData[c];
      for (int64_t r = 0; r < rows; r++) {
        int64_t gr = gr0 + r;
        int64_t qEnd = gr - gc0 < w ? gr - gc0 : w; /* columns with gc < gr */
        double acc = 0;
        for (int64_t q = 0; q < qEnd; q++) {
          double v = fabs(a[r * w + q]);
          acc += v;
          out[gc0 + q] += v;
        }
        out[gr] += acc;
      }
    }
  }
  return 0;
}
