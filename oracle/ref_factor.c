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
