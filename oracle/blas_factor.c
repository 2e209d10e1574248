/* ORACLE / CPU BASELINE (test + measurement infrastructure only; never linked into the product).
   Restatement of the reference's BLAS backend "BackendFast" (baspacho/baspacho/MatOpsFast.cpp)
   on top of the same driver loop (Solver.cpp:164-219), fp64:
     potrf        LAPACK dpotrf 'U' col-major == lower row-major          MatOpsFast.cpp:242-255
     trsm         dtrsm Left/Upper/Trans/NonUnit, m=n, n=k, lda=ldb=n       MatOpsFast.cpp:258-279
     saveSyrkGemm dsyrk(U,T) + dgemm(T,N), rule (m==n)||(m+n+k>150)        MatOpsFast.cpp:308-334
     assemble     threaded over block rows (chunks of 3)                  MatOpsFast.cpp:159-226
     doElimination threaded factorLump + per-target-row eliminateRowChain MatOpsFast.cpp:83-148,
                  prepareElimination                                      MatOpsCpuBase.h:74-114,
                  eliminateRowChain / elimDiagBlock / elimBlock            MatOpsCpuBase.h:231-319
   BLAS/LAPACK are resolved at run time with dlopen (OpenBLAS; the scipy wheel bundles one with
   `scipy_`-prefixed Fortran symbols).  Threads: OpenMP here, plus OpenBLAS' own pool. */
#include <dlfcn.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orc_skel.h"

typedef void (*dpotrf_t)(const char*, const int*, double*, const int*, int*);
typedef void (*dtrsm_t)(const char*, const char*, const char*, const char*, const int*, const int*,
                        const double*, const double*, const int*, double*, const int*);
typedef void (*dsyrk_t)(const char*, const char*, const int*, const int*, const double*,
                        const double*, const int*, const double*, double*, const int*);
typedef void (*dgemm_t)(const char*, const char*, const int*, const int*, const int*,
                        const double*, const double*, const int*, const double*, const int*,
                        const double*, double*, const int*);
typedef void (*setthreads_t)(int);
typedef char* (*getconfig_t)(void);

static dpotrf_t p_dpotrf;
static dtrsm_t p_dtrsm;
static dsyrk_t p_dsyrk;
static dgemm_t p_dgemm;
static setthreads_t p_setthreads;
static char g_blas_desc[512] = "unloaded";

static void* sym2(void* h, const char* a, const char* b) {
  void* p = dlsym(h, a);
  return p ? p : dlsym(h, b);
}

/* returns 0 on success */
int orc_blas_init(const char* path, int num_threads) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    snprintf(g_blas_desc, sizeof g_blas_desc, "dlopen failed: %s", dlerror());
    return 1;
  }
  p_dpotrf = (dpotrf_t)sym2(h, "scipy_dpotrf_", "dpotrf_");
  p_dtrsm = (dtrsm_t)sym2(h, "scipy_dtrsm_", "dtrsm_");
  p_dsyrk = (dsyrk_t)sym2(h, "scipy_dsyrk_", "dsyrk_");
  p_dgemm = (dgemm_t)sym2(h, "scipy_dgemm_", "dgemm_");
  p_setthreads = (setthreads_t)sym2(h, "scipy_openblas_set_num_threads", "openblas_set_num_threads");
  getconfig_t cfg = (getconfig_t)sym2(h, "scipy_openblas_get_config", "openblas_get_config");
  if (!p_dpotrf || !p_dtrsm || !p_dsyrk || !p_dgemm) {
    snprintf(g_blas_desc, sizeof g_blas_desc, "BLAS symbols missing in %s", path);
    return 2;
  }
  if (p_setthreads && num_threads > 0) p_setthreads(num_threads);
  if (num_threads > 0) omp_set_num_threads(num_threads);
  snprintf(g_blas_desc, sizeof g_blas_desc, "%s", cfg ? cfg() : path);
  return 0;
}

const char* orc_blas_desc(void) { return g_blas_desc; }

static void bl_potrf(int64_t n, double* A) {
  int N = (int)n, info = 0;
  p_dpotrf("U", &N, A, &N, &info);
}

static void bl_trsm(int64_t n, int64_t k, const double* A, double* B) {
  int N = (int)n, K = (int)k;
  double one = 1.0;
  p_dtrsm("L", "U", "T", "N", &N, &K, &one, A, &N, B, &N);
}

static void bl_save_syrk_gemm(int64_t m, int64_t n, int64_t k, const double* P, double* C) {
  int M = (int)m, K = (int)k;
  double one = 1.0, zero = 0.0;
  int doSyrk = (m == n) || (m + n + k > 150);
  int doGemm = !(doSyrk && m == n);
  if (doSyrk) p_dsyrk("U", "T", &M, &K, &one, P, &K, &zero, C, &M);
  if (doGemm) {
    int64_t start = doSyrk ? m : 0;
    int N2 = (int)(n - start);
    p_dgemm("T", "N", &M, &N2, &K, &one, P, &K, P + (doSyrk ? m * k : 0), &K, &zero,
            C + (doSyrk ? m * m : 0), &M);
  }
}

static void bl_factor_lump(const orc_skel* sk, double* data, int64_t lump) {
  int64_t n = sk->lumpStart[lump + 1] - sk->lumpStart[lump];
  int64_t c0 = sk->chainColPtr[lump];
  int64_t b0 = sk->boardColPtr[lump], b1 = sk->boardColPtr[lump + 1];
  int64_t belowOrd = sk->boardChainColOrd[b0 + 1], nChains = sk->boardChainColOrd[b1 - 1];
  int64_t rows = sk->chainRowsTillEnd[c0 + nChains - 1] - sk->chainRowsTillEnd[c0 + belowOrd - 1];
  double* D = data + sk->chainData[c0];
  bl_potrf(n, D);
  if (rows > 0) bl_trsm(n, rows, D, data + sk->chainData[c0 + belowOrd]);
}

/* small-lump variant used inside doElimination (MatOpsCpuBase.h:161-183 uses Eigen there) */
static void small_factor_lump(const orc_skel* sk, double* data, int64_t lump) {
  int64_t n = sk->lumpStart[lump + 1] - sk->lumpStart[lump];
  int64_t c0 = sk->chainColPtr[lump];
  int64_t b0 = sk->boardColPtr[lump], b1 = sk->boardColPtr[lump + 1];
  int64_t belowOrd = sk->boardChainColOrd[b0 + 1], nChains = sk->boardChainColOrd[b1 - 1];
  int64_t rows = sk->chainRowsTillEnd[c0 + nChains - 1] - sk->chainRowsTillEnd[c0 + belowOrd - 1];
  double* A = data + sk->chainData[c0];
  for (int64_t i = 0; i < n; i++) {
    double d = sqrt(A[i * n + i]);
    A[i * n + i] = d;
    for (int64_t j = i + 1; j < n; j++) {
      double c = A[j * n + i] / d;
      A[j * n + i] = c;
      for (int64_t k = i + 1; k <= j; k++) A[j * n + k] -= c * A[k * n + i];
    }
  }
  double* B = data + sk->chainData[c0 + belowOrd];
  for (int64_t r = 0; r < rows; r++) {
    double* v = B + r * n;
    for (int64_t i = 0; i < n; i++) {
      double x = v[i];
      for (int64_t j = 0; j < i; j++) x -= A[i * n + j] * v[j];
      v[i] = x / A[i * n + i];
    }
  }
}

/* prepareElimination (MatOpsCpuBase.h:74-114): chains of the columns [lb,le) grouped by row span */
typedef struct {
  int64_t spanRowBegin, numRows;
  int64_t *rowPtr, *colLump, *chainColOrd;
} elim_ctx;

static elim_ctx prepare_elimination(const orc_skel* sk, int64_t lb, int64_t le) {
  elim_ctx e;
  e.spanRowBegin = sk->lumpToSpan[le];
  e.numRows = sk->numSpans - e.spanRowBegin;
  e.rowPtr = (int64_t*)calloc((size_t)e.numRows + 1, sizeof(int64_t));
  for (int64_t l = lb; l < le; l++) {
    for (int64_t i = sk->chainColPtr[l]; i < sk->chainColPtr[l + 1]; i++) {
      int64_t s = sk->chainRowSpan[i];
      if (s >= e.spanRowBegin) e.rowPtr[s - e.spanRowBegin + 1]++;
    }
  }
  for (int64_t r = 0; r < e.numRows; r++) e.rowPtr[r + 1] += e.rowPtr[r];
  int64_t tot = e.rowPtr[e.numRows];
  e.colLump = (int64_t*)malloc(sizeof(int64_t) * (size_t)(tot + 1));
  e.chainColOrd = (int64_t*)malloc(sizeof(int64_t) * (size_t)(tot + 1));
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e.numRows + 1));
  memcpy(cur, e.rowPtr, sizeof(int64_t) * (size_t)(e.numRows + 1));
  for (int64_t l = lb; l < le; l++) {
    int64_t i0 = sk->chainColPtr[l];
    for (int64_t i = i0; i < sk->chainColPtr[l + 1]; i++) {
      int64_t s = sk->chainRowSpan[i];
      if (s < e.spanRowBegin) continue;
      int64_t slot = cur[s - e.spanRowBegin]++;
      e.colLump[slot] = l;
      e.chainColOrd[slot] = i - i0;
    }
  }
  free(cur);
  return e;
}

/* eliminateRowChain (MatOpsCpuBase.h:267-319) */
static void eliminate_row_chain(const elim_ctx* e, const orc_skel* sk, double* data, int64_t sRel,
                                int64_t* spanToChainOffset) {
  int64_t s = sRel + e->spanRowBegin;
  if (e->rowPtr[sRel] == e->rowPtr[sRel + 1]) return;
  int64_t t = sk->spanToLump[s];
  int64_t tSize = sk->lumpStart[t + 1] - sk->lumpStart[t];
  int64_t offInLump = sk->spanStart[s] - sk->lumpStart[t];
  for (int64_t i = sk->chainColPtr[t]; i < sk->chainColPtr[t + 1]; i++) {
    spanToChainOffset[sk->chainRowSpan[i]] = sk->chainData[i];
  }
  for (int64_t i = e->rowPtr[sRel]; i < e->rowPtr[sRel + 1]; i++) {
    int64_t lump = e->colLump[i];
    int64_t ptrStart = sk->chainColPtr[lump] + e->chainColOrd[i];
    int64_t ptrEnd = sk->chainColPtr[lump + 1];
    int64_t rowsAbove = sk->chainRowsTillEnd[ptrStart - 1];
    int64_t m = sk->chainRowsTillEnd[ptrStart] - rowsAbove;
    int64_t k = sk->lumpStart[lump + 1] - sk->lumpStart[lump];
    const double* B0 = data + sk->chainData[ptrStart];
    /* elimDiagBlock: lower triangle of the diagonal target block */
    double* T = data + offInLump + spanToChainOffset[s];
    for (int64_t a = 0; a < m; a++) {
      for (int64_t b = 0; b <= a; b++) {
        double v = 0;
        for (int64_t q = 0; q < k; q++) v += B0[a * k + q] * B0[b * k + q];
        T[a * tSize + b] -= v;
      }
    }
    /* elimBlock for every chain below */
    for (int64_t p = ptrStart + 1; p < ptrEnd; p++) {
      int64_t s2 = sk->chainRowSpan[p];
      int64_t rows2 = sk->chainRowsTillEnd[p] - sk->chainRowsTillEnd[p - 1];
      const double* B2 = data + sk->chainData[p];
      double* T2 = data + offInLump + spanToChainOffset[s2];
      for (int64_t a = 0; a < rows2; a++) {
        for (int64_t b = 0; b < m; b++) {
          double v = 0;
          for (int64_t q = 0; q < k; q++) v += B2[a * k + q] * B0[b * k + q];
          T2[a * tSize + b] -= v;
        }
      }
    }
  }
}

static void bl_do_elimination(const orc_skel* sk, double* data, int64_t lb, int64_t le) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t l = lb; l < le; l++) small_factor_lump(sk, data, l);
  elim_ctx e = prepare_elimination(sk, lb, le);
#pragma omp parallel
  {
    int64_t* s2c = (int64_t*)malloc(sizeof(int64_t) * (size_t)(sk->numSpans + 1));
#pragma omp for schedule(dynamic, 5)
    for (int64_t r = 0; r < e.numRows; r++) eliminate_row_chain(&e, sk, data, r, s2c);
    free(s2c);
  }
  free(e.rowPtr);
  free(e.colLump);
  free(e.chainColOrd);
}

static void bl_assemble(const orc_skel* sk, double* data, const double* temp,
                        const int64_t* spanToChainOffset, int64_t rectRowBegin, int64_t dstStride,
                        int64_t srcColDataOffset, int64_t srcRectWidth, int64_t numBlockRows,
                        int64_t numBlockCols) {
  const int64_t* cre = sk->chainRowsTillEnd + srcColDataOffset;
  const int64_t* toSpan = sk->chainRowSpan + srcColDataOffset;
#pragma omp parallel for schedule(dynamic, 3) if (numBlockRows > 6)
  for (int64_t r = 0; r < numBlockRows; r++) {
    int64_t rBegin = cre[r - 1] - rectRowBegin;
    int64_t rSize = cre[r] - rBegin - rectRowBegin;
    int64_t rOffset = spanToChainOffset[toSpan[r]];
    const double* rowPtr = temp + rBegin * srcRectWidth;
    int64_t cEnd = numBlockCols < r + 1 ? numBlockCols : r + 1;
    for (int64_t c = 0; c < cEnd; c++) {
      int64_t cStart = cre[c - 1] - rectRowBegin;
      int64_t cSize = cre[c] - cStart - rectRowBegin;
      double* dst = data + rOffset + sk->spanOffsetInLump[toSpan[c]];
      const double* src = rowPtr + cStart;
      for (int64_t j = 0; j < rSize; j++) {
        for (int64_t i = 0; i < cSize; i++) dst[j * dstStride + i] -= src[j * srcRectWidth + i];
      }
    }
  }
}

/* Solver::factor (Solver.cpp:149-219) on the BLAS backend; elim_seconds (optional) receives the
   time spent in doElimination ("Point Schur-Elim Time", BaAtLargeBench.cpp:92-95) */
int orc_blas_factor_f64(const orc_skel* sk, const int64_t* ranges, int64_t nRanges, double* data,
                        double* elim_seconds) {
  if (!p_dpotrf) return -3;
  double t0 = omp_get_wtime();
  for (int64_t r = 0; r + 1 < nRanges; r++) bl_do_elimination(sk, data, ranges[r], ranges[r + 1]);
  if (elim_seconds) *elim_seconds = omp_get_wtime() - t0;
  int64_t denseFrom = nRanges > 0 ? ranges[nRanges - 1] : 0;
  int64_t maxTemp = 1;
  for (int64_t l = denseFrom; l < sk->numLumps; l++) {
    for (int64_t p = sk->boardRowPtr[l]; p < sk->boardRowPtr[l + 1] - 1; p++) {
      int64_t src = sk->boardColLump[p], ord = sk->boardColOrd[p];
      if (src < denseFrom) continue;
      int64_t c0 = sk->chainColPtr[src], b0 = sk->boardColPtr[src], b1 = sk->boardColPtr[src + 1];
      int64_t belowOrd = sk->boardChainColOrd[b0 + ord];
      int64_t rb = sk->chainRowsTillEnd[c0 + belowOrd - 1];
      int64_t m = sk->chainRowsTillEnd[c0 + sk->boardChainColOrd[b0 + ord + 1] - 1] - rb;
      int64_t n = sk->chainRowsTillEnd[c0 + sk->boardChainColOrd[b1 - 1] - 1] - rb;
      if (m * n > maxTemp) maxTemp = m * n;
    }
  }
  double* temp = (double*)malloc(sizeof(double) * (size_t)maxTemp);
  int64_t* s2c = (int64_t*)malloc(sizeof(int64_t) * (size_t)(sk->numSpans + 1));
  if (!temp || !s2c) return -2;
  for (int64_t l = denseFrom; l < sk->numLumps; l++) {
    for (int64_t i = sk->chainColPtr[l]; i < sk->chainColPtr[l + 1]; i++) {
      s2c[sk->chainRowSpan[i]] = sk->chainData[i];
    }
    for (int64_t p = sk->boardRowPtr[l]; p < sk->boardRowPtr[l + 1] - 1; p++) {
      int64_t src = sk->boardColLump[p], ord = sk->boardColOrd[p];
      if (src < denseFrom) continue;
      int64_t srcSize = sk->lumpStart[src + 1] - sk->lumpStart[src];
      int64_t c0 = sk->chainColPtr[src], b0 = sk->boardColPtr[src], b1 = sk->boardColPtr[src + 1];
      int64_t belowOrd = sk->boardChainColOrd[b0 + ord];
      int64_t end0 = sk->boardChainColOrd[b0 + ord + 1], end1 = sk->boardChainColOrd[b1 - 1];
      int64_t rb = sk->chainRowsTillEnd[c0 + belowOrd - 1];
      int64_t m = sk->chainRowsTillEnd[c0 + end0 - 1] - rb;
      int64_t n = sk->chainRowsTillEnd[c0 + end1 - 1] - rb;
      bl_save_syrk_gemm(m, n, srcSize, data + sk->chainData[c0 + belowOrd], temp);
      int64_t tSize = sk->lumpStart[l + 1] - sk->lumpStart[l];
      bl_assemble(sk, data, temp, s2c, rb, tSize, c0 + belowOrd, m, end1 - belowOrd,
                  end0 - belowOrd);
    }
    bl_factor_lump(sk, data, l);
  }
  free(temp);
  free(s2c);
  return 0;
}
