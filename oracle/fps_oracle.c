/*
 * fps_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this file's
 * shared object.  Nothing under fletcherpenaltysolver.jl_amd/ (the product) may call it.
 *
 * What it is: a plain-C, single-threaded restatement of the reference's *iterative* penalty-evaluation
 * linear-solve path
 *     /root/reference/src/solve_linear_system.jl:45-140      (solve_two_extras / _least_squares / _mixed)
 *     /root/reference/src/solve_two_systems_struct.jl:167-244 (solve_least_square / solve_least_norm)
 *     /root/reference/src/model-Fletcherpenaltynlp.jl:234-252,352-437 (_compute_ys_gs!, obj, grad!, objgrad!)
 * with 64-bit indices like the reference's `Int`.
 *
 * The arithmetic of that path lives in a third-party package that is NOT under /root/reference:
 *     Krylov.jl, compat "0.10" (reference Project.toml:25; no Manifest is committed)
 *       lsqr!   (Paige & Saunders, ACM TOMS 8(1), 1982)             -> fpo_lsqr
 *       craig!  (Craig 1955; Paige 1974; Saunders 1995; SQD form: Arioli & Orban 2013) -> fpo_craig
 *       minres! (Paige & Saunders, SIAM J. Numer. Anal. 12(4), 1975) -> fpo_minres_aat
 *       sym_givens (Choi's SymOrtho)                                 -> sym_givens
 * Those are restated here from the published algorithms and from the package's documented recurrences
 * and stopping rules.  Julia is not installed in this pipeline, so Krylov.jl itself cannot be executed:
 *
 *     PARITY STATUS: iteration-level parity with Krylov.jl is UNPINNED (no golden vectors for the
 *     iterative back-end exist in the reference's tests, test/nlpmodelstest.jl:17-54 only checks
 *     self-consistency).  What IS pinned: the solutions (p1,q1,p2,q2), ys and grad(phi) against the
 *     reference's own known-answer tests (test/unit-test.jl:16-152, LDLt back-end, atol 1e-13..1e-14)
 *     -- see tests/golden/ and tests/test_oracle.py -- and against an exact dense/sparse KKT solve
 *     (oracle/oracle.py), which is what both reference back-ends approximate.
 *
 * Build: make -C oracle   (gcc -O3 -shared -fPIC)
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t m, n;           /* A is m x n */
  const int64_t *rowptr;  /* m+1, 0-based */
  const int64_t *colind;  /* nnz, 0-based */
  const double *vals;
} fpo_csr;

/* mirrors the fields of Krylov.jl's stats read at solve_two_systems_struct.jl:183,242 and
 * solve_linear_system.jl:71 */
typedef struct {
  int32_t solved;
  int32_t inconsistent;
  int32_t niter;
  int32_t status; /* see FPO_ST_* */
  double rnorm;   /* final residual estimate */
  double arnorm;  /* final ||A'r|| estimate (lsqr/minres) */
} fpo_stats;

enum {
  FPO_ST_UNKNOWN = 0,
  FPO_ST_ZERO_RHS = 1,    /* "x = 0 is a zero-residual solution" */
  FPO_ST_ZERO_ATB = 2,    /* "x = 0 is a minimum least-squares solution" */
  FPO_ST_SOLVED = 3,      /* tolerance met */
  FPO_ST_ZERO_RESID = 4,  /* zero-residual solution */
  FPO_ST_FWD_ERR = 5,     /* truncated forward error small enough */
  FPO_ST_ILL_COND = 6,    /* condition number limit */
  FPO_ST_MAXITER = 7,     /* maximum number of iterations exceeded */
  FPO_ST_INCONSISTENT = 8, /* system may be inconsistent (craig) */
  FPO_ST_SOLVED_LQ = 9     /* lnlq: "solutions xL and yL good enough" (FPO_ST_SOLVED = the CRAIG point xC, yC) */
};

typedef struct {
  double ls_atol, ls_rtol;
  int64_t ls_itmax;
  double ln_atol, ln_rtol, ln_btol, ln_conlim;
  int64_t ln_itmax;
  double ne_atol, ne_rtol, ne_etol;
  int64_t ne_itmax;
  double ne_conlim;
  /* Krylov.jl lsqr! keyword arguments that the reference leaves at their defaults (sqrt(eps), 1/sqrt(eps));
   * exposed so tests can run the same recurrences to tighter accuracy than the defaults allow */
  double ls_axtol, ls_btol, ls_etol, ls_conlim;
  /* least-norm method of solve_two_mixed: 0 = craig! (the reference's default workspace, struct.jl:121),
   * 1 = lnlq! through the generic solve_least_norm (the commented alternative of struct.jl:121; :251-281) */
  int32_t ln_method;
  int32_t pad;
} fpo_options;

/* defaults of IterativeSolver, solve_two_systems_struct.jl:99-115 */
void fpo_default_options(int64_t n, int64_t m, fpo_options *o) {
  const double se = sqrt(2.220446049250313e-16);
  o->ls_atol = se;
  o->ls_rtol = se;
  o->ls_itmax = 5 * (m + n);
  o->ln_atol = se;
  o->ln_rtol = se;
  o->ln_btol = se;
  o->ln_conlim = 1.0 / se;
  o->ln_itmax = 5 * (m + n);
  o->ne_atol = se;
  o->ne_rtol = se;
  o->ne_etol = se;
  o->ne_itmax = 0;
  o->ne_conlim = 1.0 / se;
  o->ls_axtol = se;
  o->ls_btol = se;
  o->ls_etol = se;
  o->ls_conlim = 1.0 / se;
  o->ln_method = 0;
  o->pad = 0;
}

/* ------------------------------------------------------------------ basic linear algebra */

/* y = A x   (the body of jprod! for a linear-constraint model) */
/* FPO_OMP (libfps_oracle_omp.so, the all-cores CPU baseline of bench.py only): the loops below run under OpenMP.  The
 * parity oracle is the serial build -- the threaded sums associate differently. */
#ifdef FPO_OMP
#include <omp.h>
#define FPO_PRAGMA(x) _Pragma(#x)
#else
#define FPO_PRAGMA(x)
#endif

/* ---- summation-order variants (fpo_set_sum_order; tests only).  The restatement sums every row left to right, as a
 * serial CPU code does; the device sums a LONG row in another -- equally valid -- order.  Where a recurrence is sensitive to
 * that (a dominant dense row or column: tests/test_gpu_parity.py, the awkward structures), the question "is the device's
 * different iteration count a different summation order or a bug" is answered by running the restatement itself in the
 * device's order for those rows:
 *   1 = FPO_SUM_DEVICE: a row of A with more than 2048 entries (a row group of its own in the device's column-sorted layout,
 *       csrc/fpsq_spmv.hip.h k_spmv_rgcs: one wave, tiles of 2048 entries in column order) is summed by 64 strided
 *       accumulators that add four products at a time, (d0 + d1) + (d2 + d3), tile after tile, then a halving tree over the
 *       64; a row of A' with more than 2048 entries (k_spmv's long-row branch) by 256 strided fused-multiply-add
 *       accumulators, a halving tree inside every 64 and the four results left to right.  Shorter rows stay as they are.
 *   2 = FPO_SUM_REVERSED: every row right to left (a plain perturbation of the order, for sensitivity checks).
 * 0 = the default everywhere else. */
static int g_sum_order = 0;
void fpo_set_sum_order(int mode) { g_sum_order = mode; }
/* developer aid: lsqr prints, per iteration, every stopping quantity over its threshold (how close a stop was) */
static int g_trace = 0;
void fpo_set_trace(int on) { g_trace = on; }
#define FPO_LONG_ROW 2048

static double tree64(double *a) { /* a[i] += a[i + off], off = 32 .. 1: lane 0 of a wave's shuffle-down reduction */
  for (int off = 32; off > 0; off >>= 1)
    for (int i = 0; i < off; ++i) a[i] += a[i + off];
  return a[0];
}

/* one long row of A in the device's order; entries [k0, k1) must be sorted by column (they are tiled in that order) */
static double long_row_device_a(const fpo_csr *A, int64_t k0, int64_t k1, const double *x) {
  double acc[64];
  for (int l = 0; l < 64; ++l) acc[l] = 0.0;
  for (int64_t t0 = k0; t0 < k1; t0 += FPO_LONG_ROW) {
    const int64_t t1 = t0 + FPO_LONG_ROW < k1 ? t0 + FPO_LONG_ROW : k1;
    for (int l = 0; l < 64; ++l) {
      int64_t j = t0 + l;
      for (; j + 3 * 64 < t1; j += 4 * 64) {
        const double d0 = A->vals[j] * x[A->colind[j]], d1 = A->vals[j + 64] * x[A->colind[j + 64]];
        const double d2 = A->vals[j + 128] * x[A->colind[j + 128]], d3 = A->vals[j + 192] * x[A->colind[j + 192]];
        acc[l] += (d0 + d1) + (d2 + d3);
      }
      if (j < t1) {
        const double d0 = A->vals[j] * x[A->colind[j]];
        const double d1 = j + 64 < t1 ? A->vals[j + 64] * x[A->colind[j + 64]] : 0.0;
        const double d2 = j + 128 < t1 ? A->vals[j + 128] * x[A->colind[j + 128]] : 0.0;
        acc[l] += (d0 + d1) + d2;
      }
    }
  }
  return tree64(acc);
}

static void csr_mul(const fpo_csr *A, const double *x, double *y) {
  if (g_sum_order != 0) {
    for (int64_t i = 0; i < A->m; ++i) {
      const int64_t k0 = A->rowptr[i], k1 = A->rowptr[i + 1];
      double s = 0.0;
      if (g_sum_order == 2) {
        for (int64_t k = k1 - 1; k >= k0; --k) s += A->vals[k] * x[A->colind[k]];
      } else if (k1 - k0 > FPO_LONG_ROW) {
        s = long_row_device_a(A, k0, k1, x);
      } else {
        for (int64_t k = k0; k < k1; ++k) s += A->vals[k] * x[A->colind[k]];
      }
      y[i] = s;
    }
    return;
  }
  FPO_PRAGMA(omp parallel for schedule(static))
  for (int64_t i = 0; i < A->m; ++i) {
    double s = 0.0;
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) s += A->vals[k] * x[A->colind[k]];
    y[i] = s;
  }
}

/* y = A' u  (jtprod!) */
#ifdef FPO_OMP
/* Transposed copy for the threaded A'u (row-parallel on A'): built by fpo_omp_prepare for the arrays of the next
 * calls, dropped by fpo_omp_release.  Only the all-cores baseline uses it. */
static struct {
  const int64_t *rowptr;
  const double *vals;
  int64_t m, n;
  int64_t *trp, *tci;
  double *tv;
} g_tr = {0, 0, 0, 0, 0, 0, 0};

/* n > 0: use n threads from now on; returns the thread count in effect */
int fpo_omp_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}

void fpo_omp_release(void) {
  free(g_tr.trp);
  free(g_tr.tci);
  free(g_tr.tv);
  memset(&g_tr, 0, sizeof g_tr);
}

int fpo_omp_prepare(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals) {
  fpo_omp_release();
  const int64_t nnz = rowptr[m];
  int64_t *trp = (int64_t *)calloc((size_t)n + 2, sizeof(int64_t));
  int64_t *tci = (int64_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int64_t));
  double *tv = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
  if (!trp || !tci || !tv) {
    free(trp);
    free(tci);
    free(tv);
    return 1;
  }
  for (int64_t k = 0; k < nnz; ++k) trp[colind[k] + 2]++;
  for (int64_t j = 0; j < n; ++j) trp[j + 2] += trp[j + 1];
  for (int64_t i = 0; i < m; ++i)
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int64_t q = trp[colind[k] + 1]++;
      tci[q] = i;
      tv[q] = vals[k];
    }
  g_tr.rowptr = rowptr;
  g_tr.vals = vals;
  g_tr.m = m;
  g_tr.n = n;
  g_tr.trp = trp;
  g_tr.tci = tci;
  g_tr.tv = tv;
  return 0;
}
#endif

static void csr_tmul(const fpo_csr *A, const double *u, double *y) {
#ifdef FPO_OMP
  if (g_tr.trp && g_tr.rowptr == A->rowptr && g_tr.vals == A->vals && g_tr.m == A->m && g_tr.n == A->n) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < A->n; ++j) {
      double s = 0.0;
      for (int64_t k = g_tr.trp[j]; k < g_tr.trp[j + 1]; ++k) s += g_tr.tv[k] * u[g_tr.tci[k]];
      y[j] = s;
    }
    return;
  }
  /* row blocks scatter into per-thread copies of y, summed afterwards in thread order */
  const int T = omp_get_max_threads();
  double *buf = (double *)calloc((size_t)T * (size_t)A->n, sizeof(double));
  if (buf) {
#pragma omp parallel num_threads(T)
    {
      const int t = omp_get_thread_num();
      double *yt = buf + (size_t)t * (size_t)A->n;
      const int64_t lo = A->m * t / T, hi = A->m * (t + 1) / T;
      for (int64_t i = lo; i < hi; ++i) {
        const double ui = u[i];
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) yt[A->colind[k]] += A->vals[k] * ui;
      }
#pragma omp barrier
#pragma omp for schedule(static)
      for (int64_t j = 0; j < A->n; ++j) {
        double s = 0.0;
        for (int q = 0; q < T; ++q) s += buf[(size_t)q * (size_t)A->n + j];
        y[j] = s;
      }
    }
    free(buf);
    return;
  }
#endif
  memset(y, 0, (size_t)A->n * sizeof(double));
  if (g_sum_order == 2) { /* every row of A' right to left */
    for (int64_t i = A->m - 1; i >= 0; --i) {
      const double ui = u[i];
      for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) y[A->colind[k]] += A->vals[k] * ui;
    }
    return;
  }
  int64_t *cnt = NULL;
  if (g_sum_order == 1) { /* the long rows of A' (columns of A with more than 2048 entries) are left out of the scatter ... */
    cnt = (int64_t *)calloc((size_t)A->n, sizeof(int64_t));
    for (int64_t k = 0; k < A->rowptr[A->m]; ++k) cnt[A->colind[k]]++;
  }
  for (int64_t i = 0; i < A->m; ++i) {
    const double ui = u[i];
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
      if (!cnt || cnt[A->colind[k]] <= FPO_LONG_ROW) y[A->colind[k]] += A->vals[k] * ui;
  }
  if (cnt) { /* ... and summed in the device's order: 256 strided fma accumulators over the row's entries (rows of A ascending) */
    for (int64_t j = 0; j < A->n; ++j) {
      if (cnt[j] <= FPO_LONG_ROW) continue;
      double acc[256];
      for (int t = 0; t < 256; ++t) acc[t] = 0.0;
      int64_t pos = 0;
      for (int64_t i = 0; i < A->m; ++i)
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
          if (A->colind[k] == j) {
            acc[pos & 255] = fma(A->vals[k], u[i], acc[pos & 255]);
            ++pos;
          }
      const double w0 = tree64(acc), w1 = tree64(acc + 64), w2 = tree64(acc + 128), w3 = tree64(acc + 192);
      y[j] = w0 + w1 + w2 + w3;
    }
    free(cnt);
  }
}

/* an operator that is A or A' (the reference passes `nlp.Aop'` to lsqr, solve_linear_system.jl:123) */
typedef struct {
  const fpo_csr *A;
  int transposed;
} fpo_op;
static int64_t op_rows(const fpo_op *B) { return B->transposed ? B->A->n : B->A->m; }
static int64_t op_cols(const fpo_op *B) { return B->transposed ? B->A->m : B->A->n; }
static void op_mul(const fpo_op *B, const double *x, double *y) {
  if (B->transposed) csr_tmul(B->A, x, y); else csr_mul(B->A, x, y);
}
static void op_tmul(const fpo_op *B, const double *u, double *y) {
  if (B->transposed) csr_mul(B->A, u, y); else csr_tmul(B->A, u, y);
}

static double dotp(int64_t n, const double *a, const double *b) {
  double s = 0.0;
  FPO_PRAGMA(omp parallel for reduction(+ : s) schedule(static))
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
static double nrm2(int64_t n, const double *a) { return sqrt(dotp(n, a, a)); }
static void scal(int64_t n, double s, double *a) {
  FPO_PRAGMA(omp parallel for schedule(static))
  for (int64_t i = 0; i < n; ++i) a[i] *= s;
}
static void axpy(int64_t n, double s, const double *x, double *y) {
  FPO_PRAGMA(omp parallel for schedule(static))
  for (int64_t i = 0; i < n; ++i) y[i] += s * x[i];
}
/* y = a x + b y */
static void axpby(int64_t n, double a, const double *x, double b, double *y) {
  FPO_PRAGMA(omp parallel for schedule(static))
  for (int64_t i = 0; i < n; ++i) y[i] = a * x[i] + b * y[i];
}
static double sgn(double a) { return (a > 0.0) - (a < 0.0); }

/* Krylov.jl sym_givens (after Choi's SymOrtho): reflection [c s; s -c] [a; b] = [rho; 0] */
static void sym_givens(double a, double b, double *c, double *s, double *rho) {
  if (b == 0.0) {
    *c = (a == 0.0) ? 1.0 : sgn(a);
    *s = 0.0;
    *rho = fabs(a);
  } else if (a == 0.0) {
    *c = 0.0;
    *s = sgn(b);
    *rho = fabs(b);
  } else if (fabs(b) > fabs(a)) {
    const double t = a / b;
    *s = sgn(b) / sqrt(1.0 + t * t);
    *c = *s * t;
    *rho = b / *s;
  } else {
    const double t = b / a;
    *c = sgn(a) / sqrt(1.0 + t * t);
    *s = *c * t;
    *rho = a / *c;
  }
}

/* ------------------------------------------------------------------ LSQR
 * min ||b - Bx||^2 + lambda^2 ||x||^2, B = A or A'.  Krylov.jl lsqr! with M = N = I, radius = 0,
 * defaults axtol = btol = etol = sqrt(eps), conlim = 1/sqrt(eps), window = 5; the reference overrides
 * only lambda, atol, rtol, itmax (solve_two_systems_struct.jl:173-181). */
int fpo_lsqr_op(const fpo_op *B, const double *b, double lambda, double atol, double rtol, int64_t itmax,
                double axtol, double btol, double etol, double conlim, double *x, fpo_stats *st) {
  const int64_t m = op_rows(B), n = op_cols(B);
  const double ctol = conlim > 0 ? 1.0 / conlim : 0.0;
  const double lambda2 = lambda * lambda;
  enum { WINDOW = 5 };
  double err_vec[WINDOW] = {0, 0, 0, 0, 0};
  double *u = malloc((size_t)m * 8), *v = malloc((size_t)n * 8), *w = malloc((size_t)n * 8);
  double *Av = malloc((size_t)m * 8), *Atu = malloc((size_t)n * 8);
  memset(st, 0, sizeof *st);
  memset(x, 0, (size_t)n * 8);

  memcpy(u, b, (size_t)m * 8);
  const double beta1 = nrm2(m, u);
  if (beta1 == 0.0) {
    st->solved = 1; st->inconsistent = 0; st->niter = 0; st->status = FPO_ST_ZERO_RHS;
    goto done;
  }
  double beta = beta1;
  scal(m, 1.0 / beta1, u);
  op_tmul(B, u, Atu);
  memcpy(v, Atu, (size_t)n * 8);
  double Anorm2 = dotp(n, v, v);
  double Anorm = sqrt(Anorm2);
  double alpha = Anorm;
  double Acond = 0.0, xNorm = 0.0, xNorm2 = 0.0, dNorm2 = 0.0;
  double c2 = -1.0, s2 = 0.0, z = 0.0;
  double xENorm2 = 0.0, err_lbnd = 0.0;
  int64_t iter = 0;
  if (itmax == 0) itmax = m + n;
  double rNorm = beta1, res2 = 0.0;
  double ArNorm = alpha * beta;
  const double ArNorm0 = ArNorm;
  if (alpha == 0.0) {
    st->solved = 1; st->inconsistent = 0; st->niter = 0; st->status = FPO_ST_ZERO_ATB;
    st->rnorm = rNorm; st->arnorm = 0.0;
    goto done;
  }
  scal(n, 1.0 / alpha, v);
  memcpy(w, v, (size_t)n * 8);
  double phibar = beta1, rhobar = alpha;

  int solved_lim = ArNorm / (Anorm * rNorm) <= axtol;
  int solved_mach = 1.0 + ArNorm / (Anorm * rNorm) <= 1.0;
  int solved = solved_mach | solved_lim;
  int tired = iter >= itmax;
  int ill_cond = 0, ill_cond_mach = 0, ill_cond_lim = 0;
  int zero_resid_lim = rNorm / beta1 <= axtol;
  int zero_resid_mach = 1.0 + rNorm / beta1 <= 1.0;
  int zero_resid = zero_resid_mach | zero_resid_lim;
  int fwd_err = 0;

  while (!(solved || tired || ill_cond)) {
    iter++;
    /* 1. beta u = B v - alpha u */
    op_mul(B, v, Av);
    axpby(m, 1.0, Av, -alpha, u);
    beta = nrm2(m, u);
    if (beta != 0.0) {
      scal(m, 1.0 / beta, u);
      Anorm2 += alpha * alpha + beta * beta;
      if (lambda > 0) Anorm2 += lambda2;
      /* 2. alpha v = B' u - beta v */
      op_tmul(B, u, Atu);
      axpby(n, 1.0, Atu, -beta, v);
      alpha = nrm2(n, v);
      if (alpha != 0.0) scal(n, 1.0 / alpha, v);
    }
    /* eliminate the regularisation parameter */
    double c1, s1, rhobar1;
    sym_givens(rhobar, lambda, &c1, &s1, &rhobar1);
    const double psi = s1 * phibar;
    phibar = c1 * phibar;
    /* eliminate beta */
    double c, s, rho;
    sym_givens(rhobar1, beta, &c, &s, &rho);
    const double phi = c * phibar;
    phibar = s * phibar;

    xENorm2 += phi * phi;
    err_vec[iter % WINDOW] = phi;
    if (iter >= WINDOW) err_lbnd = nrm2(WINDOW, err_vec);

    const double tau = s * phi;
    const double theta = s * alpha;
    rhobar = -c * alpha;
    dNorm2 += dotp(n, w, w) / (rho * rho);

    axpy(n, phi / rho, w, x);         /* x = x + phi/rho w */
    axpby(n, 1.0, v, -theta / rho, w); /* w = v - theta/rho w */

    /* estimate ||x|| */
    const double delta = s2 * rho;
    const double gammabar = -c2 * rho;
    const double rhs = phi - delta * z;
    const double zbar = rhs / gammabar;
    xNorm = sqrt(xNorm2 + zbar * zbar);
    double gamma;
    sym_givens(gammabar, theta, &c2, &s2, &gamma);
    z = rhs / gamma;
    xNorm2 += z * z;

    Anorm = sqrt(Anorm2);
    Acond = Anorm * sqrt(dNorm2);
    const double res1 = phibar * phibar;
    res2 += psi * psi;
    rNorm = sqrt(res1 + res2);
    ArNorm = alpha * fabs(tau);

    const double test1 = rNorm / beta1;
    const double test2 = ArNorm / (Anorm * rNorm);
    const double test3 = 1.0 / Acond;
    const double t1 = test1 / (1.0 + Anorm * xNorm / beta1);
    const double rNormtol = btol + axtol * Anorm * xNorm / beta1;

    ill_cond_mach = (1.0 + test3 <= 1.0);
    solved_mach = (1.0 + test2 <= 1.0);
    zero_resid_mach = (1.0 + t1 <= 1.0);

    tired = iter >= itmax;
    ill_cond_lim = (test3 <= ctol);
    solved_lim = (test2 <= axtol);
    const int solved_opt = ArNorm <= atol + rtol * ArNorm0;
    zero_resid_lim = (test1 <= rNormtol);
    if (iter >= WINDOW) fwd_err = err_lbnd <= etol * sqrt(xENorm2);

    ill_cond = ill_cond_mach | ill_cond_lim;
    zero_resid = zero_resid_mach | zero_resid_lim;
    solved = solved_mach | solved_lim | solved_opt | zero_resid | fwd_err;
    if (g_trace) /* every stopping quantity over its threshold: < 1 fires */
      fprintf(stderr, "lsqr it %3d  solved_lim %.6e  solved_opt %.6e  zero_resid %.6e  fwd_err %.6e  ill_cond %.6e\n", (int)iter,
              test2 / axtol, ArNorm / (atol + rtol * ArNorm0), test1 / rNormtol,
              iter >= WINDOW ? err_lbnd / (etol * sqrt(xENorm2)) : INFINITY, ctol > 0 ? test3 / ctol : INFINITY);
  }
  st->status = FPO_ST_UNKNOWN;
  if (tired) st->status = FPO_ST_MAXITER;
  if (ill_cond) st->status = FPO_ST_ILL_COND;
  if (solved) st->status = FPO_ST_SOLVED;
  if (zero_resid) st->status = FPO_ST_ZERO_RESID;
  if (fwd_err) st->status = FPO_ST_FWD_ERR;
  st->niter = (int32_t)iter;
  st->solved = solved;
  st->inconsistent = !zero_resid;
  st->rnorm = rNorm;
  st->arnorm = ArNorm;
done:
  free(u); free(v); free(w); free(Av); free(Atu);
  return 0;
}

/* ------------------------------------------------------------------ CRAIG
 * min ||x|| s.t. Bx = b, or with sqd and M = (1/delta) I the system [-I B'; B delta I][x;y] = [0;b]
 * (solve_two_systems_struct.jl:216-239).  Krylov.jl craig! with N = I.  `mu` = 1/delta when the
 * reference passes M = 1/delta * opEye (delta != 0; then sqd = true => lambda = 1), else mu = 1, lambda = 0. */
int fpo_craig_op(const fpo_op *B, const double *b, double delta_reg, double atol, double rtol, double btol,
                 double conlim, int64_t itmax, double *x, double *y, fpo_stats *st) {
  const int64_t m = op_rows(B), n = op_cols(B);
  const int sqd = (delta_reg != 0.0);
  const double lambda = sqd ? 1.0 : 0.0;
  const double mu = sqd ? 1.0 / delta_reg : 1.0; /* M = mu I */
  double *Mu = malloc((size_t)m * 8), *u = malloc((size_t)m * 8), *Nv = malloc((size_t)n * 8);
  double *w = malloc((size_t)m * 8), *w2 = malloc((size_t)n * 8);
  double *Av = malloc((size_t)m * 8), *Atu = malloc((size_t)n * 8);
  memset(st, 0, sizeof *st);
  memset(x, 0, (size_t)n * 8);
  memset(y, 0, (size_t)m * 8);

  memcpy(Mu, b, (size_t)m * 8);
  for (int64_t i = 0; i < m; ++i) u[i] = mu * Mu[i]; /* u = M Mu */
  const double beta1 = sqrt(dotp(m, u, Mu));          /* elliptic norm */
  double rNorm = beta1;
  if (beta1 == 0.0) {
    st->solved = 1; st->inconsistent = 0; st->niter = 0; st->status = FPO_ST_ZERO_RHS;
    goto done;
  }
  const double beta1sq = beta1 * beta1;
  double beta = beta1, theta = beta1, xi = -1.0, delta = lambda, rho_prev = 1.0;
  scal(m, 1.0 / beta1, u);
  scal(m, 1.0 / beta1, Mu);
  memset(Nv, 0, (size_t)n * 8);
  memset(w, 0, (size_t)m * 8);
  memset(w2, 0, (size_t)n * 8);
  double Anorm2 = 0.0, Anorm = 0.0, Dnorm2 = 0.0, Acond = 0.0, xNorm2 = 0.0, xNorm = 0.0;
  int64_t iter = 0;
  if (itmax == 0) itmax = m + n;
  const double eps_c = atol + rtol * rNorm;
  const double ctol = conlim > 0 ? 1.0 / conlim : 0.0;
  double bkwerr = 1.0;
  int solved_lim = bkwerr <= btol;
  int solved_mach = 1.0 + bkwerr <= 1.0;
  int solved_resid_tol = rNorm <= eps_c;
  int solved_resid_lim = rNorm <= btol + atol * Anorm * xNorm / beta1;
  int solved = solved_mach | solved_lim | solved_resid_tol | solved_resid_lim;
  int ill_cond = 0, ill_cond_mach = 0, ill_cond_lim = 0, inconsistent = 0;
  int tired = iter >= itmax;
  double c1 = 1.0, s1 = 0.0, rho = 1.0;

  while (!(solved || inconsistent || ill_cond || tired)) {
    /* 1. alpha v = B' u - beta v */
    op_tmul(B, u, Atu);
    axpby(n, 1.0, Atu, -beta, Nv);
    const double alpha = nrm2(n, Nv);
    if (alpha == 0.0) { inconsistent = 1; continue; }
    scal(n, 1.0 / alpha, Nv);
    Anorm2 += alpha * alpha + lambda * lambda;
    if (lambda > 0) sym_givens(alpha, delta, &c1, &s1, &rho); else rho = alpha;
    xi = -theta / rho * xi;
    if (lambda > 0) {
      axpy(n, xi * c1, Nv, x);
      axpy(n, xi * s1, w2, x);
      axpby(n, s1, Nv, -c1, w2);
    } else {
      axpy(n, xi, Nv, x);
    }
    /* recur y */
    axpby(m, 1.0, u, -theta / rho_prev, w);
    axpy(m, xi / rho, w, y);
    Dnorm2 += nrm2(m, w); /* sic: Krylov.jl accumulates the norm, not its square */

    /* 2. beta Mu = B v - alpha Mu */
    op_mul(B, Nv, Av);
    axpby(m, 1.0, Av, -alpha, Mu);
    for (int64_t i = 0; i < m; ++i) u[i] = mu * Mu[i];
    beta = sqrt(dotp(m, u, Mu));
    if (beta != 0.0) {
      scal(m, 1.0 / beta, u);
      scal(m, 1.0 / beta, Mu);
    }
    double gamma = 0.0;
    if (lambda > 0) { theta = beta * c1; gamma = beta * s1; } else theta = beta;
    if (lambda > 0) {
      double c2, s2;
      sym_givens(lambda, gamma, &c2, &s2, &delta);
      scal(n, s2, w2);
    }
    Anorm2 += beta * beta;
    Anorm = sqrt(Anorm2);
    Acond = Anorm * sqrt(Dnorm2);
    xNorm2 += xi * xi;
    xNorm = sqrt(xNorm2);
    rNorm = beta * fabs(xi);
    if (lambda > 0) rNorm *= fabs(c1);
    iter++;
    bkwerr = rNorm / sqrt(beta1sq + Anorm2 * xNorm2);
    rho_prev = rho;

    solved_lim = bkwerr <= btol;
    solved_mach = 1.0 + bkwerr <= 1.0;
    solved_resid_tol = rNorm <= eps_c;
    solved_resid_lim = rNorm <= btol + atol * Anorm * xNorm / beta1;
    solved = solved_mach | solved_lim | solved_resid_tol | solved_resid_lim;
    ill_cond_mach = 1.0 + 1.0 / Acond <= 1.0;
    ill_cond_lim = 1.0 / Acond <= ctol;
    ill_cond = ill_cond_mach | ill_cond_lim;
    inconsistent = 0;
    tired = iter >= itmax;
  }
  st->status = FPO_ST_UNKNOWN;
  if (tired) st->status = FPO_ST_MAXITER;
  if (solved) st->status = FPO_ST_SOLVED;
  if (ill_cond) st->status = FPO_ST_ILL_COND;
  if (inconsistent) st->status = FPO_ST_INCONSISTENT;
  st->niter = (int32_t)iter;
  st->solved = solved;
  st->inconsistent = inconsistent;
  st->rnorm = rNorm;
done:
  free(Mu); free(u); free(Nv); free(w); free(w2); free(Av); free(Atu);
  return 0;
}

/* ------------------------------------------------------------------ LNLQ
 * Estrin, Orban & Saunders, "LNLQ: an iterative method for least-norm problems with an error minimization
 * property", SIMAX 40(3), 2019; Krylov.jl lnlq! with N = I, lambda = 0, sigma = 0 (no error bounds),
 * transfer_to_craig = true (the default).  The reference reaches it through the GENERIC solve_least_norm
 * (solve_two_systems_struct.jl:251-281), which passes M = (1/delta) I for delta != 0 but neither `sqd` nor a
 * regularisation: M then only preconditions -- the answer is the minimum-norm solution of Bx = b whatever delta is,
 * reached after a different number of passes because the residual estimates are measured in the M-norm while
 * eps = atol + rtol ||b||_2 is not.  `mu` = 1/delta (or 1).  iter counts as in lnlq!: it is advanced at the end of
 * every pass of the loop, including the last one, so niter = passes + 1. */
int fpo_lnlq_op(const fpo_op *B, const double *b, double delta_reg, double atol, double rtol, int64_t itmax,
                double *x, double *y, fpo_stats *st) {
  const int64_t m = op_rows(B), n = op_cols(B);
  const double mu = delta_reg != 0.0 ? 1.0 / delta_reg : 1.0; /* M = mu I */
  double *Mu = malloc((size_t)m * 8), *u = malloc((size_t)m * 8), *Nv = malloc((size_t)n * 8);
  double *wbar = malloc((size_t)m * 8), *Av = malloc((size_t)m * 8), *Atu = malloc((size_t)n * 8);
  memset(st, 0, sizeof *st);
  memset(x, 0, (size_t)n * 8);
  memset(y, 0, (size_t)m * 8);
  const double bNorm = nrm2(m, b);
  if (bNorm == 0.0) {
    st->solved = 1; st->niter = 0; st->status = FPO_ST_ZERO_RHS;
    goto done;
  }
  const double eps_l = atol + rtol * bNorm;
  int64_t iter = 0;
  if (itmax == 0) itmax = m + n;
  iter = iter + 1;
  /* beta_1 M u_1 = b */
  memcpy(Mu, b, (size_t)m * 8);
  for (int64_t i = 0; i < m; ++i) u[i] = mu * Mu[i];
  double beta = sqrt(dotp(m, u, Mu));
  if (beta != 0.0) { scal(m, 1.0 / beta, u); scal(m, 1.0 / beta, Mu); }
  /* alpha_1 N v_1 = B' u_1 */
  op_tmul(B, u, Atu);
  memcpy(Nv, Atu, (size_t)n * 8);
  double alpha = nrm2(n, Nv);
  if (alpha != 0.0) scal(n, 1.0 / alpha, Nv);
  memcpy(wbar, u, (size_t)m * 8);
  double ck = 0.0, sk = 0.0, zeta_km1 = 0.0, eta = 0.0;
  double ahat = alpha, epsbar = ahat;
  double tau = beta / ahat, zetabar = tau / epsbar;
  int solved_lq = 0, solved_cg = 0, tired = 0;
  double rNorm_lq = bNorm, rNorm_cg = bNorm;
  while (!(solved_lq || solved_cg || tired)) {
    /* (x aux)_k = V_k t_k */
    axpy(n, tau, Nv, x);
    /* beta_{k+1} M u_{k+1} = B v_k - alpha_k M u_k */
    op_mul(B, Nv, Av);
    axpby(m, 1.0, Av, -alpha, Mu);
    for (int64_t i = 0; i < m; ++i) u[i] = mu * Mu[i];
    const double beta_n = sqrt(dotp(m, u, Mu));
    if (beta_n != 0.0) { scal(m, 1.0 / beta_n, u); scal(m, 1.0 / beta_n, Mu); }
    /* alpha_{k+1} N v_{k+1} = B' u_{k+1} - beta_{k+1} N v_k */
    op_tmul(B, u, Atu);
    axpby(n, 1.0, Atu, -beta_n, Nv);
    const double alpha_n = nrm2(n, Nv);
    if (alpha_n != 0.0) scal(n, 1.0 / alpha_n, Nv);
    const double bhat_n = beta_n, ahat_n = alpha_n; /* lambda = 0 */
    /* continue the LQ factorisation of (L_{k+1})' */
    double c_n, s_n, eps_k;
    sym_givens(epsbar, bhat_n, &c_n, &s_n, &eps_k);
    const double eta_n = ahat_n * s_n;
    const double epsbar_n = -ahat_n * c_n;
    const double tau_n = -bhat_n * tau / ahat_n;
    const double zeta_k = c_n * zetabar;
    const double zetabar_n = (tau_n - eta_n * zeta_k) / epsbar_n;
    /* (y^L)_{k+1} = (y^L)_k + zeta_k w_k,  w_k = c wbar_k + s u_{k+1};  wbar_{k+1} = s wbar_k - c u_{k+1} */
    axpy(m, zeta_k * c_n, wbar, y);
    axpy(m, zeta_k * s_n, u, y);
    axpby(m, -c_n, u, s_n, wbar);
    if (iter == 1) rNorm_lq = bNorm;
    else rNorm_lq = fabs(ahat) * sqrt((epsbar * zetabar) * (epsbar * zetabar) + (bhat_n * sk * zeta_km1) * (bhat_n * sk * zeta_km1));
    rNorm_cg = fabs(bhat_n * tau);
    ck = c_n; sk = s_n; alpha = alpha_n; ahat = ahat_n; beta = beta_n; eta = eta_n; epsbar = epsbar_n; tau = tau_n;
    zeta_km1 = zeta_k; zetabar = zetabar_n;
    tired = iter >= itmax;
    solved_lq = rNorm_lq <= eps_l;
    solved_cg = rNorm_cg <= eps_l;
    iter = iter + 1;
  }
  (void)ck;
  if (solved_cg) { /* transfer to the CRAIG point */
    axpy(n, tau, Nv, x);
    axpy(m, zetabar, wbar, y);
  } else {
    axpy(n, eta * zeta_km1, Nv, x);
  }
  st->status = FPO_ST_UNKNOWN;
  if (tired) st->status = FPO_ST_MAXITER;
  if (solved_lq) st->status = FPO_ST_SOLVED_LQ;
  if (solved_cg) st->status = FPO_ST_SOLVED;
  st->niter = (int32_t)iter;
  st->solved = solved_lq || solved_cg;
  st->inconsistent = 0;
  st->rnorm = solved_cg ? rNorm_cg : rNorm_lq;
done:
  free(Mu); free(u); free(Nv); free(wbar); free(Av); free(Atu);
  return 0;
}

/* ------------------------------------------------------------------ MINRES (Krylov.jl minres! with M = I) on a
 * symmetric operator given as a callback y = Op v.  Two uses:
 *   (A A' + lambda I) y = b   -- the product operator `nlp.Aop * nlp.Aop'` of solve_linear_system.jl:58-70 (fpo_minres_aat)
 *   K [p; q] = b, K = [I A'; A -delta I] -- not a path of the reference; BASELINE.json north_star / configs[1] name it and
 *                                the library offers it as fpsq_options.kkt_method = FPSQ_KKT_MINRES_K (fpo_minres_kkt) */
typedef void (*fpo_symop)(const void *ctx, const double *v, double *y);

typedef struct {
  const fpo_csr *A;
  double lambda;
  double *tmp;
} aat_ctx;

static void op_aat(const void *c, const double *v, double *y) {
  const aat_ctx *k = (const aat_ctx *)c;
  csr_tmul(k->A, v, k->tmp);
  csr_mul(k->A, k->tmp, y);
  if (k->lambda != 0.0) axpy(k->A->m, k->lambda, v, y);
}

typedef struct {
  const fpo_csr *A;
  double delta;
} kkt_ctx;

static void op_kkt(const void *c, const double *v, double *y) { /* v = [p; q]: y = [p + A' q; A p - delta q] */
  const kkt_ctx *k = (const kkt_ctx *)c;
  const int64_t n = k->A->n, m = k->A->m;
  csr_tmul(k->A, v + n, y);
  axpy(n, 1.0, v, y);
  csr_mul(k->A, v, y + n);
  if (k->delta != 0.0) axpy(m, -k->delta, v + n, y + n);
}

static int minres_core(fpo_symop op, const void *ctx, int64_t n, const double *b, double atol, double rtol, double etol,
                       double conlim, int64_t itmax, double *x, fpo_stats *st) {
  const double epsM = 2.220446049250313e-16;
  const double ctol = conlim > 0 ? 1.0 / conlim : 0.0;
  enum { WINDOW = 5 };
  double err_vec[WINDOW] = {0, 0, 0, 0, 0};
  double *r1 = malloc((size_t)n * 8), *r2 = malloc((size_t)n * 8), *w1 = malloc((size_t)n * 8);
  double *w2 = malloc((size_t)n * 8), *yv = malloc((size_t)n * 8);
  memset(st, 0, sizeof *st);
  memset(x, 0, (size_t)n * 8);
  memcpy(r1, b, (size_t)n * 8);
  memcpy(r2, r1, (size_t)n * 8);
  double *v = r2; /* M = I */
  double beta1 = dotp(n, r1, v);
  if (beta1 == 0.0) {
    st->solved = 1; st->inconsistent = 0; st->niter = 0; st->status = FPO_ST_ZERO_RHS;
    goto done;
  }
  beta1 = sqrt(beta1);
  double beta = beta1, oldbeta = 0.0, deltabar = 0.0, epsln = 0.0, rNorm = beta1, phibar = beta1;
  double rhs1 = beta1, rhs2 = 0.0, gmax = 0.0, gmin = INFINITY, cs = -1.0, sn = 0.0;
  memset(w1, 0, (size_t)n * 8);
  memset(w2, 0, (size_t)n * 8);
  double ANorm2 = 0.0, ANorm = 0.0, Acond = 0.0, ArNorm = 0.0, xNorm = 0.0, xENorm2 = 0.0, err_lbnd = 0.0;
  int64_t iter = 0;
  if (itmax == 0) itmax = 2 * n;
  const double eps_tol = atol + rtol * beta1;
  int solved = (rNorm <= rtol), tired = iter >= itmax, ill_cond = 0, fwd_err = 0;
  int zero_resid = (rNorm <= eps_tol);
  int early = 0;

  while (!(solved || tired || ill_cond)) {
    iter++;
    /* y = Op v / beta */
    op(ctx, v, yv);
    scal(n, 1.0 / beta, yv);
    if (iter >= 2) axpy(n, -beta / oldbeta, r1, yv);
    const double alpha = dotp(n, v, yv) / beta;
    axpy(n, -alpha / beta, r2, yv);

    const double delta = cs * deltabar + sn * alpha;
    double *w;
    if (iter == 1) {
      w = w2;
    } else {
      if (iter >= 3) scal(n, -epsln, w1);
      w = w1;
      axpy(n, -delta, w2, w);
    }
    axpy(n, 1.0 / beta, v, w);

    memcpy(r1, r2, (size_t)n * 8);
    memcpy(r2, yv, (size_t)n * 8);
    oldbeta = beta;
    beta = dotp(n, r2, v);
    beta = sqrt(beta);
    ANorm2 += alpha * alpha + oldbeta * oldbeta + beta * beta;

    const double gammabar = sn * deltabar - cs * alpha;
    epsln = sn * beta;
    deltabar = -cs * beta;
    const double root = sqrt(gammabar * gammabar + deltabar * deltabar);
    ArNorm = phibar * root;

    double gamma = sqrt(gammabar * gammabar + beta * beta);
    gamma = fmax(gamma, epsM);
    cs = gammabar / gamma;
    sn = beta / gamma;
    const double phi = cs * phibar;
    phibar = sn * phibar;

    scal(n, 1.0 / gamma, w);
    axpy(n, phi, w, x);
    xENorm2 += phi * phi;
    if (iter >= 2) { double *t = w1; w1 = w2; w2 = t; }

    err_vec[iter % WINDOW] = phi;
    if (iter >= WINDOW) err_lbnd = nrm2(WINDOW, err_vec);

    gmax = fmax(gmax, gamma);
    gmin = fmin(gmin, gamma);
    const double zeta = rhs1 / gamma;
    rhs1 = rhs2 - delta * zeta;
    rhs2 = -epsln * zeta;

    ANorm = sqrt(ANorm2);
    xNorm = nrm2(n, x);
    rNorm = phibar;
    const double test1 = rNorm / (ANorm * xNorm);
    const double test2 = root / ANorm;
    Acond = gmax / gmin;

    if (iter == 1 && beta / beta1 <= 10 * epsM) {
      st->niter = 1; st->solved = 1; st->inconsistent = 1; st->status = FPO_ST_ZERO_ATB;
      st->rnorm = rNorm; st->arnorm = ArNorm;
      early = 1;
      break;
    }
    const int ill_cond_mach = (1.0 + 1.0 / Acond <= 1.0);
    const int solved_mach = (1.0 + test2 <= 1.0);
    const int zero_resid_mach = (1.0 + test1 <= 1.0);
    const int resid_decrease_mach = (rNorm + 1.0 <= 1.0);
    tired = iter >= itmax;
    const int ill_cond_lim = (1.0 / Acond <= ctol);
    const int solved_lim = (test2 <= eps_tol);
    const int zero_resid_lim = (test1 <= epsM);
    const int resid_decrease_lim = (rNorm <= eps_tol);
    if (iter >= WINDOW) fwd_err = err_lbnd <= etol * sqrt(xENorm2);
    zero_resid = zero_resid_mach | zero_resid_lim;
    const int resid_decrease = resid_decrease_mach | resid_decrease_lim;
    ill_cond = ill_cond_mach | ill_cond_lim;
    solved = solved_mach | solved_lim | zero_resid | fwd_err | resid_decrease;
  }
  if (!early) {
    st->status = FPO_ST_UNKNOWN;
    if (tired) st->status = FPO_ST_MAXITER;
    if (ill_cond) st->status = FPO_ST_ILL_COND;
    if (solved) st->status = FPO_ST_SOLVED;
    if (zero_resid) st->status = FPO_ST_ZERO_RESID;
    if (fwd_err) st->status = FPO_ST_FWD_ERR;
    st->niter = (int32_t)iter;
    st->solved = solved;
    st->inconsistent = !zero_resid;
    st->rnorm = rNorm;
    st->arnorm = ArNorm;
  }
done:
  free(r1); free(r2); free(w1); free(w2); free(yv);
  return 0;
}

int fpo_minres_aat(const fpo_csr *A, const double *b, double lambda, double atol, double rtol, double etol,
                   double conlim, int64_t itmax, double *x, fpo_stats *st) {
  aat_ctx c = {A, lambda, malloc((size_t)A->n * 8)};
  const int rc = minres_core(op_aat, &c, A->m, b, atol, rtol, etol, conlim, itmax, x, st);
  free(c.tmp);
  return rc;
}

/* K [p; q] = [bp; bq] (null = zero); x = [p (n); q (m)] */
int fpo_minres_kkt(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals, double delta,
                   const double *bp, const double *bq, double atol, double rtol, double etol, double conlim,
                   int64_t itmax, double *x, fpo_stats *st) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  kkt_ctx c = {&A, delta};
  double *b = calloc((size_t)(n + m), 8);
  if (bp) memcpy(b, bp, (size_t)n * 8);
  if (bq) memcpy(b + n, bq, (size_t)m * 8);
  const int rc = minres_core(op_kkt, &c, n + m, b, atol, rtol, etol, conlim, itmax, x, st);
  free(b);
  return rc;
}

/* ------------------------------------------------------------------ flat entry points for ctypes */

int fpo_lsqr(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
             int transposed, const double *b, double lambda, double atol, double rtol, int64_t itmax,
             double axtol, double btol, double etol, double conlim, double *x, fpo_stats *st) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  fpo_op B = {&A, transposed};
  return fpo_lsqr_op(&B, b, lambda, atol, rtol, itmax, axtol, btol, etol, conlim, x, st);
}

int fpo_craig(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
              int transposed, const double *b, double delta, double atol, double rtol, double btol,
              double conlim, int64_t itmax, double *x, double *y, fpo_stats *st) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  fpo_op B = {&A, transposed};
  return fpo_craig_op(&B, b, delta, atol, rtol, btol, conlim, itmax, x, y, st);
}

int fpo_lnlq(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals, int transposed,
             const double *b, double delta, double atol, double rtol, int64_t itmax, double *x, double *y,
             fpo_stats *st) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  fpo_op B = {&A, transposed};
  return fpo_lnlq_op(&B, b, delta, atol, rtol, itmax, x, y, st);
}

int fpo_minres(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
               const double *b, double lambda, double atol, double rtol, double etol, double conlim,
               int64_t itmax, double *x, fpo_stats *st) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  return fpo_minres_aat(&A, b, lambda, atol, rtol, etol, conlim, itmax, x, st);
}

void fpo_spmv(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
              int transposed, const double *x, double *y) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  if (transposed) csr_tmul(&A, x, y); else csr_mul(&A, x, y);
}

/* solve_two_mixed, iterative back-end: solve_linear_system.jl:107-140.
 * rhs1 (n), rhs2 (m) -> p1 (n), q1 (m), p2 (n), q2 (m). */
int fpo_solve_two_mixed(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
                        double delta, const fpo_options *o, const double *rhs1, const double *rhs2,
                        double *p1, double *q1, double *p2, double *q2, fpo_stats st[2]) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  fpo_op At = {&A, 1}, Aop = {&A, 0};
  /* (q1, stats1) = solve_least_square(qds, Aop', rhs1, sqrt(delta))            :123 */
  fpo_lsqr_op(&At, rhs1, sqrt(delta), o->ls_atol, o->ls_rtol, o->ls_itmax, o->ls_axtol, o->ls_btol, o->ls_etol,
              o->ls_conlim, q1, &st[0]);
  /* p1 = rhs1 - Aop' * q1                                                       :126-127 */
  csr_tmul(&A, q1, p1);
  for (int64_t i = 0; i < n; ++i) p1[i] = rhs1[i] - p1[i];
  /* (p2, q2, stats2) = solve_least_norm(qds, Aop, -rhs2, delta); p2 = -p2       :132-133 */
  double *nrhs2 = malloc((size_t)m * 8);
  for (int64_t i = 0; i < m; ++i) nrhs2[i] = -rhs2[i];
  if (o->ln_method == 1) /* the generic solve_least_norm with an LNLQ workspace, struct.jl:251-281 */
    fpo_lnlq_op(&Aop, nrhs2, delta, o->ln_atol, o->ln_rtol, o->ln_itmax, p2, q2, &st[1]);
  else
    fpo_craig_op(&Aop, nrhs2, delta, o->ln_atol, o->ln_rtol, o->ln_btol, o->ln_conlim, o->ln_itmax, p2, q2, &st[1]);
  for (int64_t i = 0; i < n; ++i) p2[i] = -p2[i];
  free(nrhs2);
  return (st[0].solved ? 0 : 1) | (st[1].solved ? 0 : 2);
}

/* solve_two_least_squares, iterative back-end: solve_linear_system.jl:79-105.  rhs1, rhs2 both size n.
 * The reference returns the SAME array for q1 and q2 (the LSQR workspace's x); here they are separate. */
int fpo_solve_two_least_squares(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind,
                                const double *vals, double delta, const fpo_options *o, const double *rhs1,
                                const double *rhs2, double *p1, double *q1, double *p2, double *q2,
                                fpo_stats st[2]) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  fpo_op At = {&A, 1};
  fpo_lsqr_op(&At, rhs1, sqrt(delta), o->ls_atol, o->ls_rtol, o->ls_itmax, o->ls_axtol, o->ls_btol, o->ls_etol,
              o->ls_conlim, q1, &st[0]);
  csr_tmul(&A, q1, p1);
  for (int64_t i = 0; i < n; ++i) p1[i] = rhs1[i] - p1[i];
  fpo_lsqr_op(&At, rhs2, sqrt(delta), o->ls_atol, o->ls_rtol, o->ls_itmax, o->ls_axtol, o->ls_btol, o->ls_etol,
              o->ls_conlim, q2, &st[1]);
  csr_tmul(&A, q2, p2);
  for (int64_t i = 0; i < n; ++i) p2[i] = rhs2[i] - p2[i];
  return (st[0].solved ? 0 : 1) | (st[1].solved ? 0 : 2);
}

/* solve_two_extras, iterative back-end: solve_linear_system.jl:45-77.  rhs1 (n), rhs2 (m) -> two m-vectors. */
int fpo_solve_two_extras(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
                         double delta, const fpo_options *o, const double *rhs1, const double *rhs2,
                         double *invJtJJv, double *invJtJSsv, fpo_stats st[2]) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  fpo_op At = {&A, 1};
  const double tau = fmax(delta, 1e-14); /* :51 */
  fpo_lsqr_op(&At, rhs1, sqrt(tau), o->ls_atol, o->ls_rtol, o->ls_itmax, o->ls_axtol, o->ls_btol, o->ls_etol,
              o->ls_conlim, invJtJJv, &st[0]);
  fpo_minres_aat(&A, rhs2, tau, o->ne_atol, o->ne_rtol, o->ne_etol, o->ne_conlim, o->ne_itmax, invJtJSsv, &st[1]);
  return (st[0].solved ? 0 : 1) | (st[1].solved ? 0 : 2);
}

/* One penalty objgrad! on the synthetic equality-QP user model
 *     f(x) = 1/2 x' diag(q) x + d'x,   c(x) = A x - b   (lcon = ucon = 0 after the shift)
 * following model-Fletcherpenaltynlp.jl:234-252 (_compute_ys_gs!) and :403-437 (objgrad!):
 *   g = qx + d; c = Ax - b; (p1,q1,p2,q2) = solve_two_mixed(g, c);
 *   gs = p1 + sigma p2; ys = q1 + sigma q2; v = p2; w = q2;
 *   Hsv = hprod(x, ys, v; obj_weight=1) = q.*v  (constraints are linear);  Sstw = hprod(x, w, gs; 0) = 0
 *   gx = gs - Hsv + sigma v + Sstw (+ rho A'c) (+ eta (x - xk));
 *   fx = f - c'ys (+ rho/2 c'c) (+ eta/2 ||x-xk||^2). */
int fpo_qp_objgrad(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
                   const double *qdiag, const double *d, const double *b, const double *x, double sigma,
                   double rho, double delta, double eta, const double *xk, const fpo_options *o, double *gx,
                   double *fx_out, double *ys, double *gs, fpo_stats st[2]) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  double *g = malloc((size_t)n * 8), *c = calloc((size_t)(m > 0 ? m : 1), 8);
  double *p1 = malloc((size_t)n * 8), *q1 = malloc((size_t)m * 8);
  double *p2 = malloc((size_t)n * 8), *q2 = malloc((size_t)m * 8);
  double *Jc = malloc((size_t)n * 8);
  double f = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    g[i] = qdiag[i] * x[i] + d[i];
    f += x[i] * (0.5 * qdiag[i] * x[i] + d[i]);
  }
  csr_mul(&A, x, c);
  for (int64_t i = 0; i < m; ++i) c[i] -= b[i];
  const int rc = fpo_solve_two_mixed(m, n, rowptr, colind, vals, delta, o, g, c, p1, q1, p2, q2, st);
  for (int64_t i = 0; i < n; ++i) gs[i] = p1[i] + sigma * p2[i];
  for (int64_t i = 0; i < m; ++i) ys[i] = q1[i] + sigma * q2[i];
  /* v = p2, w = q2 */
  for (int64_t i = 0; i < n; ++i) {
    const double Hsv = qdiag[i] * p2[i];
    gx[i] = gs[i] - Hsv + sigma * p2[i] + 0.0;
  }
  double fx = f - dotp(m, c, ys);
  if (rho > 0.0) {
    csr_tmul(&A, c, Jc);
    for (int64_t i = 0; i < n; ++i) gx[i] += Jc[i] * rho;
    fx += rho / 2 * dotp(m, c, c);
  }
  if (eta > 0.0) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      const double dx = x[i] - xk[i];
      s += dx * dx;
      gx[i] += eta * dx;
    }
    fx += eta / 2 * s;
  }
  *fx_out = fx;
  free(g); free(c); free(p1); free(q1); free(p2); free(q2); free(Jc);
  return rc;
}

/* One penalty hprod! on the same equality-QP user model, following model-Fletcherpenaltynlp.jl:521-570 (Val(2),
 * approx = 2) and :572-634 (Val(1), approx = 1).  The constraints are linear, so
 *   hprod_nln!(x, -ys, v; obj_weight = 1) = q .* v,   hprod_nln!(x, y, v; obj_weight = 0) = 0,   ghjvprod = 0:
 *   Hsv = q.*v                                              :537-538 / :588-589
 *   (p1, _, p2, _) = solve_two_least_squares(v, Hsv)        :542 / :591
 *   Ptv = v - p1;  HsPtv = q.*Ptv                           :543-545 / :592-596
 *   Val(1) only: Ssv = 0; (invJtJJv, invJtJSsv) = solve_two_extras(v, Ssv); JtinvJtJSsv = A'invJtJSsv;
 *                SsinvJtJJv = 0                             :598-612   (their stats land in st[2], st[3])
 *   Hv = p2 - HsPtv + 2 sigma Ptv (- JtinvJtJSsv - SsinvJtJJv)          :550 / :609-612
 *   rho > 0: Hv += Hcv (= 0) + rho A'(A v)                  :552-563 / :614-625
 *   eta > 0: Hv += eta v                                    :564-566 / :626-628
 * The product does not depend on x for this model (ys only enters through the vanishing constraint Hessians). */
int fpo_qp_hprod(int64_t m, int64_t n, const int64_t *rowptr, const int64_t *colind, const double *vals,
                 const double *qdiag, const double *v, double sigma, double rho, double delta, double eta,
                 int approx, const fpo_options *o, double *Hv, fpo_stats st[4]) {
  fpo_csr A = {m, n, rowptr, colind, vals};
  double *Hsv = calloc((size_t)(n > 0 ? n : 1), 8), *p1 = malloc((size_t)n * 8), *p2 = malloc((size_t)n * 8);
  double *q1 = malloc((size_t)m * 8), *q2 = malloc((size_t)m * 8);
  double *Jv = calloc((size_t)(m > 0 ? m : 1), 8), *JtJv = malloc((size_t)n * 8);
  memset(st, 0, 4 * sizeof(fpo_stats));
  for (int64_t i = 0; i < n; ++i) Hsv[i] = qdiag[i] * v[i];
  int rc = fpo_solve_two_least_squares(m, n, rowptr, colind, vals, delta, o, v, Hsv, p1, q1, p2, q2, st);
  for (int64_t i = 0; i < n; ++i) {
    const double Ptv = v[i] - p1[i];
    Hv[i] = p2[i] - qdiag[i] * Ptv + 2.0 * sigma * Ptv;
  }
  if (approx == 1) {
    double *Ssv = calloc((size_t)(m > 0 ? m : 1), 8);
    const int rc2 = fpo_solve_two_extras(m, n, rowptr, colind, vals, delta, o, v, Ssv, q1, q2, st + 2);
    csr_tmul(&A, q2, JtJv); /* JtinvJtJSsv */
    for (int64_t i = 0; i < n; ++i) Hv[i] -= JtJv[i];
    rc |= rc2 << 2;
    free(Ssv);
  }
  if (rho > 0.0) {
    csr_mul(&A, v, Jv);
    csr_tmul(&A, Jv, JtJv);
    for (int64_t i = 0; i < n; ++i) Hv[i] += rho * JtJv[i];
  }
  if (eta > 0.0)
    for (int64_t i = 0; i < n; ++i) Hv[i] += eta * v[i];
  free(Hsv); free(p1); free(p2); free(q1); free(q2); free(Jv); free(JtJv);
  return rc;
}
