/*
 * CPU oracle for the batched differential-IK hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain C restatement of what the reference (stephane-caron/pink) computes
 * between "per-task (J, e) and stacked (G, h) arrays exist" and "dq exists":
 *
 *   stacking   pink/tasks/task.py:145-167   (per-task H_t, c_t)
 *              pink/solve_ik.py:54-67        (damping*I + sum over tasks/barriers)
 *   QP solve   pink/solve_ik.py:270          qpsolvers.solve_problem(solver="quadprog")
 *
 * The QP arithmetic of the reference lives in the third-party package quadprog
 * (not vendored, not pinned in pixi.lock/uv.lock, not installable offline).  It
 * implements the dual active-set method of Goldfarb & Idnani (Math. Prog. 27,
 * 1983); this file restates that published algorithm in its updating form
 * (J = L^-T Q, R triangular, Givens add/drop).  PARITY UNPINNED by a run of the
 * reference for the solve half: it ships no golden dq; see oracle/pink_oracle.py
 * header for how it is anchored (the known answers quadprog / qpsolvers publish,
 * tests/test_published_qp.py; KKT certificate; BVLS and SLSQP cross-checks; a
 * second independent implementation in NumPy).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pink_amd / libpinkhip.so) never does.
 *
 * Build: make -C oracle   ->  oracle/liboracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_OPTIMAL 0
#define ORACLE_MAX_ITER 1
#define ORACLE_INFEASIBLE 2
#define ORACLE_NOT_PD 3

/* Number of doubles of scratch oracle_gi_solve needs for an (n, m) problem. */
long oracle_gi_work_size(int n, int m) {
  return 3L * n * n + 8L * n + 4L * m + 16;
}

/*
 * min 1/2 x'Px + q'x   s.t.  G x <= h      (P: n x n row-major, G: m x n row-major)
 *
 * The first `meq` rows are equalities G_i x = h_i (pink/solve_ik.py:140-149 builds them from
 * `constraints=`; qpsolvers passes them to quadprog as its `meq` leading constraints): they are
 * activated first, with the normal oriented so that the residual reads as a violation, and are
 * never dropped.
 *
 * Returns a status code; x[n], lam[m] (may be NULL), *iters (may be NULL).
 * Goldfarb-Idnani, SURVEY.md Appendix B.2.  Constraints are handled in the
 * form n_i' x >= b_i with n_i = -G_i, b_i = -h_i (qpsolvers -> quadprog
 * mapping, Appendix B.1).
 */
int oracle_gi_solve_eq(int n, const double *P, const double *q, int m, int meq, const double *G,
                       const double *h, double *x, double *lam, int *iters, int max_iter,
                       double *work) {
  double *L = work;            /* n*n lower Cholesky factor            */
  double *J = L + (long)n * n; /* n*n, J = L^-T Q                      */
  double *R = J + (long)n * n; /* n*n upper triangular, leading qa*qa  */
  double *d = R + (long)n * n; /* n                                    */
  double *z = d + n;           /* n                                    */
  double *r = z + n;           /* n                                    */
  double *u = r + n;           /* n   multipliers of active rows       */
  double *y = u + n;           /* n                                    */
  double *np_ = y + n;         /* n   current normal                   */
  double *nrm = np_ + n;       /* m   row norms                        */
  double *uall = nrm + m;      /* m                                    */
  int *A = (int *)(uall + m);  /* n   active row indices (as ints)     */
  int *is_active = A + n + 2;  /* m                                    */
  int it = 0, qa = 0, eq_next = 0;
  /* violation threshold relative to 1 + |b_i| / |n_i|: round-off of the iterate grows with the dimension
   * (n dot products of length n per step), so does the threshold: 1e-13 up to n = 8, 1e-13 n / 8 beyond */
  const double tol = 1e-13 * (n > 8 ? n / 8.0 : 1.0);

  if (iters) *iters = 0;
  if (max_iter <= 0) max_iter = 20 * (n + m) + 50;

  /* Cholesky P = L L' */
  for (int j = 0; j < n; ++j) {
    double s = P[(long)j * n + j];
    for (int k = 0; k < j; ++k) s -= L[(long)j * n + k] * L[(long)j * n + k];
    if (!(s > 0.0)) return ORACLE_NOT_PD;
    double ljj = sqrt(s);
    L[(long)j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double t = P[(long)i * n + j];
      for (int k = 0; k < j; ++k) t -= L[(long)i * n + k] * L[(long)j * n + k];
      L[(long)i * n + j] = t / ljj;
    }
    for (int k = j + 1; k < n; ++k) L[(long)j * n + k] = 0.0;
  }
  /* J = L^-T: column c of L^-1 by forward substitution, stored as row c of J */
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < n; ++i) {
      double t = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) t -= L[(long)i * n + k] * J[(long)c * n + k];
      J[(long)c * n + i] = (i < c) ? 0.0 : t / L[(long)i * n + i];
    }
  }
  /* unconstrained minimum x = -P^-1 q = J (J' (-q)) */
  for (int j = 0; j < n; ++j) {
    double t = 0.0;
    for (int i = 0; i < n; ++i) t -= J[(long)i * n + j] * q[i];
    y[j] = t;
  }
  for (int i = 0; i < n; ++i) {
    double t = 0.0;
    for (int j = 0; j < n; ++j) t += J[(long)i * n + j] * y[j];
    x[i] = t;
  }
  if (lam)
    for (int i = 0; i < m; ++i) lam[i] = 0.0;
  if (m == 0) return ORACLE_OPTIMAL;

  for (int i = 0; i < m; ++i) {
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += G[(long)i * n + k] * G[(long)i * n + k];
    nrm[i] = (s > 0.0) ? sqrt(s) : 1.0;
    is_active[i] = 0;
  }

  for (;;) {
    /* Step 1: pending equality first, else the most violated inequality (violation / row
       norm, as quadprog) */
    int p = -1;
    double worst = 0.0, sp = 0.0, sgn = 1.0;
    if (eq_next < meq) {
      p = eq_next;
      double s = h[p];
      for (int k = 0; k < n; ++k) s -= G[(long)p * n + k] * x[k];
      if (s > 0.0) {  /* orient the normal so that the residual is a violation */
        sgn = -1.0;
        s = -s;
      }
      sp = s;
    } else {
      for (int i = meq; i < m; ++i) {
        if (is_active[i]) continue;
        double s = h[i]; /* slack s = h_i - G_i x; violated when s < 0 */
        for (int k = 0; k < n; ++k) s -= G[(long)i * n + k] * x[k];
        double sc = s / nrm[i];
        double thr = -tol * (1.0 + fabs(h[i]) / nrm[i]);
        if (sc < thr && (p < 0 || sc < worst)) {
          p = i;
          worst = sc;
          sp = s;
        }
      }
    }
    if (p < 0) break; /* optimal */

    for (int k = 0; k < n; ++k) np_[k] = -sgn * G[(long)p * n + k];
    double uplus = 0.0;

    for (;;) {
      if (++it > max_iter) {
        if (iters) *iters = it;
        return ORACLE_MAX_ITER;
      }
      /* Step 2a: d = J' n+, z = J2 d2, r = R^-1 d1 */
      double dd = 0.0, d2 = 0.0;
      for (int j = 0; j < n; ++j) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += J[(long)i * n + j] * np_[i];
        d[j] = t;
        dd += t * t;
        if (j >= qa) d2 += t * t;
      }
      for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = qa; j < n; ++j) t += J[(long)i * n + j] * d[j];
        z[i] = t;
      }
      for (int k = qa - 1; k >= 0; --k) {
        double t = d[k];
        for (int l = k + 1; l < qa; ++l) t -= R[(long)k * n + l] * r[l];
        r[k] = t / R[(long)k * n + k];
      }
      /* Step 2b: step lengths */
      double t1 = INFINITY, t2 = INFINITY;
      int drop = -1;
      for (int k = 0; k < qa; ++k) {
        if (A[k] < meq) continue; /* equalities are never dropped */
        if (r[k] > 0.0) {
          double cand = u[k] / r[k];
          if (cand < t1) {
            t1 = cand;
            drop = k;
          }
        }
      }
      if (d2 > 1e-24 * dd) {
        double zn = 0.0;
        for (int k = 0; k < n; ++k) zn += z[k] * np_[k];
        /* n+'x - b_p = h_p - G_p x = sp < 0.  The full step makes the row
           active: n+'(x + t z) = b_p  =>  t = -sp / (z'n+). */
        t2 = -sp / zn;
      }
      if (p < meq && !(t2 < INFINITY) && !(t1 < INFINITY) && fabs(sp) <= 1e-9 * (1.0 + fabs(h[p]))) {
        /* equality implied by the active ones and already satisfied: nothing to add */
        ++eq_next;
        break;
      }
      double t = (t1 < t2) ? t1 : t2;
      if (!(t < INFINITY)) {
        if (iters) *iters = it;
        return ORACLE_INFEASIBLE;
      }
      if (!(t2 < INFINITY)) {
        /* dual step only, then drop the blocking constraint */
        for (int k = 0; k < qa; ++k) u[k] -= t * r[k];
        uplus += t;
      } else {
        for (int k = 0; k < n; ++k) x[k] += t * z[k];
        for (int k = 0; k < qa; ++k) u[k] -= t * r[k];
        uplus += t;
        if (t2 <= t1) {
          /* full step: add constraint p.  Givens rotations fold d[qa+1..n-1]
             into d[qa], applied to the columns of J. */
          for (int j = n - 1; j > qa; --j) {
            if (d[j] == 0.0) continue;
            double hy = hypot(d[j - 1], d[j]);
            double c = d[j - 1] / hy, s = d[j] / hy;
            d[j - 1] = hy;
            d[j] = 0.0;
            for (int i = 0; i < n; ++i) {
              double a = J[(long)i * n + j - 1], b = J[(long)i * n + j];
              J[(long)i * n + j - 1] = c * a + s * b;
              J[(long)i * n + j] = -s * a + c * b;
            }
          }
          for (int k = 0; k <= qa; ++k) R[(long)k * n + qa] = d[k];
          A[qa] = p;
          u[qa] = (p < meq) ? -sgn * uplus : uplus; /* multiplier of G_p x = h_p, either sign */
          is_active[p] = 1;
          if (p < meq) ++eq_next;
          ++qa;
          break; /* back to step 1 */
        }
      }
      /* drop active constraint at position `drop` */
      is_active[A[drop]] = 0;
      for (int col = drop; col < qa - 1; ++col) {
        for (int k = 0; k <= col + 1; ++k) R[(long)k * n + col] = R[(long)k * n + col + 1];
        A[col] = A[col + 1];
        u[col] = u[col + 1];
      }
      --qa;
      for (int k = drop; k < qa; ++k) {
        double a = R[(long)k * n + k], b = R[(long)(k + 1) * n + k];
        if (b == 0.0) continue;
        double hy = hypot(a, b);
        double c = a / hy, s = b / hy;
        for (int col = k; col < qa; ++col) {
          double ra = R[(long)k * n + col], rb = R[(long)(k + 1) * n + col];
          R[(long)k * n + col] = c * ra + s * rb;
          R[(long)(k + 1) * n + col] = -s * ra + c * rb;
        }
        for (int i = 0; i < n; ++i) {
          double ja = J[(long)i * n + k], jb = J[(long)i * n + k + 1];
          J[(long)i * n + k] = c * ja + s * jb;
          J[(long)i * n + k + 1] = -s * ja + c * jb;
        }
      }
      /* recompute the slack of p at the (possibly moved) x and iterate step 2 */
      {
        double s = h[p];
        for (int k = 0; k < n; ++k) s -= G[(long)p * n + k] * x[k];
        sp = sgn * s;
      }
    }
  }
  if (iters) *iters = it;
  if (lam)
    for (int k = 0; k < qa; ++k) lam[A[k]] = u[k];
  return ORACLE_OPTIMAL;
}

int oracle_gi_solve(int n, const double *P, const double *q, int m, const double *G,
                    const double *h, double *x, double *lam, int *iters, int max_iter,
                    double *work) {
  return oracle_gi_solve_eq(n, P, q, m, 0, G, h, x, lam, iters, max_iter, work);
}

/*
 * One task's contribution, pink/tasks/task.py:145-167, accumulated into H, c.
 *   We = w .* (-gain e);  WJ = diag(w) J;  mu = lm * We.We
 *   H += WJ' WJ + mu I ;  c += -We' WJ
 * `w` has k entries (the caller expands None / scalar costs, task.py:148-156).
 */
void oracle_task_objective(int k, int nv, const double *J, const double *e, const double *w,
                           double gain, double lm, double *H, double *c, double *wj /* k*nv */) {
  double mu = 0.0;
  for (int a = 0; a < k; ++a) {
    double we = w[a] * (-gain * e[a]);
    mu += we * we;
    for (int j = 0; j < nv; ++j) wj[(long)a * nv + j] = w[a] * J[(long)a * nv + j];
  }
  mu *= lm;
  for (int i = 0; i < nv; ++i) {
    for (int j = 0; j < nv; ++j) {
      double s = 0.0;
      for (int a = 0; a < k; ++a) s += wj[(long)a * nv + i] * wj[(long)a * nv + j];
      H[(long)i * nv + j] += s;
    }
    H[(long)i * nv + i] += mu;
  }
  for (int j = 0; j < nv; ++j) {
    double s = 0.0;
    for (int a = 0; a < k; ++a) s += (w[a] * (-gain * e[a])) * wj[(long)a * nv + j];
    c[j] += -s;
  }
}

/*
 * Whole path for a batch given in the form Pink builds it: every task dense
 * (identity Jacobians included), every limit/barrier as rows of (G, h).
 *
 *   J     [B, K, nv]   stacked task Jacobians, K = rows[T]
 *   e     [B, K]
 *   cost  [K] or [B, K] (cost_batched) -- expanded weights, one per row
 *   gain, lm [T];  rows [T+1] row offsets
 *   diag_extra [B] or NULL -- sum over barriers of r_b / ||J_h,b||_F^2 (barrier.py:193-200)
 *   c_extra [B, nv] or NULL -- sum over barriers of -rho_b dq_safe,b (barrier.py:201)
 *   G [B, m, nv], h [B, m]   (solve_ik.py:107-122); the first meq rows are equalities
 *                            A dq = b (solve_ik.py:140-149)
 * Outputs dq [B, nv], status [B], iters [B]; optional H_out [B, nv, nv], c_out [B, nv].
 */
int oracle_solve_ik_batch(long B, int nv, int T, const int *rows, const double *J, const double *e,
                          const double *cost, int cost_batched, const double *gain,
                          const double *lm, double damping, const double *diag_extra,
                          const double *c_extra, int m, int meq, const double *G, const double *h,
                          double *dq, int *status, int *iters, double *H_out, double *c_out,
                          int solve, int nthreads) {
  const int K = rows[T];
  int kmax = 0;
  for (int t = 0; t < T; ++t)
    if (rows[t + 1] - rows[t] > kmax) kmax = rows[t + 1] - rows[t];
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
  int failed = 0;
#pragma omp parallel num_threads(nthreads) reduction(| : failed)
  {
    double *H = (double *)malloc(sizeof(double) * ((long)nv * nv + nv + (long)kmax * nv + 8));
    double *c = H + (long)nv * nv;
    double *wj = c + nv;
    double *work = (double *)malloc(sizeof(double) * oracle_gi_work_size(nv, m));
    if (!H || !work) failed = 1;
#pragma omp for schedule(static) /* contiguous chunks: one thread streams one range of the batch */
    for (long b = 0; b < B; ++b) {
      if (failed) continue;
      for (long i = 0; i < (long)nv * nv; ++i) H[i] = 0.0;
      for (int i = 0; i < nv; ++i) {
        H[(long)i * nv + i] = damping; /* solve_ik.py:55 */
        c[i] = 0.0;                    /* solve_ik.py:56 */
      }
      for (int t = 0; t < T; ++t) {
        const int r0 = rows[t], k = rows[t + 1] - rows[t];
        const double *w = cost_batched ? cost + b * K + r0 : cost + r0;
        oracle_task_objective(k, nv, J + (b * K + r0) * nv, e + b * K + r0, w, gain[t], lm[t], H,
                              c, wj);
      }
      if (diag_extra)
        for (int i = 0; i < nv; ++i) H[(long)i * nv + i] += diag_extra[b];
      if (c_extra)
        for (int i = 0; i < nv; ++i) c[i] += c_extra[b * nv + i];
      if (H_out) memcpy(H_out + b * nv * nv, H, sizeof(double) * nv * nv);
      if (c_out) memcpy(c_out + b * nv, c, sizeof(double) * nv);
      if (solve) {
        int it = 0;
        int st = oracle_gi_solve_eq(nv, H, c, m, meq, m ? G + b * m * nv : NULL, m ? h + b * m : NULL,
                                    dq + b * nv, NULL, &it, 0, work);
        status[b] = st;
        if (iters) iters[b] = it;
      }
    }
    free(H);
    free(work);
  }
  return failed ? -1 : 0;
}
