/* kf_oracle.c -- plain C restatement of the reference Kalman filter / RTS smoother.
 *
 * TEST INFRASTRUCTURE ONLY: the CPU checker for full-size parity runs and the
 * `cpu_baseline` leg of bench.py (kind "port").  Nothing under pymc_statespace_amd/
 * links, loads or calls this file.
 *
 * It follows the reference's own operation sequence, per draw and per time step
 * (paths relative to /root/reference/pymc_statespace):
 *   handle_missing_values   filters/kalman_filter.py:195-213
 *   predict                 filters/kalman_filter.py:215-223  (T P T' + R Q R', then 0.5(P+P'))
 *   kalman_step             filters/kalman_filter.py:231-252  (update, then predict)
 *   StandardFilter.update   filters/kalman_filter.py:255-284  (F^-1 by a positive-definite solve,
 *                           K = PZ' F^-1, Joseph form with full m x m x m products, log det F)
 *   CholeskyFilter.update   filters/kalman_filter.py:287-318  (correct L^-T solve)
 *   UnivariateFilter        filters/kalman_filter.py:444-505
 *   SingleTimeseriesFilter  filters/kalman_filter.py:321-351
 *   _postprocess            filters/kalman_filter.py:165-193
 *   KalmanSmoother          filters/kalman_smoother.py:56-104 (pinv by symmetric Jacobi eigen-
 *                           decomposition with numpy's rcond = 1e-15 cutoff)
 * Unlike the reference it hoists R Q R' out of the time loop (a pure saving that favours
 * the CPU baseline) and loops over B draws (OpenMP, one draw per thread at a time).
 *
 * Parity status: PINNED via tests/test_oracle_c.py -- checked against oracle/kalman_numpy.py
 * and, through it, the statsmodels golden vectors in tests/golden/.
 *
 * Layouts: as include/kf.h (draw-major, row-major), host pointers.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LOG_2PI 1.8378770664093454835606594728112

enum { F_STANDARD = 0, F_CHOLESKY = 1, F_UNIVARIATE = 2, F_SINGLE = 3 };
enum { FLAG_MVN_EXACT = 1u, FLAG_Y_PER_DRAW = 2u };

/* C (n x q) = A (n x k) B (k x q) */
static void mm(const double* A, const double* B, double* C, int n, int k, int q) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < q; ++j) {
      double s = 0.0;
      for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * q + j];
      C[i * q + j] = s;
    }
}
/* C (n x q) = A (n x k) B' (q x k) */
static void mmt(const double* A, const double* B, double* C, int n, int k, int q) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < q; ++j) {
      double s = 0.0;
      for (int l = 0; l < k; ++l) s += A[i * k + l] * B[j * k + l];
      C[i * q + j] = s;
    }
}
/* in-place lower Cholesky; returns 0 or 1-based failing minor */
static int chol(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    if (!(s > 0.0)) return j + 1;
    double d = sqrt(s);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / d;
    }
  }
  return 0;
}
static void solve_L(const double* L, double* x, int n) { /* x <- L^-1 x */
  for (int i = 0; i < n; ++i) {
    double t = x[i];
    for (int k = 0; k < i; ++k) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
}
static void solve_LT(const double* L, double* x, int n) { /* x <- L^-T x */
  for (int i = n - 1; i >= 0; --i) {
    double t = x[i];
    for (int k = i + 1; k < n; ++k) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}
/* log|det A| * sign via LU with partial pivoting (numpy.linalg.det route); A destroyed */
static double det_lu(double* A, int n) {
  double det = 1.0;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); piv = i; }
    if (best == 0.0) return 0.0;
    if (piv != k) {
      for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
      det = -det;
    }
    det *= A[k * n + k];
    for (int i = k + 1; i < n; ++i) {
      double f = A[i * n + k] / A[k * n + k];
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
    }
  }
  return det;
}

typedef struct {
  double *P, *Pf, *W, *W2, *RQR, *RQ, *Zm, *Hm, *M, *K, *IKZ, *F, *Fc, *Finv, *Fi, *lu, *KH, *a, *af, *v, *ym, *x, *col, *w;
  int* miss;
} scratch_t;

static size_t scratch_doubles(int m, int p, int r) {
  return (size_t)6 * m * m + (size_t)m * r + 4 * (size_t)p * m + 6 * (size_t)p * p + 3 * (size_t)m + 6 * (size_t)p + 16;
}
static void carve(scratch_t* s, double* b, int m, int p, int r) {
  int mm_ = m * m, pm = p * m, pp = p * p;
  s->P = b; b += mm_; s->Pf = b; b += mm_; s->W = b; b += mm_; s->W2 = b; b += mm_;
  s->RQR = b; b += mm_; s->IKZ = b; b += mm_; s->RQ = b; b += m * r;
  s->Zm = b; b += pm; s->M = b; b += pm; s->K = b; b += pm; s->KH = b; b += pm;
  s->Hm = b; b += pp; s->F = b; b += pp; s->Fc = b; b += pp; s->Finv = b; b += pp;
  s->Fi = b; b += pp; s->lu = b; b += pp;
  s->a = b; b += m; s->af = b; b += m; s->x = b; b += m;
  s->v = b; b += p; s->ym = b; b += p; s->col = b; b += p; s->w = b; b += p;
  s->miss = (int*)b;
}

static void filter_one(int T, int p, int m, int r, int filter, unsigned flags, const double* y,
                       const double* a0, const double* P0, const double* Tm, const double* Z,
                       const double* R, const double* H, const double* Q, const double* c,
                       const double* d, double* logp, double* ll_obs, double* filt_a, double* pred_a,
                       double* filt_P, double* pred_P, int32_t* info, double* buf) {
  scratch_t s;
  carve(&s, buf, m, p, r);
  const int mm_ = m * m;
  mm(R, Q, s.RQ, m, r, r);
  mmt(s.RQ, R, s.RQR, m, r, m);
  memcpy(s.a, a0, sizeof(double) * m);
  memcpy(s.P, P0, sizeof(double) * mm_);
  if (pred_a) memcpy(pred_a, a0, sizeof(double) * m);
  if (pred_P) memcpy(pred_P, P0, sizeof(double) * mm_);
  double total = 0.0;
  int bad = 0;
  const int exact = (flags & FLAG_MVN_EXACT) != 0;

  for (int t = 0; t < T; ++t) {
    /* mask */
    int nobs = 0;
    for (int q = 0; q < p; ++q) {
      double yv = y[(size_t)t * p + q];
      s.miss[q] = yv != yv;
      s.ym[q] = s.miss[q] ? 0.0 : yv;
      nobs += !s.miss[q];
    }
    const int all_nan = nobs == 0;
    for (int q = 0; q < p; ++q)
      for (int j = 0; j < m; ++j) s.Zm[q * m + j] = s.miss[q] ? 0.0 : Z[q * m + j];
    for (int q = 0; q < p; ++q)
      for (int q2 = 0; q2 < p; ++q2) /* rows AND columns: proper deletion (== reference for whole rows) */
        s.Hm[q * p + q2] = (s.miss[q] || s.miss[q2]) ? 0.0 : H[q * p + q2];
    double ll = 0.0;

    if (filter == F_UNIVARIATE) {
      double llsum = 0.0;
      int ncount = 0;
      for (int q = 0; q < p; ++q) {
        const double* Zr = s.Zm + q * m;
        double v = s.ym[q] - (d ? d[q] : 0.0);
        for (int j = 0; j < m; ++j) v -= Zr[j] * s.a[j];
        for (int j = 0; j < m; ++j) {
          double t_ = 0.0;
          for (int k = 0; k < m; ++k) t_ += s.P[j * m + k] * Zr[k];
          s.x[j] = t_;
        }
        double F = s.Hm[q * p + q];
        for (int j = 0; j < m; ++j) F += Zr[j] * s.x[j];
        int flag = (F == 0.0) || s.miss[q];
        F += 1e-8 * flag;
        double keep = 1.0 - flag;
        if (s.miss[q]) v = 0.0;
        for (int j = 0; j < m; ++j) s.x[j] = s.x[j] / F * keep; /* K */
        for (int j = 0; j < m; ++j) s.a[j] += s.x[j] * v * keep;
        for (int i = 0; i < m; ++i)
          for (int j = 0; j < m; ++j) s.P[i * m + j] -= s.x[i] * s.x[j] * F * keep;
        double lli = (log(F) + v * v / F) * keep;
        llsum += lli;
        ncount += exact ? !s.miss[q] : (lli != 0.0);
      }
      ll = -0.5 * (ncount * LOG_2PI + llsum);
      memcpy(s.af, s.a, sizeof(double) * m);
      memcpy(s.Pf, s.P, sizeof(double) * mm_);
    } else {
      /* v, PZT, F */
      const double dsign = (filter == F_SINGLE) ? 1.0 : -1.0;
      for (int q = 0; q < p; ++q) {
        double t_ = s.ym[q] + dsign * (d ? d[q] : 0.0);
        for (int j = 0; j < m; ++j) t_ -= s.Zm[q * m + j] * s.a[j];
        s.v[q] = s.miss[q] ? 0.0 : t_;
      }
      mmt(s.P, s.Zm, s.M, m, m, p);  /* PZT (m x p) */
      mm(s.Zm, s.M, s.F, p, m, p);
      for (int e = 0; e < p * p; ++e) s.F[e] += s.Hm[e];
      memcpy(s.Fc, s.F, sizeof(double) * p * p);
      for (int q = 0; q < p; ++q)
        if (s.miss[q]) s.Fc[q * p + q] += 1.0; /* F + I*flag, and unit diagonal on deleted rows */
      double logdet, quad;
      memcpy(s.Finv, s.Fc, sizeof(double) * p * p); /* Finv buffer holds L */
      int f = chol(s.Finv, p);
      if (f && !bad) bad = ((t + 1) << 8) | f;
      if (filter == F_STANDARD || filter == F_SINGLE) {
        /* F_inv = solve(F, I, assume_a="pos"); K = PZT F_inv; log det F by LU */
        double* L = s.Finv;
        double* Fi = s.Fi;
        for (int j = 0; j < p; ++j) {
          for (int i = 0; i < p; ++i) s.col[i] = (i == j) ? 1.0 : 0.0;
          solve_L(L, s.col, p);
          solve_LT(L, s.col, p);
          for (int i = 0; i < p; ++i) Fi[i * p + j] = s.col[i];
        }
        mm(s.M, Fi, s.K, m, p, p);
        quad = 0.0;
        for (int i = 0; i < p; ++i)
          for (int j = 0; j < p; ++j) quad += s.v[i] * Fi[i * p + j] * s.v[j];
        memcpy(s.lu, s.Fc, sizeof(double) * p * p);
        logdet = log(det_lu(s.lu, p));
      } else {
        double* L = s.Finv;
        for (int j = 0; j < m; ++j) {
          for (int q = 0; q < p; ++q) s.col[q] = s.M[j * p + q];
          solve_L(L, s.col, p);
          solve_LT(L, s.col, p);
          for (int q = 0; q < p; ++q) s.K[j * p + q] = all_nan ? 0.0 : s.col[q];
        }
        for (int q = 0; q < p; ++q) s.w[q] = s.v[q];
        solve_L(L, s.w, p);
        solve_LT(L, s.w, p);
        quad = 0.0;
        for (int q = 0; q < p; ++q) quad += s.v[q] * s.w[q];
        logdet = 0.0;
        for (int q = 0; q < p; ++q) logdet += 2.0 * log(L[q * p + q]);
      }
      double nconst = exact ? (double)nobs : (filter == F_CHOLESKY ? (double)p : 1.0);
      ll = all_nan ? 0.0 : -0.5 * (nconst * LOG_2PI + logdet + quad);
      /* I - K Z ; a_f ; Joseph */
      mm(s.K, s.Zm, s.IKZ, m, p, m);
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) s.IKZ[i * m + j] = (i == j ? 1.0 : 0.0) - s.IKZ[i * m + j];
      for (int j = 0; j < m; ++j) {
        double t_ = s.a[j];
        for (int q = 0; q < p; ++q) t_ += s.K[j * p + q] * s.v[q];
        s.af[j] = t_;
      }
      mm(s.IKZ, s.P, s.W, m, m, m);
      mmt(s.W, s.IKZ, s.Pf, m, m, m);
      mm(s.K, s.Hm, s.KH, m, p, p);
      mmt(s.KH, s.K, s.W, m, p, m);
      for (int e = 0; e < mm_; ++e) s.Pf[e] += s.W[e];
    }
    total += ll;
    if (ll_obs) ll_obs[t] = ll;
    if (filt_a) memcpy(filt_a + (size_t)t * m, s.af, sizeof(double) * m);
    if (filt_P) memcpy(filt_P + (size_t)t * mm_, s.Pf, sizeof(double) * mm_);
    /* predict */
    for (int i = 0; i < m; ++i) {
      double t_ = c ? c[i] : 0.0;
      for (int k = 0; k < m; ++k) t_ += Tm[i * m + k] * s.af[k];
      s.a[i] = t_;
    }
    mm(Tm, s.Pf, s.W, m, m, m);
    mmt(s.W, Tm, s.W2, m, m, m);
    for (int e = 0; e < mm_; ++e) s.W2[e] += s.RQR[e];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) s.P[i * m + j] = 0.5 * (s.W2[i * m + j] + s.W2[j * m + i]);
    if (pred_a) memcpy(pred_a + (size_t)(t + 1) * m, s.a, sizeof(double) * m);
    if (pred_P) memcpy(pred_P + (size_t)(t + 1) * mm_, s.P, sizeof(double) * mm_);
  }
  *logp = total;
  if (info) *info = bad;
}

int kf_oracle_filter(int64_t B, int T, int p, int m, int r, int filter, unsigned flags,
                     const double* y, const double* a0, const double* P0, const double* Tm,
                     const double* Z, const double* R, const double* H, const double* Q,
                     const double* c, const double* d, double* logp, double* ll_obs, double* filt_a,
                     double* pred_a, double* filt_P, double* pred_P, int32_t* info, int nthreads) {
  if (B < 1 || T < 1 || p < 1 || m < 1 || r < 1) return -1;
  if (filter == F_SINGLE && p != 1) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  const size_t nsc = scratch_doubles(m, p, r);
  const size_t mm_ = (size_t)m * m;
#pragma omp parallel
  {
    double* buf = (double*)malloc(sizeof(double) * nsc);
#pragma omp for schedule(dynamic, 4)
    for (int64_t b = 0; b < B; ++b) {
      const double* yb = y + ((flags & FLAG_Y_PER_DRAW) ? (size_t)b * T * p : 0);
      filter_one(T, p, m, r, filter, flags, yb, a0 + b * m, P0 + b * mm_, Tm + b * mm_,
                 Z + (size_t)b * p * m, R + (size_t)b * m * r, H + (size_t)b * p * p,
                 Q + (size_t)b * r * r, c ? c + b * m : NULL, d ? d + b * p : NULL, logp + b,
                 ll_obs ? ll_obs + (size_t)b * T : NULL, filt_a ? filt_a + (size_t)b * T * m : NULL,
                 pred_a ? pred_a + (size_t)b * (T + 1) * m : NULL,
                 filt_P ? filt_P + (size_t)b * T * mm_ : NULL,
                 pred_P ? pred_P + (size_t)b * (T + 1) * mm_ : NULL, info ? info + b : NULL, buf);
    }
    free(buf);
  }
  return 0;
}

/* symmetric pseudo-inverse by cyclic Jacobi: A (n x n, symmetric) -> Ap; numpy.linalg.pinv
 * semantics: singular values (= |eigenvalues|) <= rcond * max are treated as zero. */
static void pinv_sym(const double* A, double* Ap, int n, double rcond, double* V, double* D) {
  memcpy(D, A, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        if (i == j) diag += D[i * n + j] * D[i * n + j];
        else off += D[i * n + j] * D[i * n + j];
      }
    if (off <= 1e-60 * diag || off == 0.0) break;
    for (int pq = 0; pq < n - 1; ++pq)
      for (int q = pq + 1; q < n; ++q) {
        double apq = D[pq * n + q];
        if (apq == 0.0) continue;
        double app = D[pq * n + pq], aqq = D[q * n + q];
        double tau = (aqq - app) / (2.0 * apq);
        double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
        for (int k = 0; k < n; ++k) {
          double dkp = D[k * n + pq], dkq = D[k * n + q];
          D[k * n + pq] = cs * dkp - sn * dkq;
          D[k * n + q] = sn * dkp + cs * dkq;
        }
        for (int k = 0; k < n; ++k) {
          double dpk = D[pq * n + k], dqk = D[q * n + k];
          D[pq * n + k] = cs * dpk - sn * dqk;
          D[q * n + k] = sn * dpk + cs * dqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = V[k * n + pq], vkq = V[k * n + q];
          V[k * n + pq] = cs * vkp - sn * vkq;
          V[k * n + q] = sn * vkp + cs * vkq;
        }
      }
  }
  double smax = 0.0;
  for (int i = 0; i < n; ++i) if (fabs(D[i * n + i]) > smax) smax = fabs(D[i * n + i]);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) {
        double ev = D[k * n + k];
        if (fabs(ev) > rcond * smax) s += V[i * n + k] * V[j * n + k] / ev;
      }
      Ap[i * n + j] = s;
    }
}

int kf_oracle_smoother(int64_t B, int T, int m, int r, const double* Tm, const double* R,
                       const double* Q, const double* filt_a, const double* filt_P,
                       double* smooth_a, double* smooth_P, int nthreads) {
  if (B < 1 || T < 1 || m < 1 || r < 1) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  const size_t mm_ = (size_t)m * m;
#pragma omp parallel
  {
    double* buf = (double*)malloc(sizeof(double) * (9 * mm_ + (size_t)m * r + 4 * m));
    double *RQ = buf, *RQR = RQ + (size_t)m * r, *Ph = RQR + mm_, *Pi = Ph + mm_, *G = Pi + mm_,
           *W = G + mm_, *W2 = W + mm_, *V = W2 + mm_, *D = V + mm_, *Ps = D + mm_, *ah = Ps + mm_,
           *as = ah + m, *dx = as + m;
#pragma omp for schedule(dynamic, 4)
    for (int64_t b = 0; b < B; ++b) {
      const double* Tb = Tm + b * mm_;
      const double* fa = filt_a + (size_t)b * T * m;
      const double* fP = filt_P + (size_t)b * T * mm_;
      double* sa = smooth_a + (size_t)b * T * m;
      double* sP = smooth_P + (size_t)b * T * mm_;
      mm(R + (size_t)b * m * r, Q + (size_t)b * r * r, RQ, m, r, r);
      mmt(RQ, R + (size_t)b * m * r, RQR, m, r, m);
      memcpy(sa + (size_t)(T - 1) * m, fa + (size_t)(T - 1) * m, sizeof(double) * m);
      memcpy(sP + (size_t)(T - 1) * mm_, fP + (size_t)(T - 1) * mm_, sizeof(double) * mm_);
      memcpy(as, fa + (size_t)(T - 1) * m, sizeof(double) * m);
      memcpy(Ps, fP + (size_t)(T - 1) * mm_, sizeof(double) * mm_);
      for (int t = T - 2; t >= 0; --t) {
        const double* a = fa + (size_t)t * m;
        const double* P = fP + (size_t)t * mm_;
        for (int i = 0; i < m; ++i) {
          double s = 0.0;
          for (int k = 0; k < m; ++k) s += Tb[i * m + k] * a[k];
          ah[i] = s;
        }
        mm(Tb, P, W, m, m, m);
        mmt(W, Tb, Ph, m, m, m);
        for (size_t e = 0; e < mm_; ++e) Ph[e] += RQR[e];
        pinv_sym(Ph, Pi, m, 1e-15, V, D);
        mm(Pi, W, W2, m, m, m);              /* pinv(P_hat) T P   */
        for (int i = 0; i < m; ++i)           /* gain = (.)'       */
          for (int j = 0; j < m; ++j) G[i * m + j] = W2[j * m + i];
        for (int i = 0; i < m; ++i) dx[i] = as[i] - ah[i];
        for (int i = 0; i < m; ++i) {
          double s = a[i];
          for (int k = 0; k < m; ++k) s += G[i * m + k] * dx[k];
          as[i] = s;
        }
        for (size_t e = 0; e < mm_; ++e) W[e] = Ps[e] - Ph[e];
        mm(G, W, W2, m, m, m);
        mmt(W2, G, Ps, m, m, m);
        for (size_t e = 0; e < mm_; ++e) Ps[e] += P[e];
        memcpy(sa + (size_t)t * m, as, sizeof(double) * m);
        memcpy(sP + (size_t)t * mm_, Ps, sizeof(double) * mm_);
      }
    }
    free(buf);
  }
  return 0;
}

int kf_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
