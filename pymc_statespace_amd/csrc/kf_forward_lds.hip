// kf_forward_lds.hip -- fused Kalman filter forward pass, state resident in LDS.
//
// Replaces the reference's pytensor.scan(kalman_step) (pymc_statespace/filters/
// kalman_filter.py:126-163, 231-252) for a BATCH of parameter draws.  The whole time
// loop runs inside ONE kernel: a group of G lanes of a 64-wide wavefront owns one draw,
// the draw's (a, P, T, Z, RQR', H, c, d) tiles live in LDS for all T steps, and each
// step fuses   mask -> v, PZ', F, chol(F), K, ll -> a_f, P_f -> a_hat, P_hat
// plus the optional tape stores.  One workgroup = one wavefront (64 threads), so the
// step-internal syncs are wave-local (no cross-wave s_barrier traffic).
//
// Arithmetic notes (all fp64):
//  * F is factored by an unrolled in-register Cholesky (p <= 8) in EVERY lane -- no
//    lane exchange for the p x p work.  StandardFilter's dposv (kalman_filter.py:267)
//    and CholeskyFilter's dpotrf (:296) are the same factorisation; the two filters
//    differ only in how many log(2 pi) they add (flag KF_FLAG_MVN_CONST_EXACT).
//  * P_f uses the Joseph form expanded with M = P Z', P_f = P - K M' - M K' + (K F) K'
//    -- algebraically identical to (I-KZ) P (I-KZ)' + K H K' (:277) without forming
//    m x m x m products; measured |dlogp| vs the literal form ~5e-12 at T=1000, m=10.
//  * P_hat = T P_f T' + R Q R' is computed pairwise, P_hat[i][j] = P_hat[j][i] =
//    0.5 (S_ij + S_ji): the reference's symmetrisation (:221) for free.
//  * missing data: rows of Z and rows+columns of H of a NaN observation are dropped and a
//    unit diagonal is inserted in F ("row deletion"); a fully missing step leaves (a, P)
//    untouched and adds ll = 0, exactly as the reference's all_nan_flag path.
//  * P0 is taken to be symmetric (its upper triangle is what is read).
#include "kf_internal.h"

namespace {

__device__ __forceinline__ void wsync() { __syncthreads(); }  // block == one wavefront

template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }

// ---- unrolled p x p work, every lane redundantly --------------------------------
// L packed lower: idx(i,j) = i(i+1)/2 + j, j <= i
template <int P>
struct SmallChol {
  double L[P * (P + 1) / 2];
  double invd[P];
  __device__ __forceinline__ static constexpr int ix(int i, int j) { return i * (i + 1) / 2 + j; }

  // returns 0 or the 1-based index of the first non-positive pivot
  __device__ __forceinline__ int factor(const double* F /*LDS, p x p row-major*/) {
    int fail = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      double s = F[j * P + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[ix(j, k)] * L[ix(j, k)];
      if (!(s > 0.0) && fail == 0) fail = j + 1;
      double dj = sqrt(s);
      L[ix(j, j)] = dj;
      invd[j] = 1.0 / dj;
#pragma unroll
      for (int i = j + 1; i < P; ++i) {
        double t = F[i * P + j];
#pragma unroll
        for (int k = 0; k < j; ++k) t -= L[ix(i, k)] * L[ix(j, k)];
        L[ix(i, j)] = t * invd[j];
      }
    }
    return fail;
  }
  __device__ __forceinline__ void fwd(double (&x)[P]) const {  // x <- L^-1 x
#pragma unroll
    for (int i = 0; i < P; ++i) {
      double t = x[i];
#pragma unroll
      for (int k = 0; k < i; ++k) t -= L[ix(i, k)] * x[k];
      x[i] = t * invd[i];
    }
  }
  __device__ __forceinline__ void bwd(double (&x)[P]) const {  // x <- L^-T x
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
      double t = x[i];
#pragma unroll
      for (int k = i + 1; k < P; ++k) t -= L[ix(k, i)] * x[k];
      x[i] = t * invd[i];
    }
  }
  __device__ __forceinline__ double logdet_half() const {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) s += log(L[ix(i, i)]);
    return s;
  }
};

// F (LDS) -> quad = v' F^-1 v, half log det, K = M F^-1 (rows lig, lig+G, ... to LDS)
template <int P, int G>
__device__ __forceinline__ void dense_update_small(const double* F, const double* v, const double* M,
                                                   double* K, int m, int lig, double& quad,
                                                   double& ldh, int& fail) {
  SmallChol<P> ch;
  fail = ch.factor(F);
  double w[P];
#pragma unroll
  for (int q = 0; q < P; ++q) w[q] = v[q];
  ch.fwd(w);
  quad = 0.0;
#pragma unroll
  for (int q = 0; q < P; ++q) quad += w[q] * w[q];
  ldh = ch.logdet_half();
  for (int j = lig; j < m; j += G) {
    double x[P];
#pragma unroll
    for (int q = 0; q < P; ++q) x[q] = M[j * P + q];
    ch.fwd(x);
    ch.bwd(x);
#pragma unroll
    for (int q = 0; q < P; ++q) K[j * P + q] = x[q];
  }
}

// p > 8: lane 0 of the group factors in LDS, rows of K solved per lane from LDS.
template <int G>
__device__ __forceinline__ void dense_update_big(const double* F, double* L, const double* v, double* w,
                                                 const double* M, double* K, int m, int p, int lig,
                                                 double& quad, double& ldh, int& fail) {
  if (lig == 0) {
    int f = 0;
    for (int j = 0; j < p; ++j) {
      double s = F[j * p + j];
      for (int k = 0; k < j; ++k) s -= L[j * p + k] * L[j * p + k];
      if (!(s > 0.0) && f == 0) f = j + 1;
      double dj = sqrt(s);
      L[j * p + j] = dj;
      for (int i = j + 1; i < p; ++i) {
        double t = F[i * p + j];
        for (int k = 0; k < j; ++k) t -= L[i * p + k] * L[j * p + k];
        L[i * p + j] = t / dj;
      }
    }
    double q2 = 0.0, ld = 0.0;
    for (int i = 0; i < p; ++i) {
      double t = v[i];
      for (int k = 0; k < i; ++k) t -= L[i * p + k] * w[k];
      t /= L[i * p + i];
      w[i] = t;
      q2 += t * t;
      ld += log(L[i * p + i]);
    }
    w[p] = q2;      // spare slots of the 5p scratch block (w has 2p room)
    w[p + 1] = ld;
    w[p + 2] = (double)f;
  }
  wsync();
  quad = w[p];
  ldh = w[p + 1];
  fail = (int)w[p + 2];
  for (int j = lig; j < m; j += G) {
    for (int i = 0; i < p; ++i) {
      double t = M[j * p + i];
      for (int k = 0; k < i; ++k) t -= L[i * p + k] * K[j * p + k];
      K[j * p + i] = t / L[i * p + i];
    }
    for (int i = p - 1; i >= 0; --i) {
      double t = K[j * p + i];
      for (int k = i + 1; k < p; ++k) t -= L[k * p + i] * K[j * p + k];
      K[j * p + i] = t / L[i * p + i];
    }
  }
}

// KF_PMAX (compile-time, one translation unit each): largest p this object's kernels
// unroll in registers.  Keeps the p == 1 kernels lean (VGPRs set the waves per SIMD).
#ifndef KF_PMAX
#error "compile with -DKF_PMAX={0,1,2,4,8}"
#endif

template <int G>
__global__ __launch_bounds__(KF_WAVE) void kf_fwd_lds_kernel(const KfFwdArgs A) {
  extern __shared__ double smem[];
  constexpr int DPB = KF_WAVE / G;  // draws per block (= per wavefront)
  const int lane = threadIdx.x;
  const int lig = lane % G;
  const int grp = lane / G;
  const int m = A.m, p = A.p, r = A.r, n = A.T;
  const int mm = m * m, pm = p * m, pp = p * p;
  const int npair = m * (m + 1) / 2;
  // time-varying system matrices ([B][T][.][.], reference: time on axis 0 of a 3-D matrix,
  // filters/utilities.py:1-20): reloaded into the LDS tiles at the top of every step
  const bool tvT = A.flags & KF_FLAG_TV_T, tvZ = A.flags & KF_FLAG_TV_Z, tvR = A.flags & KF_FLAG_TV_R;
  const bool tvH = A.flags & KF_FLAG_TV_H, tvQ = A.flags & KF_FLAG_TV_Q, tvC = A.flags & KF_FLAG_TV_C;
  const bool tvD = A.flags & KF_FLAG_TV_D;
  const bool tv_any = (A.flags & KF_FLAG_TV_ANY) != 0;

  if (A.run_if && *A.run_if == 0) return;  // (fallback launch of kf_forward_whiten.hip: nothing to redo)
  long b = (long)blockIdx.x * DPB + grp;
  const bool live = b < A.B;  // tail groups compute a clone of the last draw, store nothing
  if (!live) b = A.B - 1;

  // ---- LDS carve-up ---------------------------------------------------------------
  int* pairtab = reinterpret_cast<int*>(smem);
  double* S = smem + kf_pair_table_doubles(m) + (size_t)grp * kf_fwd_region_doubles(m, p);
  double* Pa = S;            // current predicted covariance
  double* Pb = Pa + mm;      // filtered covariance
  double* W = Pb + mm;       // T P_f
  double* Tl = W + mm;
  double* RQR = Tl + mm;
  double* Zl = RQR + mm;     // p x m
  double* M = Zl + pm;       // m x p   (P Z')
  double* K = M + pm;        // m x p
  double* KF = K + pm;       // m x p   (K F)
  double* Hl = KF + pm;      // p x p
  double* F = Hl + pp;       // p x p
  double* Lb = F + pp;       // p x p   (big-p Cholesky only)
  double* al = Lb + pp;      // m
  double* af = al + m;       // m
  double* cl = af + m;       // m
  double* dl = cl + m;       // p
  double* vl = dl + p;       // p
  double* ym = vl + p;       // p
  double* ob = ym + p;       // p   1.0 observed / 0.0 missing
  double* wl = ob + p;       // p + 3 (big-p path keeps quad / logdet / fail behind w)
  (void)wl;

  for (int i = lane; i < m; i += KF_WAVE) {
    int base = i * m - (i * (i - 1)) / 2;
    for (int j = i; j < m; ++j) pairtab[base + (j - i)] = (i << 16) | j;
  }

  // ---- per-draw constants into LDS --------------------------------------------------
  {
    const double* gT = A.Tm + (tvT ? b * n : b) * mm;
    const double* gP = A.P0 + b * mm;
    for (int e = lig; e < mm; e += G) {
      Tl[e] = gT[e];
      Pa[e] = gP[e];
    }
    const double* gZ = A.Z + (tvZ ? b * n : b) * pm;
    for (int e = lig; e < pm; e += G) Zl[e] = gZ[e];
    const double* gH = A.H + (tvH ? b * n : b) * pp;
    for (int e = lig; e < pp; e += G) Hl[e] = gH[e];
    const double* ga = A.a0 + b * m;
    for (int e = lig; e < m; e += G) {
      al[e] = ga[e];
      cl[e] = A.c ? A.c[(tvC ? b * n : b) * m + e] : 0.0;
    }
    for (int e = lig; e < p; e += G) dl[e] = A.d ? A.d[(tvD ? b * n : b) * p + e] : 0.0;
    // R Q R' once per draw (the reference recomputes it every step, kalman_filter.py:218)
    const double* gR = A.R + (tvR ? b * n : b) * (long)m * r;
    const double* gQ = A.Q + (tvQ ? b * n : b) * (long)r * r;
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = 0.0;
      for (int l = 0; l < r; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
        s += rq * gR[j * r + l];
      }
      RQR[e] = s;
    }
  }
  wsync();
  // symmetrise P0's use: mirror the upper triangle (documented contract)
  for (int e = lig; e < npair; e += G) {
    int ij = pairtab[e];
    int i = ij >> 16, j = ij & 0xffff;
    Pa[j * m + i] = Pa[i * m + j];
  }
  // tape slot 0 of the predicted outputs = (a0, P0)   (_postprocess_scan_results :165-193)
  if (live) {
    if (A.pred_a) {
      double* o = A.pred_a + b * (long)(n + 1) * m;
      for (int e = lig; e < m; e += G) st_stream(o + e, al[e]);
    }
    if (A.pred_P) {
      double* o = A.pred_P + b * (long)(n + 1) * mm;
      const double* gP = A.P0 + b * mm;
      for (int e = lig; e < mm; e += G) st_stream(o + e, gP[e]);
    }
  }
  wsync();

  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L);

  double logp = 0.0;
  int info = 0;

  for (int t = 0; t < n; ++t) {
    if (tv_any && t > 0) {  // (t == 0 was loaded above)
      const long bt = b * n + t;
      if (tvT) for (int e = lig; e < mm; e += G) Tl[e] = A.Tm[bt * mm + e];
      if (tvZ) for (int e = lig; e < pm; e += G) Zl[e] = A.Z[bt * pm + e];
      if (tvH) for (int e = lig; e < pp; e += G) Hl[e] = A.H[bt * pp + e];
      if (tvC) for (int e = lig; e < m; e += G) cl[e] = A.c[bt * m + e];
      if (tvD) for (int e = lig; e < p; e += G) dl[e] = A.d[bt * p + e];
      if (tvR || tvQ) {
        const double* gR = A.R + (tvR ? bt : b) * (long)m * r;
        const double* gQ = A.Q + (tvQ ? bt : b) * (long)r * r;
        for (int e = lig; e < mm; e += G) {
          int i = e / m, j = e - i * m;
          double s = 0.0;
          for (int l = 0; l < r; ++l) {
            double rq = 0.0;
            for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
            s += rq * gR[j * r + l];
          }
          RQR[e] = s;
        }
      }
    }
    // ---- observation + mask (handle_missing_values :195-213) --------------------------
    for (int q = lig; q < p; q += G) {
      double yv = ybase[(long)t * p + q];
      bool miss = yv != yv;
      ym[q] = miss ? 0.0 : yv;
      ob[q] = miss ? 0.0 : 1.0;
    }
    wsync();
    double nobs = 0.0;
    for (int q = 0; q < p; ++q) nobs += ob[q];

    double ll;
    const double* Pf;   // filtered covariance lives here after the update
    const double* afp;  // filtered state

    if (!univariate) {
      // ---- v = y - Z a - d ;  M = P Z' --------------------------------------------------
      for (int q = lig; q < p; q += G) {
        double s = ym[q] + dsign * dl[q];
        for (int j = 0; j < m; ++j) s -= Zl[q * m + j] * al[j];
        vl[q] = ob[q] != 0.0 ? s : 0.0;
      }
      for (int e = lig; e < pm; e += G) {
        int j = e / p, q = e - j * p;
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += Pa[j * m + k] * Zl[q * m + k];
        M[e] = ob[q] != 0.0 ? s : 0.0;
      }
      wsync();
      // ---- F = Z M + H, deleted rows/cols -> unit diagonal -------------------------------
      for (int e = lig; e < pp; e += G) {
        int q1 = e / p, q2 = e - q1 * p;
        double s;
        if (ob[q1] != 0.0 && ob[q2] != 0.0) {
          s = Hl[e];
          for (int j = 0; j < m; ++j) s += Zl[q1 * m + j] * M[j * p + q2];
        } else {
          s = (q1 == q2) ? 1.0 : 0.0;
        }
        F[e] = s;
      }
      wsync();
      // ---- chol(F), quad form, log det, gain ---------------------------------------------
      double quad, ldh;
      int fail;
#if KF_PMAX == 1
      dense_update_small<1, G>(F, vl, M, K, m, lig, quad, ldh, fail);
#elif KF_PMAX == 2
      dense_update_small<2, G>(F, vl, M, K, m, lig, quad, ldh, fail);
#elif KF_PMAX == 4
      if (p == 3) dense_update_small<3, G>(F, vl, M, K, m, lig, quad, ldh, fail);
      else dense_update_small<4, G>(F, vl, M, K, m, lig, quad, ldh, fail);
#elif KF_PMAX == 8
      switch (p) {
        case 5: dense_update_small<5, G>(F, vl, M, K, m, lig, quad, ldh, fail); break;
        case 6: dense_update_small<6, G>(F, vl, M, K, m, lig, quad, ldh, fail); break;
        case 7: dense_update_small<7, G>(F, vl, M, K, m, lig, quad, ldh, fail); break;
        default: dense_update_small<8, G>(F, vl, M, K, m, lig, quad, ldh, fail); break;
      }
#else
      dense_update_big<G>(F, Lb, vl, wl, M, K, m, p, lig, quad, ldh, fail);
#endif
      if (fail != 0 && info == 0) info = ((t + 1) << 8) | fail;
      double nconst = exact_const ? nobs : (A.filter == KF_FILTER_CHOLESKY ? (double)p : 1.0);
      ll = (nobs == 0.0) ? 0.0 : (-0.5 * (nconst * KF_LOG_2PI + quad) - ldh);
      wsync();
      // ---- a_f = a + K v ;  KF = K F ------------------------------------------------------
      for (int j = lig; j < m; j += G) {
        double s = al[j];
        for (int q = 0; q < p; ++q) s += K[j * p + q] * vl[q];
        af[j] = s;
      }
      for (int e = lig; e < pm; e += G) {
        int j = e / p, q = e - j * p;
        double s = 0.0;
        for (int q2 = 0; q2 < p; ++q2) s += K[j * p + q2] * F[q2 * p + q];
        KF[e] = s;
      }
      wsync();
      // ---- P_f = P - K M' - M K' + (K F) K'   (expanded Joseph form) -----------------------
      for (int e = lig; e < npair; e += G) {
        int ij = pairtab[e];
        int i = ij >> 16, j = ij & 0xffff;
        double s = Pa[i * m + j];
        for (int q = 0; q < p; ++q) {
          double kj = K[j * p + q];
          s -= K[i * p + q] * M[j * p + q];
          s -= M[i * p + q] * kj;
          s += KF[i * p + q] * kj;
        }
        Pb[i * m + j] = s;
        Pb[j * m + i] = s;
      }
      Pf = Pb;
      afp = af;
    } else {
      // ---- UnivariateFilter: p sequential scalar updates (:460-505) --------------------------
      double llsum = 0.0, ncount = 0.0;
      for (int q = 0; q < p; ++q) {
        const bool obs = ob[q] != 0.0;
        double vq = ym[q] - dl[q];
        for (int j = 0; j < m; ++j) vq -= Zl[q * m + j] * al[j];
        for (int j = lig; j < m; j += G) {
          double s = 0.0;
          for (int k = 0; k < m; ++k) s += Pa[j * m + k] * Zl[q * m + k];
          M[j] = obs ? s : 0.0;
        }
        if (!obs) vq = 0.0;  // Z row and y are zeroed by the mask; d is 0 through the public API
        wsync();
        double Fq = obs ? Hl[q * p + q] : 0.0;
        if (obs)
          for (int j = 0; j < m; ++j) Fq += Zl[q * m + j] * M[j];
        const bool degenerate = (Fq == 0.0) || !obs;
        if (degenerate) Fq += 1e-8;
        const double keep = degenerate ? 0.0 : 1.0;
        const double invF = 1.0 / Fq;
        for (int j = lig; j < m; j += G) al[j] += (M[j] * invF * keep) * (vq * keep);
        for (int e = lig; e < npair; e += G) {
          int ij = pairtab[e];
          int i = ij >> 16, j = ij & 0xffff;
          double ki = M[i] * invF * keep, kj = M[j] * invF * keep;
          double s = Pa[i * m + j] - ki * kj * Fq * keep;
          Pa[i * m + j] = s;
          Pa[j * m + i] = s;
        }
        double lli = (log(Fq) + vq * vq * invF) * keep;
        llsum += lli;
        if (exact_const) ncount += obs ? 1.0 : 0.0;
        else ncount += (lli != 0.0) ? 1.0 : 0.0;
        wsync();
      }
      ll = -0.5 * (ncount * KF_LOG_2PI + llsum);
      for (int j = lig; j < m; j += G) af[j] = al[j];
      Pf = Pa;
      afp = af;
    }
    wsync();
    logp += ll;

    // ---- filtered tapes ------------------------------------------------------------------
    if (live) {
      if (A.filt_a) {
        double* o = A.filt_a + (b * (long)n + t) * m;
        for (int e = lig; e < m; e += G) st_stream(o + e, afp[e]);
      }
      if (A.filt_P) {
        double* o = A.filt_P + (b * (long)n + t) * mm;
        for (int e = lig; e < mm; e += G) st_stream(o + e, Pf[e]);
      }
      if (A.ll_obs && lig == 0) st_stream(A.ll_obs + b * (long)n + t, ll);
    }

    // ---- predict (:215-223): a_hat = T a_f + c ; W = T P_f -------------------------------------
    for (int i = lig; i < m; i += G) {
      double s = cl[i];
      for (int k = 0; k < m; ++k) s += Tl[i * m + k] * afp[k];
      al[i] = s;
    }
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[i * m + k] * Pf[k * m + j];
      W[e] = s;
    }
    wsync();
    // P_hat = sym(W T' + RQR')
    for (int e = lig; e < npair; e += G) {
      int ij = pairtab[e];
      int i = ij >> 16, j = ij & 0xffff;
      double sij = RQR[i * m + j], sji = RQR[j * m + i];
      for (int k = 0; k < m; ++k) {
        sij += W[i * m + k] * Tl[j * m + k];
        sji += W[j * m + k] * Tl[i * m + k];
      }
      double s = 0.5 * (sij + sji);
      Pa[i * m + j] = s;
      Pa[j * m + i] = s;
    }
    wsync();
    // ---- predicted tapes ---------------------------------------------------------------------
    if (live) {
      if (A.pred_a) {
        double* o = A.pred_a + (b * (long)(n + 1) + t + 1) * m;
        for (int e = lig; e < m; e += G) st_stream(o + e, al[e]);
      }
      if (A.pred_P) {
        double* o = A.pred_P + (b * (long)(n + 1) + t + 1) * mm;
        for (int e = lig; e < mm; e += G) st_stream(o + e, Pa[e]);
      }
    }
  }

  if (live && lig == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

template <int G>
int launch_g(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  const int dpb = KF_WAVE / G;
  size_t lds = sizeof(double) *
               ((size_t)kf_pair_table_doubles(a.m) + (size_t)dpb * kf_fwd_region_doubles(a.m, a.p));
  if (lds > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "forward(lds): m=%d p=%d needs %zu B of LDS per workgroup (> %d)", a.m, a.p, lds,
               KF_LDS_BYTES_MAX);
    return KF_ERR_UNSUPPORTED;
  }
  KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_fwd_lds_kernel<G>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long blocks = (a.B + dpb - 1) / dpb;
  hipLaunchKernelGGL(kf_fwd_lds_kernel<G>, dim3((unsigned)blocks), dim3(KF_WAVE), lds, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

#define KF_CAT2(a, b) a##b
#define KF_CAT(a, b) KF_CAT2(a, b)
int KF_CAT(kf_launch_forward_lds_p, KF_PMAX)(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  int G = kf_pick_lanes(h, a.m, a.B);
  // shrink draws-per-wave until the workgroup's tiles fit in LDS
  while (G < KF_WAVE) {
    size_t lds = sizeof(double) * ((size_t)kf_pair_table_doubles(a.m) +
                                   (size_t)(KF_WAVE / G) * kf_fwd_region_doubles(a.m, a.p));
    if (lds <= KF_LDS_BYTES_MAX / 2) break;
    G *= 2;
  }
  switch (G) {
    case 4: return launch_g<4>(h, a, stream);
    case 8: return launch_g<8>(h, a, stream);
    case 16: return launch_g<16>(h, a, stream);
    case 32: return launch_g<32>(h, a, stream);
    default: return launch_g<64>(h, a, stream);
  }
}
