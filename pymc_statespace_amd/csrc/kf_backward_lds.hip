// kf_backward_lds.hip -- reverse-mode gradient of the Kalman log-likelihood, fused per draw.
//
// The reference has no hand-written gradient: PyMC differentiates the filter's scan with
// PyTensor autodiff (reference README.md:6, core/statespace.py:174).  This kernel is the
// hand-derived adjoint of the reference operation sequence (StandardFilter.update +
// predict, pymc_statespace/filters/kalman_filter.py:215-223, 255-284), every matrix entry an
// independent variable.  The derivation is written out and checked against torch.autograd
// in oracle/kalman_adjoint_numpy.py + tests/test_oracle_torch.py; the symbols below are the
// ones used there.
//
// One kernel runs the whole reverse time loop.  A group of G lanes owns one draw; the draw's
// static tiles (T, Z, H, c, d), the adjoint state (ab, Pb) and the accumulators
// (Tb, Zb, Hb, cb, db, Gs = sum of symmetrised Pb) stay in LDS for all T steps, while the
// predicted tape (a_pred[t], P_pred[t]) written by the forward pass is streamed from HBM.
// Each step recomputes the cheap O(m^2 p) update quantities instead of taping them
// (tape traffic stays at one m x m matrix per draw-step) and spends its flops on four
// m x m x m products:  T P_f,  Gm (T P_f),  Gm T,  T' (Gm T).
//
// Every filter type shares this adjoint: for diagonal H the Univariate/Cholesky/Single
// Time-varying system matrices (KF_FLAG_TV_*, [B][T][.][.]): step t loads slice t of each flagged matrix, and the
// gradient of a flagged matrix is a tape of the same shape -- its accumulator is written to slice t and cleared at the
// end of step t (R and Q: from the step's own Gm, since R_t Q_t R_t' enters P_pred[t+1] only).
//
// filters evaluate the same function as the Standard one (tests/test_oracle_torch.py);
// Univariate reads diag(H) only (:493), so off-diagonal H is dropped on load and its gradient
// is zero; Single flips the sign of d (:335).
#include "kf_internal.h"

#ifndef KF_PMAX
#error "compile with -DKF_PMAX={0,1,2,4,8}"
#endif

namespace {

__device__ __forceinline__ void wsync() { __syncthreads(); }

template <int P>
struct SmallChol {
  double L[P * (P + 1) / 2];
  double invd[P];
  __device__ __forceinline__ static constexpr int ix(int i, int j) { return i * (i + 1) / 2 + j; }
  __device__ __forceinline__ void factor(const double* F) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      double s = F[j * P + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[ix(j, k)] * L[ix(j, k)];
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
  }
  __device__ __forceinline__ void solve(double (&x)[P]) const {  // x <- F^-1 x
#pragma unroll
    for (int i = 0; i < P; ++i) {
      double t = x[i];
#pragma unroll
      for (int k = 0; k < i; ++k) t -= L[ix(i, k)] * x[k];
      x[i] = t * invd[i];
    }
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
      double t = x[i];
#pragma unroll
      for (int k = i + 1; k < P; ++k) t -= L[ix(k, i)] * x[k];
      x[i] = t * invd[i];
    }
  }
};

// Fi = F^-1 (p x p, LDS): every lane factors F in registers, lane q < p solves column q.
template <int P, int G>
__device__ __forceinline__ void inverse_small(const double* F, double* Fi, int lig) {
  SmallChol<P> ch;
  ch.factor(F);
  for (int q = lig; q < P; q += G) {
    double x[P];
#pragma unroll
    for (int i = 0; i < P; ++i) x[i] = (i == q) ? 1.0 : 0.0;
    ch.solve(x);
#pragma unroll
    for (int i = 0; i < P; ++i) Fi[i * P + q] = x[i];
  }
}

// p > 8: lane 0 factors in LDS (L), then lane q solves column q reading L.
template <int G>
__device__ __forceinline__ void inverse_big(const double* F, double* L, double* Fi, int p, int lig) {
  if (lig == 0) {
    for (int j = 0; j < p; ++j) {
      double s = F[j * p + j];
      for (int k = 0; k < j; ++k) s -= L[j * p + k] * L[j * p + k];
      double dj = sqrt(s);
      L[j * p + j] = dj;
      for (int i = j + 1; i < p; ++i) {
        double t = F[i * p + j];
        for (int k = 0; k < j; ++k) t -= L[i * p + k] * L[j * p + k];
        L[i * p + j] = t / dj;
      }
    }
  }
  wsync();
  for (int q = lig; q < p; q += G) {
    for (int i = 0; i < p; ++i) {
      double t = (i == q) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) t -= L[i * p + k] * Fi[k * p + q];
      Fi[i * p + q] = t / L[i * p + i];
    }
    for (int i = p - 1; i >= 0; --i) {
      double t = Fi[i * p + q];
      for (int k = i + 1; k < p; ++k) t -= L[k * p + i] * Fi[k * p + q];
      Fi[i * p + q] = t / L[i * p + i];
    }
  }
}

__host__ __device__ inline int bwd_region_doubles(int m, int p) {
  return (9 * m * m + 10 * p * m + 10 * p * p + 6 * m + 6 * p + 2) | 1;
}

// MT > 0: compile-time k_states (loops unroll, LDS offsets become immediates); 0 = run time.
template <int G, int MT>
__global__ __launch_bounds__(KF_WAVE) void kf_bwd_lds_kernel(const KfBwdArgs A) {
  extern __shared__ double smem[];
  if (A.run_if && *A.run_if == 0) return;  // (the dense redo behind a whitened value-and-gradient call: nothing to redo)
  constexpr int DPB = KF_WAVE / G;
  const int lane = threadIdx.x;
  const int lig = lane % G;
  const int grp = lane / G;
  const int m = MT > 0 ? MT : A.m;
#if KF_PMAX == 1
  const int p = 1;
#else
  const int p = A.p;
#endif
  const int r = A.r, n = A.T;
  const int mm = m * m, pm = p * m, pp = p * p;
  const int npair = m * (m + 1) / 2;

  long b = (long)blockIdx.x * DPB + grp;
  const bool live = b < A.B;
  if (!live) b = A.B - 1;

  int* pairtab = reinterpret_cast<int*>(smem);
  double* S = smem + kf_pair_table_doubles(m) + (size_t)grp * bwd_region_doubles(m, p);
  // m x m
  double* Tl = S;
  double* Tb = Tl + mm;
  double* Gs = Tb + mm;
  double* Pb = Gs + mm;
  double* P = Pb + mm;
  double* AP = P + mm;
  double* Pf = AP + mm;   // P_f, then X1 = Gm T
  double* Wb = Pf + mm;   // T P_f, then Sb = T' Gm T
  double* Gm = Wb + mm;
  // p x m / m x p
  double* Zl = Gm + mm;   // p x m
  double* Zb = Zl + pm;   // p x m
  double* M = Zb + pm;    // m x p
  double* K = M + pm;     // m x p
  double* KF = K + pm;    // m x p   K (Z~M + H~)
  double* AM = KF + pm;   // m x p   M - K (Z~ M)
  double* U = AM + pm;    // m x p   Sb K
  double* Kb = U + pm;    // m x p
  double* Mb = Kb + pm;   // m x p
  double* KZ = Mb + pm;   // p x m   (K'U) Z~
  // p x p
  double* Hl = KZ + pm;
  double* Hb = Hl + pp;
  double* F = Hb + pp;
  double* Fi = F + pp;
  double* ZM = Fi + pp;
  double* Fp = ZM + pp;   // Z~M + H~
  double* KtU = Fp + pp;
  double* Fib = KtU + pp;
  double* Fb = Fib + pp;
  double* Tp = Fb + pp;   // scratch (Fi Fib, Cholesky L of the big-p path)
  // vectors
  double* al = Tp + pp;
  double* af = al + m;
  double* ab = af + m;
  double* abf = ab + m;
  double* cl = abf + m;
  double* cb = cl + m;
  double* dl = cb + m;
  double* db = dl + p;
  double* vl = db + p;
  double* vb = vl + p;
  double* ym = vb + p;
  double* ob = ym + p;

  for (int i = lane; i < m; i += KF_WAVE) {
    int base = i * m - (i * (i - 1)) / 2;
    for (int j = i; j < m; ++j) pairtab[base + (j - i)] = (i << 16) | j;
  }
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  {
    const double* gT = A.Tm + b * mm;
    for (int e = lig; e < mm; e += G) {
      Tl[e] = gT[e];
      Tb[e] = 0.0;
      Gs[e] = 0.0;
      Pb[e] = 0.0;
    }
    const double* gZ = A.Z + b * pm;
    for (int e = lig; e < pm; e += G) {
      Zl[e] = gZ[e];
      Zb[e] = 0.0;
    }
    const double* gH = A.H + b * pp;
    for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      Hl[e] = (univariate && q1 != q2) ? 0.0 : gH[e];
      Hb[e] = 0.0;
    }
    for (int e = lig; e < m; e += G) {
      cl[e] = A.c ? A.c[b * m + e] : 0.0;
      cb[e] = 0.0;
      ab[e] = 0.0;
    }
    for (int e = lig; e < p; e += G) {
      dl[e] = A.d ? A.d[b * p + e] : 0.0;
      db[e] = 0.0;
    }
  }
  wsync();

  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L);
  const double* gpa = A.pred_a + b * (long)(n + 1) * m;
  const double* gpP = A.pred_P + b * (long)(n + 1) * mm;
  const double g_all = A.g_logp ? A.g_logp[b] : 0.0;
  // cotangents of the tape outputs (kf_filter_backward_tapes): (ab, Pb) -- the adjoint of the predicted moments of step t + 1 when
  // step t starts -- begins at the cotangent of predicted slot T and collects slot t's at the end of step t; the filtered
  // moments' cotangents join abf = T' ab and Sb = T' Gm T (P_f is symmetric by construction: its cotangent's symmetric part)
  const double* gfa = A.g_filt_a ? A.g_filt_a + b * (long)n * m : nullptr;
  const double* gpa_c = A.g_pred_a ? A.g_pred_a + b * (long)(n + 1) * m : nullptr;
  const double* gfP = A.g_filt_P ? A.g_filt_P + b * (long)n * mm : nullptr;
  const double* gpP_c = A.g_pred_P ? A.g_pred_P + b * (long)(n + 1) * mm : nullptr;
  if (gpa_c) for (int e = lig; e < m; e += G) ab[e] = gpa_c[(long)n * m + e];
  if (gpP_c) for (int e = lig; e < mm; e += G) Pb[e] = gpP_c[(long)n * mm + e];
  if (gpa_c || gpP_c) wsync();
  const unsigned tv = A.flags & KF_FLAG_TV_ANY;
  const bool tv_rq = (tv & (KF_FLAG_TV_R | KF_FLAG_TV_Q)) != 0;
  const int mr = m * r, rr = r * r;
  if (tv_rq && live) {
    // R Q R' differs per step: g_R / g_Q come from each step's Gm.  The one that is NOT time-varying is summed in place in
    // global memory (lane e owns element e at every step)
    if (!(tv & KF_FLAG_TV_R) && A.g_R) for (int e = lig; e < mr; e += G) A.g_R[b * (long)mr + e] = 0.0;
    if (!(tv & KF_FLAG_TV_Q) && A.g_Q) for (int e = lig; e < rr; e += G) A.g_Q[b * (long)rr + e] = 0.0;
  }

  for (int t = n - 1; t >= 0; --t) {
    const double g = g_all + (A.g_ll_obs ? A.g_ll_obs[b * (long)n + t] : 0.0);
    if (tv) {  // slice t of the flagged matrices
      const long bt = b * (long)n + t;
      if (tv & KF_FLAG_TV_T) for (int e = lig; e < mm; e += G) Tl[e] = A.Tm[bt * mm + e];
      if (tv & KF_FLAG_TV_Z) for (int e = lig; e < pm; e += G) Zl[e] = A.Z[bt * pm + e];
      if (tv & KF_FLAG_TV_H) for (int e = lig; e < pp; e += G) {
        int q1 = e / p, q2 = e - q1 * p;
        Hl[e] = (univariate && q1 != q2) ? 0.0 : A.H[bt * pp + e];
      }
      if ((tv & KF_FLAG_TV_D) && A.d) for (int e = lig; e < p; e += G) dl[e] = A.d[bt * p + e];
    }
    // ---- (1) tape + observation ---------------------------------------------------------------
    for (int e = lig; e < m; e += G) al[e] = gpa[(long)t * m + e];
    for (int e = lig; e < mm; e += G) P[e] = gpP[(long)t * mm + e];
    for (int q = lig; q < p; q += G) {
      double yv = ybase[(long)t * p + q];
      bool miss = yv != yv;
      ym[q] = miss ? 0.0 : yv;
      ob[q] = miss ? 0.0 : 1.0;
    }
    wsync();
    // ---- (2) v, M = P Z~' ------------------------------------------------------------------------
    for (int q = lig; q < p; q += G) {
      double s = ym[q] + dsign * dl[q];
      for (int j = 0; j < m; ++j) s -= Zl[q * m + j] * al[j];
      vl[q] = ob[q] != 0.0 ? s : 0.0;
    }
    for (int e = lig; e < pm; e += G) {
      int j = e / p, q = e - j * p;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += P[j * m + k] * Zl[q * m + k];
      M[e] = ob[q] != 0.0 ? s : 0.0;
    }
    wsync();
    // ---- (3) ZM = Z~ M ; Fp = ZM + H~ ; F = Fp + D_miss -----------------------------------------------
    for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      double zm = 0.0, fp = 0.0, f;
      if (ob[q1] != 0.0 && ob[q2] != 0.0) {
        for (int j = 0; j < m; ++j) zm += Zl[q1 * m + j] * M[j * p + q2];
        fp = zm + Hl[e];
        f = fp;
      } else {
        f = (q1 == q2) ? 1.0 : 0.0;
      }
      ZM[e] = zm;
      Fp[e] = fp;
      F[e] = f;
    }
    wsync();
    // ---- (4) Fi = F^-1 ------------------------------------------------------------------------------
#if KF_PMAX == 1
    inverse_small<1, G>(F, Fi, lig);
#elif KF_PMAX == 2
    inverse_small<2, G>(F, Fi, lig);
#elif KF_PMAX == 4
    if (p == 3) inverse_small<3, G>(F, Fi, lig);
    else inverse_small<4, G>(F, Fi, lig);
#elif KF_PMAX == 8
    switch (p) {
      case 5: inverse_small<5, G>(F, Fi, lig); break;
      case 6: inverse_small<6, G>(F, Fi, lig); break;
      case 7: inverse_small<7, G>(F, Fi, lig); break;
      default: inverse_small<8, G>(F, Fi, lig); break;
    }
#else
    inverse_big<G>(F, Tp, Fi, p, lig);
#endif
    wsync();
    // ---- (5) K = M Fi ---------------------------------------------------------------------------------
    for (int e = lig; e < pm; e += G) {
      int j = e / p, q = e - j * p;
      double s = 0.0;
      for (int q2 = 0; q2 < p; ++q2) s += M[j * p + q2] * Fi[q2 * p + q];
      K[e] = s;
    }
    wsync();
    // ---- (6) a_f, AP = P - K M', KF = K Fp, AM = M - K ZM ------------------------------------------------
    for (int j = lig; j < m; j += G) {
      double s = al[j];
      for (int q = 0; q < p; ++q) s += K[j * p + q] * vl[q];
      af[j] = s;
    }
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = P[e];
      for (int q = 0; q < p; ++q) s -= K[i * p + q] * M[j * p + q];
      AP[e] = s;
    }
    for (int e = lig; e < pm; e += G) {
      int j = e / p, q = e - j * p;
      double s1 = 0.0, s2 = M[e];
      for (int q2 = 0; q2 < p; ++q2) {
        double kk = K[j * p + q2];
        s1 += kk * Fp[q2 * p + q];
        s2 -= kk * ZM[q2 * p + q];
      }
      KF[e] = s1;
      AM[e] = s2;
    }
    wsync();
    // ---- (7) P_f (expanded Joseph, as the forward kernel) ----------------------------------------------------
    for (int e = lig; e < npair; e += G) {
      int ij = pairtab[e];
      int i = ij >> 16, j = ij & 0xffff;
      double s = P[i * m + j];
      for (int q = 0; q < p; ++q) {
        double kj = K[j * p + q];
        s -= K[i * p + q] * M[j * p + q];
        s -= M[i * p + q] * kj;
        s += KF[i * p + q] * kj;
      }
      Pf[i * m + j] = s;
      Pf[j * m + i] = s;
    }
    wsync();
    // ---- (8) W = T P_f ; Gm = sym(Pb) ; abf = T' ab ; cb += ab -------------------------------------------------
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[i * m + k] * Pf[k * m + j];
      Wb[e] = s;
      Gm[e] = 0.5 * (Pb[e] + Pb[j * m + i]);
    }
    for (int i = lig; i < m; i += G) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[k * m + i] * ab[k];
      abf[i] = gfa ? s + gfa[(long)t * m + i] : s;
      cb[i] += ab[i];
    }
    wsync();
    // ---- (9) Tb += ab a_f' + 2 Gm W ; Gs += Gm ; X1 = Gm T (into Pf) -----------------------------------------------
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < m; ++k) {
        double gik = Gm[i * m + k];
        s1 += gik * Wb[k * m + j];
        s2 += gik * Tl[k * m + j];
      }
      Tb[e] += ab[i] * af[j] + 2.0 * s1;
      Gs[e] += Gm[e];
      Pf[e] = s2;
    }
    wsync();
    // ---- (10) Sb = T' X1 (into Wb) ---------------------------------------------------------------------------------------
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[k * m + i] * Pf[k * m + j];
      if (gfP) s += 0.5 * (gfP[(long)t * mm + e] + gfP[(long)t * mm + j * m + i]);
      Wb[e] = s;
    }
    wsync();
    // ---- (11) U = Sb K ---------------------------------------------------------------------------------------------------
    for (int e = lig; e < pm; e += G) {
      int i = e / p, q = e - i * p;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Wb[i * m + k] * K[k * p + q];
      U[e] = s;
    }
    wsync();
    // ---- (12) KtU = K' U ; vb ; Kb --------------------------------------------------------------------------------------------
    for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      double s = 0.0;
      for (int i = 0; i < m; ++i) s += K[i * p + q1] * U[i * p + q2];
      KtU[e] = s;
    }
    for (int q = lig; q < p; q += G) {
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += K[j * p + q] * abf[j];
      double fv = 0.0;
      for (int q2 = 0; q2 < p; ++q2) fv += Fi[q * p + q2] * vl[q2];
      vb[q] = s - g * fv;
    }
    for (int e = lig; e < pm; e += G) {
      int i = e / p, q = e - i * p;
      double s = abf[i] * vl[q];
      for (int q2 = 0; q2 < p; ++q2) {
        double hm1 = (ob[q2] != 0.0 && ob[q] != 0.0) ? (Hl[q2 * p + q] + Hl[q * p + q2]) : 0.0;
        s += U[i * p + q2] * hm1;
      }
      double s3 = 0.0;
      for (int k = 0; k < m; ++k) s3 += Wb[i * m + k] * AM[k * p + q];
      Kb[e] = s - 2.0 * s3;
    }
    wsync();
    // ---- (13) Fib = -g/2 v v' + M' Kb ; KZ = KtU Z~ ----------------------------------------------------------------------------------
    for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      double s = -0.5 * g * vl[q1] * vl[q2];
      for (int j = 0; j < m; ++j) s += M[j * p + q1] * Kb[j * p + q2];
      Fib[e] = s;
    }
    for (int e = lig; e < pm; e += G) {
      int q1 = e / m, j = e - q1 * m;
      double s = 0.0;
      for (int q2 = 0; q2 < p; ++q2)
        if (ob[q2] != 0.0) s += KtU[q1 * p + q2] * Zl[q2 * m + j];
      KZ[e] = s;
    }
    wsync();
    // ---- (14) Fb = -g/2 Fi - Fi Fib Fi ---------------------------------------------------------------------------------------------------
    for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      double s = 0.0;
      for (int k = 0; k < p; ++k) s += Fi[q1 * p + k] * Fib[k * p + q2];
      Tp[e] = s;
    }
    wsync();
    for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      double s = -0.5 * g * Fi[e];
      for (int k = 0; k < p; ++k) s -= Tp[q1 * p + k] * Fi[k * p + q2];
      Fb[e] = s;
    }
    wsync();
    // ---- (15) Mb = Kb Fi + Z~' Fb -----------------------------------------------------------------------------------------------------------
    for (int e = lig; e < pm; e += G) {
      int j = e / p, q = e - j * p;
      double s = 0.0;
      for (int q2 = 0; q2 < p; ++q2) {
        s += Kb[j * p + q2] * Fi[q2 * p + q];
        if (ob[q2] != 0.0) s += Zl[q2 * m + j] * Fb[q2 * p + q];
      }
      Mb[e] = s;
    }
    wsync();
    // ---- (16) new Pb, ab ; accumulate Zb, Hb, db -------------------------------------------------------------------------------------------------
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = Wb[e];
      for (int q = 0; q < p; ++q) {
        if (ob[q] == 0.0) continue;
        double zqi = Zl[q * m + i], zqj = Zl[q * m + j];
        s -= zqi * U[j * p + q];
        s -= U[i * p + q] * zqj;
        s += zqi * KZ[q * m + j];
        s += Mb[i * p + q] * zqj;
      }
      Pb[e] = gpP_c ? s + gpP_c[(long)t * mm + e] : s;
    }
    for (int e = lig; e < pm; e += G) {
      int q = e / m, j = e - q * m;
      if (ob[q] == 0.0) continue;
      double s = -vb[q] * al[j];
      for (int i = 0; i < m; ++i) {
        s -= 2.0 * U[i * p + q] * AP[i * m + j];
        s += Mb[i * p + q] * P[i * m + j];
      }
      for (int q2 = 0; q2 < p; ++q2) s += Fb[q * p + q2] * M[j * p + q2];
      Zb[e] += s;
    }
    for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      if (ob[q1] != 0.0 && ob[q2] != 0.0) Hb[e] += KtU[e] + Fb[e];
    }
    for (int j = lig; j < m; j += G) {
      double s = abf[j];
      for (int q = 0; q < p; ++q)
        if (ob[q] != 0.0) s -= Zl[q * m + j] * vb[q];
      ab[j] = gpa_c ? s + gpa_c[(long)t * m + j] : s;
    }
    for (int q = lig; q < p; q += G)
      if (ob[q] != 0.0) db[q] += dsign * vb[q];
    wsync();
    if (tv) {  // gradients of the flagged matrices: this step's terms go to slice t, the accumulators restart
      const long bt = b * (long)n + t;
      if (tv_rq) {
        const double* gR = A.R + ((tv & KF_FLAG_TV_R) ? bt : b) * (long)mr;
        const double* gQ = A.Q + ((tv & KF_FLAG_TV_Q) ? bt : b) * (long)rr;
        double* GR = AP;  // AP .. Gm (4 m^2 doubles) are dead until the next step
        for (int e = lig; e < mr; e += G) {
          int i = e / r, l = e - i * r;
          double s = 0.0;
          for (int j = 0; j < m; ++j) s += Gs[i * m + j] * gR[j * r + l];
          GR[e] = s;
        }
        wsync();
        if (live && A.g_R) {
          double* o = A.g_R + ((tv & KF_FLAG_TV_R) ? bt : b) * (long)mr;
          for (int e = lig; e < mr; e += G) {
            int i = e / r, l = e - i * r;
            double s = 0.0;
            for (int k = 0; k < r; ++k) s += GR[i * r + k] * (gQ[k * r + l] + gQ[l * r + k]);
            o[e] = (tv & KF_FLAG_TV_R) ? s : o[e] + s;
          }
        }
        if (live && A.g_Q) {
          double* o = A.g_Q + ((tv & KF_FLAG_TV_Q) ? bt : b) * (long)rr;
          for (int e = lig; e < rr; e += G) {
            int k = e / r, l = e - k * r;
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += gR[i * r + k] * GR[i * r + l];
            o[e] = (tv & KF_FLAG_TV_Q) ? s : o[e] + s;
          }
        }
        for (int e = lig; e < mm; e += G) Gs[e] = 0.0;
      }
      if (tv & KF_FLAG_TV_T) for (int e = lig; e < mm; e += G) {
        if (live && A.g_T) A.g_T[bt * mm + e] = Tb[e];
        Tb[e] = 0.0;
      }
      if (tv & KF_FLAG_TV_Z) for (int e = lig; e < pm; e += G) {
        if (live && A.g_Z) A.g_Z[bt * pm + e] = Zb[e];
        Zb[e] = 0.0;
      }
      if (tv & KF_FLAG_TV_H) for (int e = lig; e < pp; e += G) {
        int q1 = e / p, q2 = e - q1 * p;
        if (live && A.g_H) A.g_H[bt * pp + e] = (univariate && q1 != q2) ? 0.0 : Hb[e];
        Hb[e] = 0.0;
      }
      if (tv & KF_FLAG_TV_C) for (int e = lig; e < m; e += G) {
        if (live && A.g_c) A.g_c[bt * m + e] = cb[e];
        cb[e] = 0.0;
      }
      if (tv & KF_FLAG_TV_D) for (int e = lig; e < p; e += G) {
        if (live && A.g_d) A.g_d[bt * p + e] = db[e];
        db[e] = 0.0;
      }
      wsync();
    }
  }

  // ---- gradients out ------------------------------------------------------------------------------------
  // R_b = Gs R (Q + Q') ; Q_b = R' Gs R   (R Q R' is static: one accumulated Gs serves all steps)
  const double* gR = A.R + b * (long)m * r;
  const double* gQ = A.Q + b * (long)r * r;
  double* GR = AP;  // m x r scratch (r <= m is not required: AP..Gm are contiguous 4 m^2)
  if (!tv_rq && (long)m * r <= 4L * mm) {
    for (int e = lig; e < m * r; e += G) {
      int i = e / r, l = e - i * r;
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += Gs[i * m + j] * gR[j * r + l];
      GR[e] = s;
    }
  }
  wsync();
  if (live) {
    if (A.g_a0) for (int e = lig; e < m; e += G) A.g_a0[b * m + e] = ab[e];
    if (A.g_P0) for (int e = lig; e < mm; e += G) A.g_P0[b * mm + e] = Pb[e];
    if (A.g_T && !(tv & KF_FLAG_TV_T)) for (int e = lig; e < mm; e += G) A.g_T[b * mm + e] = Tb[e];
    if (A.g_Z && !(tv & KF_FLAG_TV_Z)) for (int e = lig; e < pm; e += G) A.g_Z[b * pm + e] = Zb[e];
    if (A.g_H && !(tv & KF_FLAG_TV_H)) for (int e = lig; e < pp; e += G) {
      int q1 = e / p, q2 = e - q1 * p;
      A.g_H[b * pp + e] = (univariate && q1 != q2) ? 0.0 : Hb[e];
    }
    if (A.g_c && !(tv & KF_FLAG_TV_C)) for (int e = lig; e < m; e += G) A.g_c[b * m + e] = cb[e];
    if (A.g_d && !(tv & KF_FLAG_TV_D)) for (int e = lig; e < p; e += G) A.g_d[b * p + e] = db[e];
    if (A.g_R && !tv_rq) for (int e = lig; e < m * r; e += G) {
      int i = e / r, l = e - i * r;
      double s = 0.0;
      for (int k = 0; k < r; ++k) s += GR[i * r + k] * (gQ[k * r + l] + gQ[l * r + k]);
      A.g_R[b * (long)m * r + e] = s;
    }
    if (A.g_Q && !tv_rq) for (int e = lig; e < r * r; e += G) {
      int k = e / r, l = e - k * r;
      double s = 0.0;
      for (int i = 0; i < m; ++i) s += gR[i * r + k] * GR[i * r + l];
      A.g_Q[b * (long)r * r + e] = s;
    }
  }
}

template <int G>
int launch_g(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  const int dpb = KF_WAVE / G;
  size_t lds = sizeof(double) *
               ((size_t)kf_pair_table_doubles(a.m) + (size_t)dpb * bwd_region_doubles(a.m, a.p));
  if (lds > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "backward(lds): m=%d p=%d needs %zu B of LDS per workgroup (> %d)", a.m, a.p, lds,
               KF_LDS_BYTES_MAX);
    return KF_ERR_UNSUPPORTED;
  }
  long blocks = (a.B + dpb - 1) / dpb;
#define KF_BWD_LAUNCH(MT_)                                                                          \
  do {                                                                                              \
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_bwd_lds_kernel<G, MT_>),   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL((kf_bwd_lds_kernel<G, MT_>), dim3((unsigned)blocks), dim3(KF_WAVE), lds,     \
                       stream, a);                                                                  \
  } while (0)
#if KF_PMAX == 1
  if (a.m == 10 && G == 64) KF_BWD_LAUNCH(10);
  else KF_BWD_LAUNCH(0);
#else
  KF_BWD_LAUNCH(0);
#endif
#undef KF_BWD_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

#define KF_CAT2(a, b) a##b
#define KF_CAT(a, b) KF_CAT2(a, b)
int KF_CAT(kf_launch_backward_lds_p, KF_PMAX)(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  if ((long)a.m * a.r > 4L * a.m * a.m) {
    KF_SET_ERR(h, "backward(lds): r=%d > 4 m=%d is not supported", a.r, a.m);
    return KF_ERR_UNSUPPORTED;
  }
  int G = kf_pick_lanes(h, a.m, a.B);
  while (G < KF_WAVE) {
    size_t lds = sizeof(double) * ((size_t)kf_pair_table_doubles(a.m) +
                                   (size_t)(KF_WAVE / G) * bwd_region_doubles(a.m, a.p));
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
