// kf_smoother_lds.hip -- fused RTS (fixed-interval) smoother, state resident in LDS.
//
// Replaces KalmanSmoother.build_graph / smoother_step (pymc_statespace/filters/
// kalman_smoother.py:56-104) for a batch of draws: one kernel runs the whole backward time
// loop; a group of G lanes owns one draw, streams that draw's filtered tapes (a_f[t], P_f[t])
// from HBM, keeps (a_s, P_s, T, RQR') in LDS and writes the smoothed tapes.
//
// Per step (reference order, :87-104):
//    a_hat = T a_f ;  P_hat = T P_f T' + R Q R'          (no intercept, no symmetrisation)
//    G     = (pinv(P_hat) T P_f)'
//    a_s   = a_f + G (a_s' - a_hat) ;  P_s = P_f + G (P_s' - P_hat) G'
// pinv: P_hat is symmetric PSD.  When its Cholesky pivots stay above 1e-12 * max diag the
// pseudo-inverse IS the inverse and X = pinv(P_hat) (T P_f) is obtained by two triangular
// solves (one lane per right-hand side, no lane exchange).  Otherwise -- numerically singular
// P_hat, the case the reference's `pinv` is there for -- a parallel-ordered (round-robin)
// Jacobi eigen-decomposition in LDS builds the Moore-Penrose inverse with numpy's cutoff
// (eigenvalues <= 1e-15 * max |eigenvalue| are dropped; numpy.linalg.pinv default rcond).
#include "kf_internal.h"

namespace {

__device__ __forceinline__ void wsync() { __syncthreads(); }

template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
  return v;
}

__host__ __device__ inline int smooth_region_doubles(int m) {
  int M2 = (m + 1) / 2;
  return (9 * m * m + 6 * m + 4 * M2 + 2) | 1;
}

// Moore-Penrose inverse of the symmetric matrix held in D (destroyed) -> Pi.
template <int G>
__device__ void pinv_jacobi(double* D, double* V, double* Pi, double* ev, double* cs, int m, int lig) {
  const int mm = m * m;
  const int M = (m + 1) & ~1;  // players of the round-robin tournament (one dummy if m is odd)
  const int half = M / 2;
  for (int e = lig; e < mm; e += G) V[e] = (e / m == e % m) ? 1.0 : 0.0;
  wsync();
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (int e = lig; e < mm; e += G) {
      double x = D[e];
      if (e / m == e % m) dg += x * x; else off += x * x;
    }
    off = group_sum<G>(off);
    dg = group_sum<G>(dg);
    const bool busy = off > 1e-30 * dg && off > 0.0;
    if (!__any(busy)) break;  // wave-uniform exit: converged groups keep doing identity rotations
    for (int rd = 0; rd < M - 1; ++rd) {
      // (a) one lane per pair: rotation angle
      for (int pr = lig; pr < half; pr += G) {
        int p_, q_;
        if (pr == 0) { p_ = M - 1; q_ = rd; }
        else { p_ = (rd + pr) % (M - 1); q_ = (rd - pr + (M - 1)) % (M - 1); }
        if (p_ > q_) { int t_ = p_; p_ = q_; q_ = t_; }
        double c = 1.0, s = 0.0;
        if (q_ < m) {
          double apq = D[p_ * m + q_];
          if (apq != 0.0) {
            double app = D[p_ * m + p_], aqq = D[q_ * m + q_];
            double tau = (aqq - app) / (2.0 * apq);
            double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            s = t * c;
          }
        } else {
          p_ = q_ = -1;
        }
        cs[4 * pr + 0] = c;
        cs[4 * pr + 1] = s;
        cs[4 * pr + 2] = (double)p_;
        cs[4 * pr + 3] = (double)q_;
      }
      wsync();
      // (b) columns of D and V
      for (int idx = lig; idx < half * m; idx += G) {
        int pr = idx / m, k = idx - pr * m;
        int p_ = (int)cs[4 * pr + 2], q_ = (int)cs[4 * pr + 3];
        if (p_ < 0) continue;
        double c = cs[4 * pr + 0], s = cs[4 * pr + 1];
        double dkp = D[k * m + p_], dkq = D[k * m + q_];
        D[k * m + p_] = c * dkp - s * dkq;
        D[k * m + q_] = s * dkp + c * dkq;
        double vkp = V[k * m + p_], vkq = V[k * m + q_];
        V[k * m + p_] = c * vkp - s * vkq;
        V[k * m + q_] = s * vkp + c * vkq;
      }
      wsync();
      // (c) rows of D
      for (int idx = lig; idx < half * m; idx += G) {
        int pr = idx / m, k = idx - pr * m;
        int p_ = (int)cs[4 * pr + 2], q_ = (int)cs[4 * pr + 3];
        if (p_ < 0) continue;
        double c = cs[4 * pr + 0], s = cs[4 * pr + 1];
        double dpk = D[p_ * m + k], dqk = D[q_ * m + k];
        D[p_ * m + k] = c * dpk - s * dqk;
        D[q_ * m + k] = s * dpk + c * dqk;
      }
      wsync();
    }
  }
  double smax = 0.0;
  for (int k = 0; k < m; ++k) smax = fmax(smax, fabs(D[k * m + k]));
  for (int k = lig; k < m; k += G) {
    double lam = D[k * m + k];
    ev[k] = (fabs(lam) > 1e-15 * smax) ? 1.0 / lam : 0.0;
  }
  wsync();
  for (int e = lig; e < mm; e += G) {
    int i = e / m, j = e - i * m;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += V[i * m + k] * V[j * m + k] * ev[k];
    Pi[e] = s;
  }
  wsync();
}

// MT > 0: compile-time k_states -- the triangular solves then run on a register-resident
// right-hand side (the run-time-m path round-trips every element through LDS: 2 m^2 dependent
// LDS accesses per lane and step, which dominated the kernel at m = 10).
template <int G, int MT>
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_lds_kernel(const KfSmoothArgs A) {
  extern __shared__ double smem[];
  constexpr int DPB = KF_WAVE / G;
  const int lane = threadIdx.x;
  const int lig = lane % G;
  const int grp = lane / G;
  const int m = MT > 0 ? MT : A.m;
  const int r = A.r, n = A.T;
  const int mm = m * m;

  long b = (long)blockIdx.x * DPB + grp;
  const bool live = b < A.B;
  if (!live) b = A.B - 1;
  // fallback pass behind kf_smooth_mfma_kernel (launched with G = 64: one draw per wavefront)
  if (A.only_flagged && DPB == 1 && A.only_flagged[b] == 0) return;

  double* S = smem + (size_t)grp * smooth_region_doubles(m);
  double* Ps = S;
  double* P = Ps + mm;
  double* W = P + mm;    // T P_f, then X = pinv(P_hat) T P_f
  double* Ph = W + mm;   // P_hat, then P_s' - P_hat
  double* L = Ph + mm;   // Cholesky factor / Jacobi work matrix / E = G (P_s' - P_hat)
  double* V = L + mm;    // Jacobi eigenvectors
  double* Xb = V + mm;   // pinv(P_hat) on the Jacobi path
  double* Tl = Xb + mm;
  double* RQR = Tl + mm;
  double* al = RQR + mm;  // a_f[t]
  double* as = al + m;    // smoothed state
  double* ah = as + m;
  double* dx = ah + m;
  double* Ld = dx + m;    // Cholesky diagonal / inverse eigenvalues
  double* tmpv = Ld + m;
  double* cs = tmpv + m;  // 4 * ceil(m/2) Jacobi rotations

  const double* gfa = A.filt_a + b * (long)n * m;
  const double* gfP = A.filt_P + b * (long)n * mm;
  double* gsa = A.smooth_a + b * (long)n * m;
  double* gsP = A.smooth_P + b * (long)n * mm;

  // time-varying matrices (KF_FLAG_TV_*): [B][n][.][.]; the step that smooths t predicts from t with slice t, the
  // pairing of the forward filter (kalman_filter.py:231-252: step t predicts with T[t], R[t], Q[t])
  // (KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING: slice t + 1 instead, what the reference's scan hands its step, kalman_smoother.py:62-77)
  const bool tvT = (A.flags & KF_FLAG_TV_T) != 0, tvR = (A.flags & KF_FLAG_TV_R) != 0, tvQ = (A.flags & KF_FLAG_TV_Q) != 0;
  const int tsh = (A.flags & KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING) ? 1 : 0;
  auto load_T = [&](int t) {
    const double* gT = A.Tm + (tvT ? b * (long)n + t : b) * mm;
    for (int e = lig; e < mm; e += G) Tl[e] = gT[e];
  };
  auto load_RQR = [&](int t) {
    const double* gR = A.R + (tvR ? b * (long)n + t : b) * (long)m * r;
    const double* gQ = A.Q + (tvQ ? b * (long)n + t : b) * (long)r * r;
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
  };
  {
    for (int e = lig; e < mm; e += G) Ps[e] = gfP[(long)(n - 1) * mm + e];
    for (int e = lig; e < m; e += G) as[e] = gfa[(long)(n - 1) * m + e];
    if (!tvT) load_T(0);
    if (!(tvR || tvQ)) load_RQR(0);
  }
  wsync();
  if (live) {  // smoothed[-1] = filtered[-1]  (:79-85)
    for (int e = lig; e < m; e += G) st_stream(gsa + (long)(n - 1) * m + e, as[e]);
    for (int e = lig; e < mm; e += G) st_stream(gsP + (long)(n - 1) * mm + e, Ps[e]);
  }

  for (int t = n - 2; t >= 0; --t) {
    for (int e = lig; e < m; e += G) al[e] = gfa[(long)t * m + e];
    for (int e = lig; e < mm; e += G) P[e] = gfP[(long)t * mm + e];
    if (tvT) load_T(t + tsh);
    if (tvR || tvQ) load_RQR(t + tsh);
    wsync();
    // a_hat = T a ; W = T P
    for (int i = lig; i < m; i += G) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[i * m + k] * al[k];
      ah[i] = s;
    }
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[i * m + k] * P[k * m + j];
      W[e] = s;
    }
    wsync();
    // P_hat = W T' + RQR'
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = RQR[e];
      for (int k = 0; k < m; ++k) s += W[i * m + k] * Tl[j * m + k];
      Ph[e] = s;
      L[e] = s;
    }
    wsync();
    // ---- Cholesky (left-looking; strict lower in L, diagonal in Ld) ------------------------
    double maxdiag = 0.0;
    for (int i = 0; i < m; ++i) maxdiag = fmax(maxdiag, Ph[i * m + i]);
    bool ok = maxdiag > 0.0;
    for (int j = 0; j < m; ++j) {
      double dj = L[j * m + j];
      for (int k = 0; k < j; ++k) dj -= L[j * m + k] * L[j * m + k];
      if (!(dj > 1e-12 * maxdiag)) ok = false;
      double sj = sqrt(dj), inv = 1.0 / sj;
      for (int i = j + 1 + lig; i < m; i += G) {
        double s = L[i * m + j];
        for (int k = 0; k < j; ++k) s -= L[i * m + k] * L[j * m + k];
        L[i * m + j] = s * inv;
      }
      if (lig == 0) Ld[j] = sj;
      wsync();
    }
    if (!__any(!ok)) {
      // X = P_hat^-1 W : one right-hand side (column of W) per lane, in place
      if (MT > 0) {
        constexpr int MS = MT > 0 ? MT : 1;
        for (int c = lig; c < MS; c += G) {
          double x[MS];
#pragma unroll
          for (int i = 0; i < MS; ++i) x[i] = W[i * MS + c];
#pragma unroll
          for (int i = 0; i < MS; ++i) {
            double s = x[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s -= L[i * MS + k] * x[k];
            x[i] = s / Ld[i];
          }
#pragma unroll
          for (int i = MS - 1; i >= 0; --i) {
            double s = x[i];
#pragma unroll
            for (int k = i + 1; k < MS; ++k) s -= L[k * MS + i] * x[k];
            x[i] = s / Ld[i];
          }
#pragma unroll
          for (int i = 0; i < MS; ++i) W[i * MS + c] = x[i];
        }
      } else {
        for (int c = lig; c < m; c += G) {
          for (int i = 0; i < m; ++i) {
            double s = W[i * m + c];
            for (int k = 0; k < i; ++k) s -= L[i * m + k] * W[k * m + c];
            W[i * m + c] = s / Ld[i];
          }
          for (int i = m - 1; i >= 0; --i) {
            double s = W[i * m + c];
            for (int k = i + 1; k < m; ++k) s -= L[k * m + i] * W[k * m + c];
            W[i * m + c] = s / Ld[i];
          }
        }
      }
    } else {
      for (int e = lig; e < mm; e += G) L[e] = Ph[e];
      wsync();
      pinv_jacobi<G>(L, V, Xb, Ld, cs, m, lig);
      for (int e = lig; e < mm; e += G) {  // X = pinv W  (into V, then back to W)
        int i = e / m, j = e - i * m;
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += Xb[i * m + k] * W[k * m + j];
        V[e] = s;
      }
      wsync();
      for (int e = lig; e < mm; e += G) W[e] = V[e];
    }
    wsync();
    // dx = a_s' - a_hat ; D = P_s' - P_hat (in place of P_hat)
    for (int k = lig; k < m; k += G) dx[k] = as[k] - ah[k];
    for (int e = lig; e < mm; e += G) Ph[e] = Ps[e] - Ph[e];
    wsync();
    // a_s = a + G dx  with G[i][k] = X[k][i] ;  E = G D
    for (int i = lig; i < m; i += G) {
      double s = al[i];
      for (int k = 0; k < m; ++k) s += W[k * m + i] * dx[k];
      as[i] = s;
    }
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += W[k * m + i] * Ph[k * m + j];
      L[e] = s;
    }
    wsync();
    // P_s = P + E G'
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = P[e];
      for (int k = 0; k < m; ++k) s += L[i * m + k] * W[k * m + j];
      Ps[e] = s;
    }
    wsync();
    if (live) {
      for (int e = lig; e < m; e += G) st_stream(gsa + (long)t * m + e, as[e]);
      for (int e = lig; e < mm; e += G) st_stream(gsP + (long)t * mm + e, Ps[e]);
    }
  }
}

template <int G>
int launch_g(kf_handle* h, const KfSmoothArgs& a, hipStream_t stream) {
  const int dpb = KF_WAVE / G;
  size_t lds = sizeof(double) * (size_t)dpb * smooth_region_doubles(a.m);
  if (lds > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "smoother(lds): m=%d needs %zu B of LDS per workgroup (> %d)", a.m, lds,
               KF_LDS_BYTES_MAX);
    return KF_ERR_UNSUPPORTED;
  }
  long blocks = (a.B + dpb - 1) / dpb;
#define KF_SM_LAUNCH(MT_)                                                                            \
  do {                                                                                               \
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_smooth_lds_kernel<G, MT_>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
    hipLaunchKernelGGL((kf_smooth_lds_kernel<G, MT_>), dim3((unsigned)blocks), dim3(KF_WAVE), lds,   \
                       stream, a);                                                                   \
  } while (0)
  switch (a.m) {
    case 2: KF_SM_LAUNCH(2); break;
    case 4: KF_SM_LAUNCH(4); break;
    case 6: KF_SM_LAUNCH(6); break;
    case 8: KF_SM_LAUNCH(8); break;
    case 10: KF_SM_LAUNCH(10); break;
    case 15: KF_SM_LAUNCH(15); break;
    default: KF_SM_LAUNCH(0); break;
  }
#undef KF_SM_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

// Lean pinv smoother for the draws the workgroup kernel (kf_smoother_mfma_wg.hip, 47 < m <= 64)
// flagged as not safely positive definite: one wavefront per flagged draw, every step with the
// Moore-Penrose inverse as the reference does (kalman_smoother.py:92).  Four m x m regions in LDS
// (A: P_hat -> W -> E;  V: Jacobi vectors -> X;  Pi: pinv -> D;  RQR static), everything else
// streams from the tapes; throughput is irrelevant here (rare draws), ~5 m^3 LDS-fed MACs per step.
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_pinv_wide_kernel(const KfSmoothArgs A, const int32_t* flags) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const long b = blockIdx.x;
  if (!flags[b]) return;
  const int lane = threadIdx.x;
  const int n = A.T, m = A.m, r_ = A.r, mm = m * m;
  double* Am = smem;          // P_hat, then W = T P_f, then E
  double* V = Am + mm;        // Jacobi eigenvectors, then X = pinv(P_hat) W
  double* Pi = V + mm;        // pinv(P_hat), then D = P_s[t+1] - P_hat
  double* RQR = Pi + mm;      // R Q R'
  double* ev = RQR + mm;      // m
  double* cs = ev + m;        // 4 ceil(m / 2)
  double* va = cs + 4 * ((m + 1) / 2);  // a_f (m), diff (m)
  const double* gT = A.Tm + b * (long)mm;
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  const double* gfa = A.filt_a + b * (long)n * m;
  const double* gfP = A.filt_P + b * (long)n * mm;
  double* gsa = A.smooth_a + b * (long)n * m;
  double* gsP = A.smooth_P + b * (long)n * mm;
  // R Q R' (R Q, m x r with r <= 2 m, staged across Am | V | Pi)
  for (int e = lane; e < m * r_; e += KF_WAVE) {
    const int i = e / r_, l = e - i * r_;
    double v = 0.0;
    for (int k = 0; k < r_; ++k) v += gR[i * r_ + k] * gQ[k * r_ + l];
    Am[e] = v;
  }
  wsync();
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    double v = 0.0;
    for (int l = 0; l < r_; ++l) v += Am[i * r_ + l] * gR[j * r_ + l];
    RQR[e] = v;
  }
  for (int e = lane; e < mm; e += KF_WAVE) gsP[(long)(n - 1) * mm + e] = gfP[(long)(n - 1) * mm + e];
  for (int e = lane; e < m; e += KF_WAVE) gsa[(long)(n - 1) * m + e] = gfa[(long)(n - 1) * m + e];
  wsync();
  for (int t = n - 2; t >= 0; --t) {
    const double* Pf = gfP + (long)t * mm;
    const double* Psn = gsP + (long)(t + 1) * mm;
    __threadfence_block();
    // W = T P_f (in V for now) ; P_hat = W T' + RQR' (in Am)
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(gT[i * m + k], Pf[k * m + j], v);
      V[e] = v;
    }
    wsync();
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double v = RQR[e];
      for (int k = 0; k < m; ++k) v = fma(V[i * m + k], gT[j * m + k], v);
      Am[e] = v;
    }
    wsync();
    // the Jacobi iteration wants an exactly symmetric matrix
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      if (i < j) {
        const double v = 0.5 * (Am[i * m + j] + Am[j * m + i]);
        Am[i * m + j] = v;
        Am[j * m + i] = v;
      }
    }
    wsync();
    pinv_jacobi<KF_WAVE>(Am, V, Pi, ev, cs, m, lane);  // Pi = pinv(P_hat); Am, V destroyed
    // W again (Am) ; X = Pi W (V)
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(gT[i * m + k], Pf[k * m + j], v);
      Am[e] = v;
    }
    wsync();
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(Pi[i * m + k], Am[k * m + j], v);
      V[e] = v;
    }
    wsync();
    // D = P_s[t+1] - P_hat (Pi), P_hat = W T' + RQR' recomputed from W
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double v = RQR[e];
      for (int k = 0; k < m; ++k) v = fma(Am[i * m + k], gT[j * m + k], v);
      Pi[e] = Psn[e] - v;
    }
    // a_f, diff = a_s[t+1] - T a_f
    for (int i = lane; i < m; i += KF_WAVE) va[i] = gfa[(long)t * m + i];
    wsync();
    for (int i = lane; i < m; i += KF_WAVE) {
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(gT[i * m + k], va[k], v);
      va[m + i] = gsa[(long)(t + 1) * m + i] - v;
    }
    // E = D X (Am)
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(Pi[i * m + k], V[k * m + j], v);
      Am[e] = v;
    }
    wsync();
    // P_s[t] = P_f + X' E ; a_s[t] = a_f + X' diff
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double v = Pf[e];
      for (int k = 0; k < m; ++k) v = fma(V[k * m + i], Am[k * m + j], v);
      gsP[(long)t * mm + e] = v;
    }
    for (int i = lane; i < m; i += KF_WAVE) {
      double v = va[i];
      for (int k = 0; k < m; ++k) v = fma(V[k * m + i], va[m + k], v);
      gsa[(long)t * m + i] = v;
    }
    wsync();
  }
}

}  // namespace

// fallback of the workgroup smoother: draws with a non-zero flag, Jacobi pinv at every step
int kf_launch_smoother_pinv_wide(kf_handle* h, const KfSmoothArgs& a, const int32_t* flags, hipStream_t stream) {
  const int m = a.m;
  const size_t lds = sizeof(double) * (size_t)(4 * m * m + 3 * m + 4 * ((m + 1) / 2) + 2);
  if (lds > KF_LDS_BYTES_MAX || a.r > 3 * m) {
    KF_SET_ERR(h, "smoother(pinv_wide): m=%d needs %zu B of LDS", m, lds);
    return KF_ERR_UNSUPPORTED;
  }
  KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_smooth_pinv_wide_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kf_smooth_pinv_wide_kernel, dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a, flags);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

int kf_launch_smoother_lds(kf_handle* h, const KfSmoothArgs& a, hipStream_t stream) {
  int G = kf_pick_lanes(h, a.m, a.B);
  if (a.only_flagged) G = KF_WAVE;
  while (G < KF_WAVE &&
         sizeof(double) * (size_t)(KF_WAVE / G) * smooth_region_doubles(a.m) > KF_LDS_BYTES_MAX / 2)
    G *= 2;
  switch (G) {
    case 4: return launch_g<4>(h, a, stream);
    case 8: return launch_g<8>(h, a, stream);
    case 16: return launch_g<16>(h, a, stream);
    case 32: return launch_g<32>(h, a, stream);
    default: return launch_g<64>(h, a, stream);
  }
}
