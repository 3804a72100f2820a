// kf_backward_rowsplit.hip -- register-resident reverse-mode gradient of the Kalman
// log-likelihood for p = 1 and compile-time even M = k_states <= 10 (the NUTS hot loop of the
// reference is logp + gradient: SURVEY.md section 3.2).
//
// Same contract as kf_bwd_lds_kernel (adjoint of StandardFilter.update + predict,
// reference pymc_statespace/filters/kalman_filter.py:215-223, 255-284; derivation and symbols:
// oracle/kalman_adjoint_numpy.py).  For p = 1 the Joseph form is, for a general (not necessarily
// symmetric) P, EXACTLY  P_f = P - (P z)(z'P)/F,  so the adjoint of the update collapses to
// rank-one corrections (z = Z', M = P z, s = Sb M, beta = M'ab_f, gamma = M's):
//     Fb  = (gamma - beta v)/F^2 - g/2 (1/F - v^2/F^2)        vb = beta/F - g v/F
//     m1b = (ab_f v - s)/F        m2b = -s/F
//     Pb  = Sb + Fb z z' + m1b z' + z m2b'                     ab = ab_f - z vb
//     zb += -a vb + 2 Fb M + P (m1b + m2b)        hb += Fb     db -= vb
// and only the predict step keeps m x m x m products:  W = T P_f,  Gm W,  X1 = Gm T,
// Sb = T' X1  (Gm = sym(Pb')).
//
// Mapping = kf_fwd_rowsplit_kernel's: M*H lanes per draw, lane (i,h) owns row i and the column
// pairs c = h, h+H, ...; four wave-local LDS exchanges per time step:
//   (1) all-gather (M_i, a_i)                         (2) P_f rows, Pb' rows, ab'
//   (3) W rows, X1 rows                               (4) partial s, ab_f
// The predicted tape row of step t-1 is prefetched from HBM while step t computes.
#include "kf_internal.h"

#if !defined(KF_ROW_M) || !defined(KF_ROW_H)
#error "compile with -DKF_ROW_M=<even k_states> -DKF_ROW_H=<lanes per row>"
#endif

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int M, int H>
struct BwdCfg {
  static_assert(M % 2 == 0, "row-split kernel needs an even state dimension");
  static constexpr int MM = M * M;
  static constexpr int NPAIR = M / 2;
  static constexpr int NP = (NPAIR + H - 1) / H;
  static constexpr int LPD = M * H;
  static_assert(LPD <= KF_WAVE, "M*H must fit a wavefront");
  static constexpr int DPW = KF_WAVE / LPD;
  // TL (T), X (P_f), Y (Pb'), U1 (W), U2 (X1), V1 (2M: M_i,a_i), V2 (M: ab'), V3 (H*M: s parts),
  // V4 (M: ab_f)
  static constexpr int RAW = 5 * MM + 2 * M + M + H * M + M + 2;
  static constexpr int STRIDE = ((RAW + 29) / 32) * 32 + 2;
  static constexpr int LDS_DOUBLES = DPW * STRIDE;
};

template <int M, int H>
__global__ __launch_bounds__(KF_WAVE) void kf_bwd_rowsplit_kernel(const KfBwdArgs A) {
  using C = BwdCfg<M, H>;
  constexpr int MM = C::MM, NP = C::NP;
  __shared__ __attribute__((aligned(16))) double smem[C::LDS_DOUBLES];
  const int lane = threadIdx.x;
  int grp = lane / C::LPD;
  const int lid = lane - grp * C::LPD;
  const int i = lid / H;
  const int h = lid - i * H;
  const bool lane_ok = grp < C::DPW;
  if (!lane_ok) grp = C::DPW - 1;
  long b = (long)blockIdx.x * C::DPW + grp;
  const bool live = lane_ok && b < A.B;
  if (b >= A.B) b = A.B - 1;
  const int n = A.T, r = A.r;

  double* TL = smem + grp * C::STRIDE;
  double* X = TL + MM;
  double* Y = X + MM;
  double* U1 = Y + MM;
  double* U2 = U1 + MM;
  double* V1 = U2 + MM;     // (M_k, a_k) pairs
  double* V2 = V1 + 2 * M;  // ab'
  double* V3 = V2 + M;      // s partials, [h][k]
  double* V4 = V3 + H * M;  // ab_f

  int j0[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    int c = h + pp * H;
    if (c > C::NPAIR - 1) c = C::NPAIR - 1;
    j0[pp] = 2 * c;
  }
  // clones (surplus pair slots of a lane) must not double count in the accumulators
  bool own_pair[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) own_pair[pp] = (h + pp * H) <= C::NPAIR - 1;

  // ---- static per-draw data ------------------------------------------------------------------------
  double Trow[M], Tcol[M], Zr[M], Zo[NP][2];
  {
    const double* gT = A.Tm + b * MM;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      Trow[k] = gT[i * M + k];
      Tcol[k] = gT[k * M + i];
      TL[i * M + k] = Trow[k];
      Zr[k] = A.Z[b * M + k];
    }
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      Zo[pp][0] = A.Z[b * M + j0[pp]];
      Zo[pp][1] = A.Z[b * M + j0[pp] + 1];
    }
  }
  const double Zi = A.Z[b * M + i];
  const double Hs = A.H[b];
  const double ds = A.d ? A.d[b] : 0.0;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
  const double* gpa = A.pred_a + b * (long)(n + 1) * M + i;
  const double* gpP = A.pred_P + b * (long)(n + 1) * MM + i * M;
  const double g_all = A.g_logp ? A.g_logp[b] : 0.0;
  const double* gll = A.g_ll_obs ? A.g_ll_obs + b * (long)n : nullptr;

  // adjoint state and accumulators
  double ab = 0.0, zb = 0.0, cb = 0.0, hb = 0.0, db = 0.0;
  double Pbo[NP][2], Tbo[NP][2], Gso[NP][2];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int e = 0; e < 2; ++e) Pbo[pp][e] = Tbo[pp][e] = Gso[pp][e] = 0.0;

  // tape prefetch (row i of P_pred[t], a_pred[t][i], own pairs)
  double Prow_n[M], Po_n[NP][2], a_n;
  auto prefetch = [&](int t) {
    const double* rp = gpP + (long)t * MM;
#pragma unroll
    for (int k = 0; k < M; ++k) Prow_n[k] = rp[k];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      Po_n[pp][0] = rp[j0[pp]];
      Po_n[pp][1] = rp[j0[pp] + 1];
    }
    a_n = gpa[(long)t * M];
  };
  prefetch(n - 1);
  __syncthreads();  // TL visible

  for (int t = n - 1; t >= 0; --t) {
    double Prow[M], Po[NP][2];
#pragma unroll
    for (int k = 0; k < M; ++k) Prow[k] = Prow_n[k];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) { Po[pp][0] = Po_n[pp][0]; Po[pp][1] = Po_n[pp][1]; }
    const double ai = a_n;
    const double yv = ybase[t];
    const double g = g_all + (gll ? gll[t] : 0.0);
    if (t > 0) prefetch(t - 1);
    const bool obs = !(yv != yv);

    // ---- (1) M_i = P[i,:] z ; gather (M, a) -----------------------------------------------------------
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      m0 = fma(Prow[k], Zr[k], m0);
      m1 = fma(Prow[k + 1], Zr[k + 1], m1);
    }
    const double Mi = m0 + m1;
    {
      double2_t t2 = {Mi, ai};
      *reinterpret_cast<double2_t*>(V1 + 2 * i) = t2;
    }
    __syncthreads();
    double Mall[M];
    double F0 = Hs, F1 = 0.0, z0 = 0.0, z1 = 0.0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      Mall[k] = V1[2 * k];
      Mall[k + 1] = V1[2 * k + 2];
      F0 = fma(Zr[k], Mall[k], F0);
      F1 = fma(Zr[k + 1], Mall[k + 1], F1);
      z0 = fma(Zr[k], V1[2 * k + 1], z0);
      z1 = fma(Zr[k + 1], V1[2 * k + 3], z1);
    }
    const double F = F0 + F1;
    const double invF = obs ? 1.0 / F : 0.0;
    const double v = obs ? (yv + dsign * ds) - (z0 + z1) : 0.0;
    double Mo[NP][2], ao[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      const double2_t q0 = *reinterpret_cast<const double2_t*>(V1 + 2 * j0[pp]);
      const double2_t q1 = *reinterpret_cast<const double2_t*>(V1 + 2 * j0[pp] + 2);
      Mo[pp][0] = q0.x; ao[pp][0] = q0.y;
      Mo[pp][1] = q1.x; ao[pp][1] = q1.y;
    }
    const double Ki = Mi * invF;
    // ---- (2) P_f own pairs, Pb' rows, ab' -> tiles -----------------------------------------------------
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      double2_t pf = {Po[pp][0] - Ki * Mo[pp][0], Po[pp][1] - Ki * Mo[pp][1]};  // P - M M'/F
      *reinterpret_cast<double2_t*>(X + i * M + j0[pp]) = pf;
      double2_t pb = {Pbo[pp][0], Pbo[pp][1]};
      *reinterpret_cast<double2_t*>(Y + i * M + j0[pp]) = pb;
    }
    V2[i] = ab;
    __syncthreads();
    // Gm row i = sym(Pb')[i,:] ; own pairs of Gm ; ab_f = T' ab'
    double Gm[M];
#pragma unroll
    for (int k = 0; k < M; ++k) Gm[k] = 0.5 * (Y[i * M + k] + Y[k * M + i]);
    double Gmo[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      Gmo[pp][0] = 0.5 * (Pbo[pp][0] + Y[j0[pp] * M + i]);
      Gmo[pp][1] = 0.5 * (Pbo[pp][1] + Y[(j0[pp] + 1) * M + i]);
    }
    double abf = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) abf = fma(Tcol[k], V2[k], abf);
    // W own = T[i,:] P_f[:,own] ; X1 own = Gm[i,:] T[:,own]
    double Wo[NP][2], X1o[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) Wo[pp][0] = Wo[pp][1] = X1o[pp][0] = X1o[pp][1] = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double tk = Trow[k], gk = Gm[k];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const double2_t x2 = *reinterpret_cast<const double2_t*>(X + k * M + j0[pp]);
        const double2_t t2 = *reinterpret_cast<const double2_t*>(TL + k * M + j0[pp]);
        Wo[pp][0] = fma(tk, x2.x, Wo[pp][0]);
        Wo[pp][1] = fma(tk, x2.y, Wo[pp][1]);
        X1o[pp][0] = fma(gk, t2.x, X1o[pp][0]);
        X1o[pp][1] = fma(gk, t2.y, X1o[pp][1]);
      }
    }
    cb += ab;
    // ---- (3) W rows, X1 rows -> tiles ---------------------------------------------------------------------
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      double2_t w2 = {Wo[pp][0], Wo[pp][1]};
      *reinterpret_cast<double2_t*>(U1 + i * M + j0[pp]) = w2;
      double2_t x2 = {X1o[pp][0], X1o[pp][1]};
      *reinterpret_cast<double2_t*>(U2 + i * M + j0[pp]) = x2;
    }
    __syncthreads();
    // (Gm W) own, Sb own = T[:,i]' X1[:,own]
    double GWo[NP][2], Sbo[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) GWo[pp][0] = GWo[pp][1] = Sbo[pp][0] = Sbo[pp][1] = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double gk = Gm[k], tk = Tcol[k];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const double2_t w2 = *reinterpret_cast<const double2_t*>(U1 + k * M + j0[pp]);
        const double2_t x2 = *reinterpret_cast<const double2_t*>(U2 + k * M + j0[pp]);
        GWo[pp][0] = fma(gk, w2.x, GWo[pp][0]);
        GWo[pp][1] = fma(gk, w2.y, GWo[pp][1]);
        Sbo[pp][0] = fma(tk, x2.x, Sbo[pp][0]);
        Sbo[pp][1] = fma(tk, x2.y, Sbo[pp][1]);
      }
    }
    // accumulate Tb += ab' a_f' + 2 Gm W ; Gs += Gm ; partial s = Sb[i,own] . M[own]
    double sp = 0.0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double afj = fma(Mo[pp][e] * invF, v, ao[pp][e]);
        Tbo[pp][e] += fma(ab, afj, 2.0 * GWo[pp][e]);
        Gso[pp][e] += Gmo[pp][e];
        if (own_pair[pp]) sp = fma(Sbo[pp][e], Mo[pp][e], sp);
      }
    }
    // ---- (4) gather partial s and ab_f --------------------------------------------------------------------
    V3[h * M + i] = sp;
    V4[i] = abf;
    __syncthreads();
    double beta = 0.0, gamma = 0.0, pdot = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      double sk = V3[k];
#pragma unroll
      for (int hh = 1; hh < H; ++hh) sk += V3[hh * M + k];
      const double afk = V4[k];
      beta = fma(Mall[k], afk, beta);
      gamma = fma(Mall[k], sk, gamma);
      pdot = fma(Prow[k], (afk * v - 2.0 * sk) * invF, pdot);  // P[i,:] (m1b + m2b)
    }
    const double Fb = (gamma - beta * v) * invF * invF - 0.5 * g * (invF - v * v * invF * invF);
    const double vb = (beta - g * v) * invF;
    double si = V3[i];
#pragma unroll
    for (int hh = 1; hh < H; ++hh) si += V3[hh * M + i];
    const double m1bi = (abf * v - si) * invF;
    // new Pb own pairs = Sb + Fb z z' + m1b z' + z m2b'
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        double sj = V3[j0[pp] + e];
#pragma unroll
        for (int hh = 1; hh < H; ++hh) sj += V3[hh * M + j0[pp] + e];
        const double zj = Zo[pp][e];
        double pbn = Sbo[pp][e];
        pbn = fma(Fb * Zi + m1bi, zj, pbn);
        pbn = fma(-Zi * invF, sj, pbn);
        Pbo[pp][e] = pbn;
      }
    }
    zb += fma(-ai, vb, fma(2.0 * Fb, Mi, pdot));
    ab = fma(-Zi, vb, abf);
    if (obs) {
      hb += Fb;
      db += dsign * vb;
    }
  }

  // ---- gradients out ----------------------------------------------------------------------------------------
  // Gs (own pairs) -> tile X for  R_b = Gs R (Q + Q'),  Q_b = R' Gs R
  __syncthreads();
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    double2_t g2 = {Gso[pp][0], Gso[pp][1]};
    *reinterpret_cast<double2_t*>(X + i * M + j0[pp]) = g2;
  }
  __syncthreads();
  if (live) {
    if (A.g_a0 && h == 0) A.g_a0[b * M + i] = ab;
    if (A.g_c && h == 0) A.g_c[b * M + i] = cb;
    if (A.g_Z && h == 0) A.g_Z[b * M + i] = zb;
    if (lid == 0) {
      if (A.g_H) A.g_H[b] = hb;
      if (A.g_d) A.g_d[b] = db;
    }
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      if (!own_pair[pp]) continue;
      if (A.g_P0) {
        A.g_P0[b * MM + i * M + j0[pp]] = Pbo[pp][0];
        A.g_P0[b * MM + i * M + j0[pp] + 1] = Pbo[pp][1];
      }
      if (A.g_T) {
        A.g_T[b * MM + i * M + j0[pp]] = Tbo[pp][0];
        A.g_T[b * MM + i * M + j0[pp] + 1] = Tbo[pp][1];
      }
    }
    const double* gR = A.R + b * (long)M * r;
    const double* gQ = A.Q + b * (long)r * r;
    if (A.g_R && h == 0) {
      for (int l = 0; l < r; ++l) {
        double s = 0.0;
        for (int k = 0; k < r; ++k) {
          double grk = 0.0;  // (Gs R)[i][k]
#pragma unroll
          for (int j = 0; j < M; ++j) grk = fma(X[i * M + j], gR[j * r + k], grk);
          s = fma(grk, gQ[k * r + l] + gQ[l * r + k], s);
        }
        A.g_R[b * (long)M * r + i * r + l] = s;
      }
    }
    if (A.g_Q) {
      for (int e = lid; e < r * r; e += C::LPD) {
        const int k = e / r, l = e - k * r;
        double s = 0.0;
        for (int ii = 0; ii < M; ++ii) {
          double grl = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) grl = fma(X[ii * M + j], gR[j * r + l], grl);
          s = fma(gR[ii * r + k], grl, s);
        }
        A.g_Q[b * (long)r * r + e] = s;
      }
    }
  }
}

}  // namespace

#define KF_CAT3_(a, b, c, d) a##b##c##d
#define KF_CAT3(a, b, c, d) KF_CAT3_(a, b, c, d)
int KF_CAT3(kf_launch_backward_rowsplit_m, KF_ROW_M, _h, KF_ROW_H)(kf_handle* h, const KfBwdArgs& a,
                                                                  hipStream_t stream) {
  constexpr int M = KF_ROW_M, HH = KF_ROW_H;
  using C = BwdCfg<M, HH>;
  long blocks = (a.B + C::DPW - 1) / C::DPW;
  hipLaunchKernelGGL((kf_bwd_rowsplit_kernel<M, HH>), dim3((unsigned)blocks), dim3(KF_WAVE), 0, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
