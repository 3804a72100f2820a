// kf_backward_big.hip -- reverse-mode gradient of the Kalman log-likelihood for LARGE state
// dimension (m > 16 with scalar observations: p == 1, the Univariate filter, or a diagonal H
// asserted by the caller; BASELINE config C5, m = 367, whose NUTS use needs exactly this).
//
// Maths = kf_backward_mfma.hip's (scalar-observation rank-one adjoints, symmetric adjoint tile;
// derivation: oracle/kalman_adjoint_numpy.py); here every m x m matrix of a draw lives in a
// caller-provided HBM workspace and a time step is a short sequence of launches:
//   bb_fwd_step   per draw: recompute the p scalar updates from the predicted tape (M_q = P^q z_q
//                 column-wise, F_q, v_q, P^{q+1} = P^q - M_q M_q'/F_q), a_f; Gs += Pb;
//                 ab_f = T' ab';  rows of the rank-one part of T_b are parked for one final GEMM
//   sb_gemm x3    X1 = Pb T ;  Sb = X1' T (X1' from a transpose kernel) ;  Tb_m += X1 P_f
//   bb_adj_step   per draw, observations backwards: s = Sb M, beta, gamma, F_b, v_b, u;
//                 P^q rebuilt in the pass that also forms P^q u;  Sb += F_b z z' + (u z' + z u')/2
// GEMMs: the fp64-MFMA sub-block kernel of kf_smoother_big.hip (declared in kf_internal.h).
// g_P0: exact for p == 1, symmetric part for p > 1; g_H: diagonal only (see include/kf.h).
#include "kf_internal.h"

namespace {

constexpr int NT = 512;

__device__ __forceinline__ double block_sum(double v, double* red) {
  // red: NT/64 doubles of shared memory
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, KF_WAVE);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NT / KF_WAVE; ++w) s += red[w];
  return s;
}

struct BbPtrs {
  double* Pc;    // [B][mm] working covariance
  double* PB;    // [B][mm] adjoint of the predicted covariance of step t+1 (symmetric)
  double* SB;    // [B][mm] T' Gm T, then the adjoint for step t
  double* GS;    // [B][mm] sum of Gm
  double* AbT;   // [B][m][n] ab'_t as columns
  double* Af;    // [B][n][m] a_f,t as rows
  double* MQ;    // [B][p][m]
  double* AQ;    // [B][p][m]
  double* SCq;   // [B][p][4]  F, v, 1/F
  double* abp;   // [B][m] adjoint of the predicted state of step t+1
  double* abf;   // [B][m]
  double* zb;    // [B][p][m]
  double* hb;    // [B][p]
  double* db;    // [B][p]
  double* cb;    // [B][m]
  double* GR;    // [B][m][r]  Gs R
};

// ---- forward recompute of step t + the O(m^2) bookkeeping of the predict adjoint ------------------------
__global__ __launch_bounds__(NT) void bb_fwd_step_kernel(const KfBwdArgs A, const BbPtrs W, int t) {
  extern __shared__ double sh[];
  const int m = A.m, p = A.p, n = A.T;
  double* av = sh;          // m  state
  double* Ms = av + m;      // m  M_q
  double* zs = Ms + m;      // m  Z_q
  double* abs_ = zs + m;    // m  ab'
  double* red = abs_ + m;   // NT/64
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const long mm = (long)m * m;
  const double* Ppred = A.pred_P + (b * (long)(n + 1) + t) * mm;
  double* Pc = W.Pc + b * mm;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double* yrow = A.y + (((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L) + t) * p;

  for (int j = tid; j < m; j += NT) {
    av[j] = A.pred_a[(b * (long)(n + 1) + t) * m + j];
    abs_[j] = W.abp[b * m + j];
  }
  __syncthreads();
  for (int q = 0; q < p; ++q) {
    const double* src = (q == 0) ? Ppred : Pc;
    for (int j = tid; j < m; j += NT) zs[j] = A.Z[(b * p + q) * (long)m + j];
    __syncthreads();
    // M_j = sum_i P[i][j] z_i  (P symmetric: column sums are coalesced)
    for (int j = tid; j < m; j += NT) {
      double s = 0.0;
      for (int i = 0; i < m; ++i) s = fma(src[(long)i * m + j], zs[i], s);
      Ms[j] = s;
    }
    __syncthreads();
    double fpart = 0.0, apart = 0.0;
    for (int j = tid; j < m; j += NT) {
      fpart = fma(zs[j], Ms[j], fpart);
      apart = fma(zs[j], av[j], apart);
    }
    const double F = A.H[(b * p + q) * (long)p + q] + block_sum(fpart, red);
    const double za = block_sum(apart, red);
    const double yq = yrow[q];
    const bool obs = !(yq != yq);
    const bool degenerate = !obs || (univariate && F == 0.0);
    const double invF = degenerate ? 0.0 : 1.0 / F;
    const double v = degenerate ? 0.0 : (yq + dsign * (A.d ? A.d[b * p + q] : 0.0)) - za;
    for (int j = tid; j < m; j += NT) {
      W.MQ[(b * p + q) * (long)m + j] = Ms[j];
      W.AQ[(b * p + q) * (long)m + j] = av[j];
    }
    if (tid == 0) {
      W.SCq[(b * p + q) * 4] = F;
      W.SCq[(b * p + q) * 4 + 1] = v;
      W.SCq[(b * p + q) * 4 + 2] = invF;
    }
    // P^{q+1} = P^q - M M'/F
    for (long e = tid; e < mm; e += NT) {
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      Pc[e] = fma(-(Ms[i] * Ms[j]), invF, src[e]);
    }
    __syncthreads();
    for (int j = tid; j < m; j += NT) av[j] = fma(Ms[j] * invF, v, av[j]);
    __syncthreads();
  }
  // a_f and ab' parked for the final GEMM  sum_t ab'_t a_f,t' ;  c_b += ab'
  for (int j = tid; j < m; j += NT) {
    W.Af[(b * n + t) * (long)m + j] = av[j];
    W.AbT[(b * m + j) * (long)n + t] = abs_[j];
    W.cb[b * m + j] += abs_[j];
  }
  // Gs += Pb ;  ab_f = T' ab'
  const double* PB = W.PB + b * mm;
  double* GS = W.GS + b * mm;
  for (long e = tid; e < mm; e += NT) GS[e] += PB[e];
  const double* gT = A.Tm + b * mm;
  for (int j = tid; j < m; j += NT) {
    double s = 0.0;
    for (int i = 0; i < m; ++i) s = fma(gT[(long)i * m + j], abs_[i], s);
    W.abf[b * m + j] = s;
  }
}

// ---- adjoint of the p scalar updates of step t (Sb = T' Gm T is in W.SB) --------------------------------------
__global__ __launch_bounds__(NT) void bb_adj_step_kernel(const KfBwdArgs A, const BbPtrs W, int t) {
  extern __shared__ double sh[];
  const int m = A.m, p = A.p, n = A.T;
  double* Ms = sh;         // m
  double* zs = Ms + m;     // m
  double* sv = zs + m;     // m  s = Sb M
  double* uv = sv + m;     // m  u
  double* af = uv + m;     // m  ab_f
  double* m1 = af + m;     // m  m1b (exact g_P0 only)
  double* red = m1 + m;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const long mm = (long)m * m;
  double* Pc = W.Pc + b * mm;
  double* SB = W.SB + b * mm;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double gt = (A.g_logp ? A.g_logp[b] : 0.0) + (A.g_ll_obs ? A.g_ll_obs[b * (long)n + t] : 0.0);
  for (int j = tid; j < m; j += NT) af[j] = W.abf[b * m + j];
  __syncthreads();
  for (int q = p - 1; q >= 0; --q) {
    for (int j = tid; j < m; j += NT) {
      Ms[j] = W.MQ[(b * p + q) * (long)m + j];
      zs[j] = A.Z[(b * p + q) * (long)m + j];
    }
    const double v = W.SCq[(b * p + q) * 4 + 1], invF = W.SCq[(b * p + q) * 4 + 2];
    __syncthreads();
    // s_j = sum_i Sb[i][j] M_i  (Sb symmetric)
    for (int j = tid; j < m; j += NT) {
      double s = 0.0;
      for (int i = 0; i < m; ++i) s = fma(SB[(long)i * m + j], Ms[i], s);
      sv[j] = s;
    }
    __syncthreads();
    double bpart = 0.0, gpart = 0.0;
    for (int j = tid; j < m; j += NT) {
      bpart = fma(Ms[j], af[j], bpart);
      gpart = fma(Ms[j], sv[j], gpart);
    }
    const double beta = block_sum(bpart, red);
    const double gamma = block_sum(gpart, red);
    const double Fb = (gamma - beta * v) * invF * invF - 0.5 * gt * (invF - v * v * invF * invF);
    const double vb = (beta - gt * v) * invF;
    for (int j = tid; j < m; j += NT) {
      uv[j] = (af[j] * v - 2.0 * sv[j]) * invF;
      m1[j] = (af[j] * v - sv[j]) * invF;
    }
    __syncthreads();
    // P^q = P^{q+1} + M M'/F, and in the same pass  pdot_j = sum_i P^q[i][j] u_i
    for (int j = tid; j < m; j += NT) {
      double s = 0.0;
      const double mj = Ms[j] * invF;
      for (int i = 0; i < m; ++i) {
        const double pv = fma(Ms[i], mj, Pc[(long)i * m + j]);
        Pc[(long)i * m + j] = pv;
        s = fma(pv, uv[i], s);
      }
      const double aq = W.AQ[(b * p + q) * (long)m + j];
      W.zb[(b * p + q) * (long)m + j] += fma(-aq, vb, fma(2.0 * Fb, Ms[j], s));
    }
    // exact entrywise g_P0 (p == 1): Sb + (Fb z + m1b) z' - z s'/F
    if (p == 1 && t == 0 && A.g_P0) {
      for (long e = tid; e < mm; e += NT) {
        const int i = (int)(e / m), j = (int)(e - (long)i * m);
        double gp = SB[e];
        gp = fma(fma(Fb, zs[i], m1[i]), zs[j], gp);
        gp = fma(-zs[i] * invF, sv[j], gp);
        A.g_P0[b * mm + e] = gp;
      }
    }
    // symmetrised adjoint of P^q ; adjoint of a^q
    for (long e = tid; e < mm; e += NT) {
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double sb = SB[e];
      sb = fma(Fb * zs[i], zs[j], sb);
      sb = fma(0.5 * uv[i], zs[j], sb);
      sb = fma(0.5 * zs[i], uv[j], sb);
      SB[e] = sb;
    }
    __syncthreads();
    for (int j = tid; j < m; j += NT) af[j] = fma(-zs[j], vb, af[j]);
    if (tid == 0) {
      W.hb[b * p + q] += Fb;
      W.db[b * p + q] += dsign * vb;
    }
    __syncthreads();
  }
  for (int j = tid; j < m; j += NT) W.abp[b * m + j] = af[j];
}

__global__ __launch_bounds__(256) void bb_scale_kernel(double* X, long count, double f) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < count) X[e] *= f;
}

// small outputs and R / Q gradients from Gs
__global__ __launch_bounds__(256) void bb_finish_kernel(const KfBwdArgs A, const BbPtrs W) {
  const int m = A.m, p = A.p, r = A.r;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const long mm = (long)m * m;
  for (int j = tid; j < m; j += 256) {
    if (A.g_a0) A.g_a0[b * m + j] = W.abp[b * m + j];
    if (A.g_c) A.g_c[b * m + j] = W.cb[b * m + j];
  }
  if (A.g_Z)
    for (int e = tid; e < p * m; e += 256) A.g_Z[b * (long)p * m + e] = W.zb[b * (long)p * m + e];
  if (A.g_H)
    for (int e = tid; e < p * p; e += 256) {
      const int q1 = e / p, q2 = e - q1 * p;
      A.g_H[b * (long)p * p + e] = (q1 == q2) ? W.hb[b * p + q1] : 0.0;
    }
  if (A.g_d && tid < p) A.g_d[b * p + tid] = W.db[b * p + tid];
  if (A.g_P0 && p > 1)
    for (long e = tid; e < mm; e += 256) A.g_P0[b * mm + e] = W.PB[b * mm + e];
  const double* GS = W.GS + b * mm;
  const double* gR = A.R + b * (long)m * r;
  const double* gQ = A.Q + b * (long)r * r;
  // GR = Gs R (Gs symmetric: column sums are coalesced) ; R_b = GR (Q + Q') ; Q_b = R' GR
  double* GR = W.GR + b * (long)m * r;
  for (int e = tid; e < m * r; e += 256) {
    const int k = e / m, i = e - k * m;  // consecutive threads -> consecutive columns of Gs
    double s = 0.0;
    for (int j = 0; j < m; ++j) s = fma(GS[(long)j * m + i], gR[j * r + k], s);
    GR[i * r + k] = s;
  }
  __syncthreads();
  if (A.g_R)
    for (int e = tid; e < m * r; e += 256) {
      const int i = e / r, l = e - i * r;
      double sacc = 0.0;
      for (int k = 0; k < r; ++k) sacc = fma(GR[i * r + k], gQ[k * r + l] + gQ[l * r + k], sacc);
      A.g_R[b * (long)m * r + e] = sacc;
    }
  if (A.g_Q)
    for (int e = tid; e < r * r; e += 256) {
      const int k = e / r, l = e - k * r;
      double sacc = 0.0;
      for (int i = 0; i < m; ++i) sacc = fma(gR[i * r + k], GR[i * r + l], sacc);
      A.g_Q[b * (long)r * r + e] = sacc;
    }
}

inline size_t bb_doubles(long B, int m, int p, int n, int r) {
  const long mm = (long)m * m;
  return (size_t)B * (size_t)(7 * mm + 2L * m * n + 3L * p * m + 4L * p + 3L * m + 2L * p + (long)m * r);
}

}  // namespace

size_t kf_backward_big_workspace_bytes(long B, int m, int p, int n, int r) {
  return sizeof(double) * bb_doubles(B, m, p, n, r) + 256;
}

int kf_launch_backward_big(kf_handle* h, const KfBwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const long B = a.B;
  const int m = a.m, p = a.p, n = a.T;
  const long mm = (long)m * m;
  const size_t need = kf_backward_big_workspace_bytes(B, m, p, n, a.r);
  if (!workspace || workspace_bytes < need) {
    KF_SET_ERR(h, "backward(big): m=%d needs a workspace of %zu bytes (kf_workspace_bytes), got %zu", m, need,
               workspace_bytes);
    return KF_ERR_WORKSPACE;
  }
  if (!(a.p == 1 || a.filter == KF_FILTER_UNIVARIATE || (a.flags & KF_FLAG_H_DIAGONAL))) {
    KF_SET_ERR(h, "backward(big): m=%d > 16 needs scalar observations: p == 1, the Univariate filter or "
               "KF_FLAG_H_DIAGONAL (got p=%d)", m, p);
    return KF_ERR_UNSUPPORTED;
  }
  if (a.flags & KF_FLAG_TV_ANY) {
    KF_SET_ERR(h, "backward(big): time-varying system matrices are not supported");
    return KF_ERR_UNSUPPORTED;
  }
  double* ws = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  KF_HIP_CHECK(h, hipMemsetAsync(ws, 0, sizeof(double) * bb_doubles(B, m, p, n, a.r), st));
  BbPtrs W;
  double* q = ws;
  W.Pc = q; q += B * mm;
  W.PB = q; q += B * mm;
  W.SB = q; q += B * mm;
  W.GS = q; q += B * mm;
  double* X1 = q; q += B * mm;
  double* X1t = q; q += B * mm;
  double* TBm = q; q += B * mm;
  W.AbT = q; q += B * (long)m * n;
  W.Af = q; q += B * (long)n * m;
  W.MQ = q; q += B * (long)p * m;
  W.AQ = q; q += B * (long)p * m;
  W.zb = q; q += B * (long)p * m;
  W.SCq = q; q += B * 4L * p;
  W.abp = q; q += B * (long)m;
  W.abf = q; q += B * (long)m;
  W.cb = q; q += B * (long)m;
  W.hb = q; q += B * (long)p;
  W.db = q; q += B * (long)p;
  W.GR = q; q += B * (long)m * a.r;

  const size_t sh_f = sizeof(double) * (4 * m + NT / KF_WAVE);
  const size_t sh_a = sizeof(double) * (6 * m + NT / KF_WAVE);
  const unsigned tb = (unsigned)((m + 31) / 32);
  for (int t = n - 1; t >= 0; --t) {
    hipLaunchKernelGGL(bb_fwd_step_kernel, dim3((unsigned)B), dim3(NT), sh_f, st, a, W, t);
    // X1 = Pb T ; Sb = X1' T ; Tb_m += X1 P_f
    kf_sb_gemm(st, B, false, 0, W.PB, mm, m, a.Tm, mm, m, nullptr, 0, 0, X1, mm, m, m, m, m);
    kf_sb_transpose(st, B, X1, mm, X1t, mm, m);
    kf_sb_gemm(st, B, false, 0, X1t, mm, m, a.Tm, mm, m, nullptr, 0, 0, W.SB, mm, m, m, m, m);
    kf_sb_gemm(st, B, false, 1, X1, mm, m, W.Pc, mm, m, TBm, mm, m, TBm, mm, m, m, m, m);
    hipLaunchKernelGGL(bb_adj_step_kernel, dim3((unsigned)B), dim3(NT), sh_a, st, a, W, t);
    double* tmp = W.PB;  // the updated Sb is the adjoint of the predicted covariance of step t
    W.PB = W.SB;
    W.SB = tmp;
  }
  (void)tb;
  // g_T = 2 Tb_m + sum_t ab'_t a_f,t'
  if (a.g_T) {
    hipLaunchKernelGGL(bb_scale_kernel, dim3((unsigned)((B * mm + 255) / 256)), dim3(256), 0, st, TBm, B * mm, 2.0);
    kf_sb_gemm(st, B, false, 1, W.AbT, (long)m * n, n, W.Af, (long)n * m, m, TBm, mm, m, a.g_T, mm, m, m, m, n);
  }
  hipLaunchKernelGGL(bb_finish_kernel, dim3((unsigned)B), dim3(256), 0, st, a, W);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
