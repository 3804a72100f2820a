// kf_dare.hip -- batched steady-state (filtering) Riccati solve and batched unconditional
// simulation: SURVEY.md section 8f, row 4 (the reference's SteadyStateFilter and
// simulate_statespace).
//
// kf_dare: reference filters/kalman_filter.py:388-390 solves, once per draw on the host,
//     P = solve_discrete_are(T', Z', R Q R', H)        F_inv = (Z P Z' + H)^-1
// i.e. the fixed point of the filter's own covariance recursion
//     P <- T (P - P Z' (Z P Z' + H)^-1 Z P) T' + R Q R'.
// Here all B draws iterate that recursion in LDS (one wavefront per draw) until the largest
// change of an entry is <= tol x (1 + max |P|); info[b] = 0 converged, 1 = max_iter reached,
// 2 = Z P Z' + H not positive definite.  (A structured-doubling solver would need H^-1; the
// models of the reference routinely have H = 0.)
//
// kf_simulate_statespace: reference utils/simulation.py:29-64 (simulate_statespace), one LANE per
// trajectory, with the reference's exact recursion -- including its conventions: innovations are
// `noise @ chol` (row vector times the LOWER factor), the observation at t reads the state at
// t-1, and row 0 holds (x0, Z x0) or zeros.  Standard normals are an input.
#include "kf_internal.h"

namespace {

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, KF_WAVE));
  return v;
}

__host__ __device__ inline int dare_region_doubles(int m, int p) {
  return 6 * m * m + 3 * m * p + 3 * p * p + 8;  // T, RQR, P, Pf, W, S | Z, M, K | H, F, L
}

__global__ __launch_bounds__(KF_WAVE) void kf_dare_kernel(long B, int m, int p, int r, const double* Tm,
                                                          const double* Zm, const double* Rm, const double* Hm,
                                                          const double* Qm, const double* Pinit, double* Pout,
                                                          int32_t* info, int max_iter, double tol) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const int mm = m * m, mp = m * p, pp = p * p;
  double* Tl = smem;
  double* RQR = Tl + mm;
  double* P = RQR + mm;
  double* Pf = P + mm;
  double* W = Pf + mm;
  double* S = W + mm;
  double* Zl = S + mm;
  double* M = Zl + mp;
  double* K = M + mp;
  double* Hl = K + mp;
  double* F = Hl + pp;
  double* L = F + pp;
  int* flag = reinterpret_cast<int*>(L + pp);

  const double* gR = Rm + b * (long)m * r;
  const double* gQ = Qm + b * (long)r * r;
  for (int e = lane; e < mm; e += KF_WAVE) {
    Tl[e] = Tm[b * mm + e];
    const int i = e / m, j = e - i * m;
    double s = 0.0;
    for (int l = 0; l < r; ++l) {
      double rq = 0.0;
      for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
      s += rq * gR[j * r + l];
    }
    RQR[e] = s;
    P[e] = Pinit ? Pinit[b * mm + e] : s;
  }
  for (int e = lane; e < mp; e += KF_WAVE) Zl[e] = Zm[b * mp + e];
  for (int e = lane; e < pp; e += KF_WAVE) Hl[e] = Hm[b * pp + e];
  if (lane == 0) *flag = 0;
  __syncthreads();

  int status = 1;
  for (int it = 0; it < max_iter; ++it) {
    // M = P Z'
    for (int e = lane; e < mp; e += KF_WAVE) {
      const int i = e / p, q = e - i * p;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += P[i * m + k] * Zl[q * m + k];
      M[e] = s;
    }
    __syncthreads();
    // F = Z M + H
    for (int e = lane; e < pp; e += KF_WAVE) {
      const int q1 = e / p, q2 = e - q1 * p;
      double s = Hl[e];
      for (int k = 0; k < m; ++k) s += Zl[q1 * m + k] * M[k * p + q2];
      F[e] = s;
    }
    __syncthreads();
    // L = chol(F)  (p is small: one lane)
    if (lane == 0) {
      for (int j = 0; j < p; ++j) {
        double d = F[j * p + j];
        for (int k = 0; k < j; ++k) d -= L[j * p + k] * L[j * p + k];
        if (!(d > 0.0)) {
          *flag = 2;
          d = 1.0;
        }
        d = sqrt(d);
        L[j * p + j] = d;
        for (int i = j + 1; i < p; ++i) {
          double s = F[i * p + j];
          for (int k = 0; k < j; ++k) s -= L[i * p + k] * L[j * p + k];
          L[i * p + j] = s / d;
        }
      }
    }
    __syncthreads();
    if (*flag == 2) {
      status = 2;
      break;
    }
    // K = M F^-1: row i solves K_i L L' = M_i
    for (int i = lane; i < m; i += KF_WAVE) {
      double* k_ = K + i * p;
      for (int j = 0; j < p; ++j) {  // w L' = M_i  ->  forward in j
        double s = M[i * p + j];
        for (int k = 0; k < j; ++k) s -= k_[k] * L[j * p + k];
        k_[j] = s / L[j * p + j];
      }
      for (int j = p - 1; j >= 0; --j) {  // k L = w
        double s = k_[j];
        for (int k = j + 1; k < p; ++k) s -= k_[k] * L[k * p + j];
        k_[j] = s / L[j * p + j];
      }
    }
    __syncthreads();
    // P_f = P - K M' (symmetric by construction: entry (i,j) and (j,i) use the same formula pairwise)
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double sij = 0.0, sji = 0.0;
      for (int q = 0; q < p; ++q) {
        sij += K[i * p + q] * M[j * p + q];
        sji += K[j * p + q] * M[i * p + q];
      }
      Pf[e] = 0.5 * ((P[e] - sij) + (P[j * m + i] - sji));
    }
    __syncthreads();
    // W = T P_f
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[i * m + k] * Pf[k * m + j];
      W[e] = s;
    }
    __syncthreads();
    // S = W T' + RQR'
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double s = RQR[e];
      for (int k = 0; k < m; ++k) s += W[i * m + k] * Tl[j * m + k];
      S[e] = s;
    }
    __syncthreads();
    double dmax = 0.0, pmax = 0.0;
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      const double pn = 0.5 * (S[e] + S[j * m + i]);
      dmax = fmax(dmax, fabs(pn - P[e]));
      pmax = fmax(pmax, fabs(pn));
      Pf[e] = pn;  // staging: P is still read by other lanes' (j,i) entries above? no -- P is not read here
    }
    dmax = wave_max(dmax);
    pmax = wave_max(pmax);
    __syncthreads();
    for (int e = lane; e < mm; e += KF_WAVE) P[e] = Pf[e];
    __syncthreads();
    if (!(dmax > tol * (1.0 + pmax)) && dmax == dmax) {
      status = 0;
      break;
    }
  }
  for (int e = lane; e < mm; e += KF_WAVE) Pout[b * mm + e] = P[e];
  if (info && lane == 0) info[b] = status;
}

// One lane per trajectory (draw b, simulation j).  zs [B][n_sims][n][r], zo [B][n_sims][n][p].
__global__ __launch_bounds__(KF_WAVE) void kf_simulate_ss_kernel(long B, int n_sims, int n, int p, int m, int r,
                                                                 const double* Tm, const double* Zm, const double* Rm,
                                                                 const double* Hm, const double* Qm, const double* x0,
                                                                 const double* zs, const double* zo, double* states,
                                                                 double* obs, int32_t* info) {
  extern __shared__ double smem[];  // per workgroup (= one draw): T, Z, R, LQ, LH
  const long b = blockIdx.y;
  const int lane = threadIdx.x;
  const int j = blockIdx.x * KF_WAVE + lane;
  double* Tl = smem;
  double* Zl = Tl + m * m;
  double* Rl = Zl + p * m;
  double* LQ = Rl + m * r;
  double* LH = LQ + r * r;
  int* st = reinterpret_cast<int*>(LH + p * p);
  for (int e = lane; e < m * m; e += KF_WAVE) Tl[e] = Tm[b * (long)m * m + e];
  for (int e = lane; e < p * m; e += KF_WAVE) Zl[e] = Zm[b * (long)p * m + e];
  for (int e = lane; e < m * r; e += KF_WAVE) Rl[e] = Rm[b * (long)m * r + e];
  if (lane == 0) {
    // lower Cholesky factors (np.linalg.cholesky); H == 0 means "no observation noise" (:33)
    int bad = 0;
    const double* gQ = Qm + b * (long)r * r;
    for (int c = 0; c < r; ++c)
      for (int i = 0; i < r; ++i) LQ[i * r + c] = 0.0;
    for (int c = 0; c < r; ++c) {
      double d = gQ[c * r + c];
      for (int k = 0; k < c; ++k) d -= LQ[c * r + k] * LQ[c * r + k];
      if (!(d > 0.0)) bad = 1;
      d = sqrt(d);
      LQ[c * r + c] = d;
      for (int i = c + 1; i < r; ++i) {
        double s = gQ[i * r + c];
        for (int k = 0; k < c; ++k) s -= LQ[i * r + k] * LQ[c * r + k];
        LQ[i * r + c] = s / d;
      }
    }
    const double* gH = Hm + b * (long)p * p;
    bool hzero = true;
    for (int e = 0; e < p * p; ++e) hzero = hzero && (gH[e] == 0.0);
    for (int e = 0; e < p * p; ++e) LH[e] = 0.0;
    if (!hzero) {
      for (int c = 0; c < p; ++c) {
        double d = gH[c * p + c];
        for (int k = 0; k < c; ++k) d -= LH[c * p + k] * LH[c * p + k];
        if (!(d > 0.0)) bad = 2;
        d = sqrt(d);
        LH[c * p + c] = d;
        for (int i = c + 1; i < p; ++i) {
          double s = gH[i * p + c];
          for (int k = 0; k < c; ++k) s -= LH[i * p + k] * LH[c * p + k];
          LH[i * p + c] = s / d;
        }
      }
    }
    *st = bad;
    if (info && blockIdx.x == 0) info[b] = bad;
  }
  __syncthreads();
  if (j >= n_sims) return;
  const long traj = b * n_sims + j;
  double* so = states + traj * (long)n * m;
  double* oo = obs + traj * (long)n * p;
  const double* zsj = zs + traj * (long)n * r;
  const double* zoj = zo + traj * (long)n * p;
  constexpr int MMAX = 64;
  double x[MMAX], xn[MMAX];
  for (int i = 0; i < m; ++i) x[i] = x0 ? x0[b * (long)m + i] : 0.0;
  for (int i = 0; i < m; ++i) so[i] = x[i];
  for (int q = 0; q < p; ++q) {
    double s = 0.0;
    if (x0)
      for (int k = 0; k < m; ++k) s += Zl[q * m + k] * x[k];
    oo[q] = s;
  }
  for (int t = 1; t < n; ++t) {
    // observation from the PREVIOUS state (:58,:62) + noise @ chol(H)
    for (int q = 0; q < p; ++q) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Zl[q * m + k] * x[k];
      for (int k = q; k < p; ++k) s += zoj[(long)t * p + k] * LH[k * p + q];  // (z L)_q = sum_k z_k L[k][q]
      oo[(long)t * p + q] = s;
    }
    // state: T x + R (noise @ chol(Q))
    double eta[16];
    for (int l = 0; l < r; ++l) {
      double s = 0.0;
      for (int k = l; k < r; ++k) s += zsj[(long)t * r + k] * LQ[k * r + l];
      eta[l] = s;
    }
    for (int i = 0; i < m; ++i) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Tl[i * m + k] * x[k];
      for (int l = 0; l < r; ++l) s += Rl[i * r + l] * eta[l];
      xn[i] = s;
    }
    for (int i = 0; i < m; ++i) {
      x[i] = xn[i];
      so[(long)t * m + i] = xn[i];
    }
  }
}

// ---- reverse mode through the Riccati fixed point ---------------------------------------------------------------------
// Reference: SolveDiscreteARE.grad (utils/pytensor_scipy.py:33-60, Kao & Hennequin 2020) applied to
// X = solve_discrete_are(A = T', B = Z', Q_d = R Q R', R_d = H) (kalman_filter.py:388).  Per draw, in LDS:
//     F = Z X Z' + H            K = F^-1 Z X T'  (p x m)            A~ = T' - Z' K
//     S = A~ S A~' + sym(g_X)   (Smith doubling, as kf_lyap_lds_kernel)
//     W = X A~ S                A_bar = 2 W      B_bar = -2 W K'    Q_bar = S      R_bar = K S K'
// and the chain rule back to the filter's matrices:  g_T = A_bar', g_Z = B_bar', g_H = R_bar,
// g_R = S R (Q + Q'), g_Q = R' S R.  accumulate != 0: added to what the outputs hold (the direct gradient of the
// filter started from X).  info[b] = 1: the Lyapunov iteration did not converge, 2: F not positive definite.
__host__ __device__ inline int dare_bwd_region_doubles(int m, int p, int r) {
  const int mr = m * r > m * p ? m * r : m * p;
  return 7 * m * m + 3 * m * p + 3 * p * p + mr + 8;  // T, X, At, S, W, A2, V | Z, ZX, K | F, L, Fi | SR/KS
}

__global__ __launch_bounds__(KF_WAVE) void kf_dare_bwd_kernel(long B, int m, int p, int r, const double* Tm,
                                                              const double* Zm, const double* Rm, const double* Hm,
                                                              const double* Qm, const double* Xm, const double* gX,
                                                              double* g_T, double* g_Z, double* g_R, double* g_H,
                                                              double* g_Q, int accumulate, int32_t* info, int h_diag_only) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const int mm = m * m, mp = m * p, pp = p * p, mr = m * r;
  double* Tl = smem;
  double* X = Tl + mm;
  double* At = X + mm;
  double* S = At + mm;
  double* W = S + mm;
  double* A2 = W + mm;
  double* V = A2 + mm;
  double* Zl = V + mm;
  double* ZX = Zl + mp;
  double* K = ZX + mp;
  double* F = K + mp;
  double* L = F + pp;
  double* Fi = L + pp;
  double* SR = Fi + pp;  // m x max(r, p)
  int status = 0;

  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    Tl[e] = Tm[b * mm + e];
    X[e] = Xm[b * mm + e];
    S[e] = 0.5 * (gX[b * mm + e] + gX[b * mm + j * m + i]);
  }
  for (int e = lane; e < mp; e += KF_WAVE) Zl[e] = Zm[b * mp + e];
  __syncthreads();
  // ZX = Z X
  for (int e = lane; e < mp; e += KF_WAVE) {
    const int q = e / m, j = e - q * m;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += Zl[q * m + k] * X[k * m + j];
    ZX[e] = s;
  }
  __syncthreads();
  // F = sym(ZX Z' + H) ; K <- ZX T' (before the solve)
  for (int e = lane; e < pp; e += KF_WAVE) {
    const int q1 = e / p, q2 = e - q1 * p;
    double s = 0.5 * (Hm[b * pp + e] + Hm[b * pp + q2 * p + q1]);
    for (int j = 0; j < m; ++j) s += 0.5 * (ZX[q1 * m + j] * Zl[q2 * m + j] + ZX[q2 * m + j] * Zl[q1 * m + j]);
    F[e] = s;
  }
  for (int e = lane; e < mp; e += KF_WAVE) {
    const int q = e / m, j = e - q * m;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += ZX[q * m + k] * Tl[j * m + k];
    K[e] = s;
  }
  __syncthreads();
  // Cholesky of F by lane 0, then K <- F^-1 K column by column (lane j owns column j of the p x m right-hand side)
  if (lane == 0) {
    for (int j = 0; j < p; ++j) {
      double s = F[j * p + j];
      for (int k = 0; k < j; ++k) s -= L[j * p + k] * L[j * p + k];
      if (!(s > 0.0)) status = 2;
      const double dj = sqrt(s);
      L[j * p + j] = dj;
      for (int i = j + 1; i < p; ++i) {
        double t = F[i * p + j];
        for (int k = 0; k < j; ++k) t -= L[i * p + k] * L[j * p + k];
        L[i * p + j] = t / dj;
      }
    }
  }
  __syncthreads();
  for (int j = lane; j < m; j += KF_WAVE) {
    for (int i = 0; i < p; ++i) {
      double t = K[i * m + j];
      for (int k = 0; k < i; ++k) t -= L[i * p + k] * K[k * m + j];
      K[i * m + j] = t / L[i * p + i];
    }
    for (int i = p - 1; i >= 0; --i) {
      double t = K[i * m + j];
      for (int k = i + 1; k < p; ++k) t -= L[k * p + i] * K[k * m + j];
      K[i * m + j] = t / L[i * p + i];
    }
  }
  __syncthreads();
  // At = T' - Z' K ; V = X At (needs the original At: the doubling below overwrites its working copy A2)
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    double s = Tl[j * m + i];
    for (int q = 0; q < p; ++q) s -= Zl[q * m + i] * K[q * m + j];
    At[e] = s;
  }
  __syncthreads();
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += X[i * m + k] * At[k * m + j];
    V[e] = s;
  }
  __syncthreads();
  // S = sum_j At^j D At'^j by doubling: S <- S + A S A', A <- A A  (Tl is free from here on: it holds A A)
  {
    int st = 1;
    for (int it = 0; it < 48; ++it) {
      for (int e = lane; e < mm; e += KF_WAVE) {
        const int i = e / m, j = e - i * m;
        double s = 0.0, s2 = 0.0;
        for (int k = 0; k < m; ++k) {
          const double aik = At[i * m + k];
          s += aik * S[k * m + j];
          s2 += aik * At[k * m + j];
        }
        W[e] = s;
        A2[e] = s2;
      }
      __syncthreads();
      double dmax = 0.0, pmax = 0.0;
      for (int e = lane; e < mm; e += KF_WAVE) {
        const int i = e / m, j = e - i * m;
        double sij = 0.0, sji = 0.0;
        for (int k = 0; k < m; ++k) {
          sij += W[i * m + k] * At[j * m + k];
          sji += W[j * m + k] * At[i * m + k];
        }
        const double inc = 0.5 * (sij + sji);
        const double pn = S[e] + inc;
        dmax = fmax(dmax, fabs(inc));
        pmax = fmax(pmax, fabs(pn));
        S[e] = pn;
      }
      dmax = wave_max(dmax);
      pmax = wave_max(pmax);
      __syncthreads();
      for (int e = lane; e < mm; e += KF_WAVE) At[e] = A2[e];
      __syncthreads();
      if (!(dmax > 1e-17 * pmax)) {
        if (pmax < 1.0e300 && dmax == dmax) st = 0;
        break;
      }
    }
    if (st && !status) status = 1;
  }
  // W = V S  (= X At S)
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += V[i * m + k] * S[k * m + j];
    W[e] = s;
  }
  __syncthreads();
  const double keep = accumulate ? 1.0 : 0.0;
  if (g_T)
    for (int e = lane; e < mm; e += KF_WAVE) {
      const int i = e / m, j = e - i * m;
      double* o = g_T + b * mm + e;
      *o = (accumulate ? *o : 0.0) + 2.0 * W[j * m + i];
    }
  if (g_Z)
    for (int e = lane; e < mp; e += KF_WAVE) {
      const int q = e / m, j = e - q * m;
      double s = 0.0;
      for (int i = 0; i < m; ++i) s += K[q * m + i] * W[j * m + i];
      double* o = g_Z + b * mp + e;
      *o = (accumulate ? *o : 0.0) - 2.0 * s;
    }
  if (g_H) {
    for (int e = lane; e < mp; e += KF_WAVE) {  // KS = K S  (p x m)
      const int q = e / m, j = e - q * m;
      double s = 0.0;
      for (int i = 0; i < m; ++i) s += K[q * m + i] * S[i * m + j];
      SR[e] = s;
    }
    __syncthreads();
    for (int e = lane; e < pp; e += KF_WAVE) {
      const int q1 = e / p, q2 = e - q1 * p;
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += SR[q1 * m + j] * K[q2 * m + j];
      double* o = g_H + b * pp + e;
      *o = (accumulate ? *o : 0.0) + ((h_diag_only && q1 != q2) ? 0.0 : s);
    }
    __syncthreads();
  }
  if (g_R || g_Q) {
    const double* gR = Rm + b * (long)mr;
    const double* gQ = Qm + b * (long)r * r;
    for (int e = lane; e < mr; e += KF_WAVE) {  // SR = S R  (m x r)
      const int i = e / r, l = e - i * r;
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += S[i * m + j] * gR[j * r + l];
      SR[e] = s;
    }
    __syncthreads();
    if (g_R)
      for (int e = lane; e < mr; e += KF_WAVE) {
        const int i = e / r, l = e - i * r;
        double s = 0.0;
        for (int k = 0; k < r; ++k) s += SR[i * r + k] * (gQ[k * r + l] + gQ[l * r + k]);
        double* o = g_R + b * (long)mr + e;
        *o = (accumulate ? *o : 0.0) + s;
      }
    if (g_Q)
      for (int e = lane; e < r * r; e += KF_WAVE) {
        const int k = e / r, l = e - k * r;
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += gR[i * r + k] * SR[i * r + l];
        double* o = g_Q + b * (long)r * r + e;
        *o = (accumulate ? *o : 0.0) + s;
      }
  }
  (void)keep;
  if (info) {
    status = __shfl(status, 0, KF_WAVE);  // the Cholesky verdict lives in lane 0
    if (lane == 0) info[b] = status;
  }
}

}  // namespace

int kf_launch_dare(kf_handle* h, long B, int m, int p, int r, const double* T, const double* Z, const double* R,
                   const double* H, const double* Q, const double* Pinit, double* P, int32_t* info, int max_iter,
                   double tol, hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)dare_region_doubles(m, p);
  if (lds > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "dare: m=%d p=%d needs %zu B of LDS per workgroup (> %d)", m, p, lds, KF_LDS_BYTES_MAX);
    return KF_ERR_UNSUPPORTED;
  }
  KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_dare_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kf_dare_kernel, dim3((unsigned)B), dim3(KF_WAVE), lds, stream, B, m, p, r, T, Z, R, H, Q, Pinit,
                     P, info, max_iter, tol);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

int kf_launch_dare_backward(kf_handle* h, long B, int m, int p, int r, const double* T, const double* Z, const double* R,
                            const double* H, const double* Q, const double* P, const double* g_P, double* g_T,
                            double* g_Z, double* g_R, double* g_H, double* g_Q, int accumulate, int32_t* info,
                            hipStream_t stream, int h_diag_only) {
  const size_t lds = sizeof(double) * (size_t)dare_bwd_region_doubles(m, p, r);
  if (lds > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "dare_backward: m=%d p=%d r=%d needs %zu B of LDS per workgroup (> %d)", m, p, r, lds,
               KF_LDS_BYTES_MAX);
    return KF_ERR_UNSUPPORTED;
  }
  KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_dare_bwd_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kf_dare_bwd_kernel, dim3((unsigned)B), dim3(KF_WAVE), lds, stream, B, m, p, r, T, Z, R, H, Q, P,
                     g_P, g_T, g_Z, g_R, g_H, g_Q, accumulate, info, h_diag_only);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

int kf_launch_simulate_statespace(kf_handle* h, long B, int n_sims, int n, int p, int m, int r, const double* T,
                                  const double* Z, const double* R, const double* H, const double* Q,
                                  const double* x0, const double* zs, const double* zo, double* states, double* obs,
                                  int32_t* info, hipStream_t stream) {
  if (m > 64 || r > 16) {
    KF_SET_ERR(h, "simulate_statespace: m=%d (<= 64) r=%d (<= 16) out of range", m, r);
    return KF_ERR_UNSUPPORTED;
  }
  const size_t lds = sizeof(double) * (size_t)(m * m + p * m + m * r + r * r + p * p + 2);
  if (lds > 64 * 1024) {
    KF_SET_ERR(h, "simulate_statespace: matrices need %zu B of LDS", lds);
    return KF_ERR_UNSUPPORTED;
  }
  if (B > 65535) {  // the draw axis sits in gridDim.y
    KF_SET_ERR(h, "simulate_statespace: B=%ld > 65535 draws per call (chunk the batch)", B);
    return KF_ERR_UNSUPPORTED;
  }
  dim3 grid((unsigned)((n_sims + KF_WAVE - 1) / KF_WAVE), (unsigned)B);
  hipLaunchKernelGGL(kf_simulate_ss_kernel, grid, dim3(KF_WAVE), lds, stream, B, n_sims, n, p, m, r, T, Z, R, H, Q, x0,
                     zs, zo, states, obs, info);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
