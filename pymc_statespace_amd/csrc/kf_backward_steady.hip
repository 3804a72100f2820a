// kf_backward_steady.hip -- the STEADY-STATE ADJOINT: gradient of the log-likelihood over the steps a forward kernel ran in its
// steady-state phase (static model, scalar observations: p == 1, the Univariate filter, or KF_FLAG_H_DIAGONAL; m <= 16, p <= 8).
//
// The reference's gradient is autodiff of the scan (core/statespace.py:174): the adjoint of the FULL recursion at every step,
// three m^3 products per step and the update's adjoint per observation (kf_backward_mfma.hip: ~6000 cycles per draw-step at
// C4).  But once the covariance recursion has converged (step ts: P_t = P to rounding for t >= ts; the forward kernel's own
// test, kf_forward_b4.hip) the log-likelihood of the remaining steps depends on the parameters
//   * directly and through the CONSTANT gain  K = P Z' W,  W = F^-1,  F = Z P Z' + H  -- a backward VECTOR recursion
//         v_t = y~_t - Z a_t           mu_t = T' lam_{t+1}           vb_t = -gl_t W v_t + K' mu_t
//         lam_t = mu_t - Z' vb_t       ( = (T (I - K Z))' lam_{t+1} + gl_t Z' W v_t )
//     and sums of outer products:  T_b += lam_{t+1} (a_t + K v_t)',  K_b += mu_t v_t',  Z_b -= vb_t a_t',  c_b += lam_{t+1},
//     d_b += dsign vb_t,  W_b += gl_t (F - v_t v_t') / 2   -- no m^3 work per step, ~100 instructions per draw-step;
//   * through P(theta), the fixpoint of the Riccati map: ONE Riccati adjoint per draw (kf_dare_backward: the reference's
//     SolveDiscreteARE.grad, utils/pytensor_scipy.py:33-60) applied to  P_b = M_b Z + Z' F_b Z  with  M_b = K_b W,
//     W_b += M' K_b,  F_b = -W sym(W_b) W;
// and the steps before ts get the exact adjoint (kf_backward_mfma.hip, per-draw step counts) started from
// (a_b, P_b) = (lam_ts, 0).  Checked against torch autograd of the recursion at every step: scripts/r03/steady_adjoint_proto.py
// (<= 1e-14 relative on every input, BASELINE workloads); the two differ by the derivative of the convergence error,
// O(ts rho^(2 ts)).  The symmetric parts of the gradients with respect to symmetric inputs agree; H gets its diagonal only,
// as in every scalar-observation adjoint here.
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int SA_M = 16, SA_P = 8;

// LDS map (doubles), PS = the stride of the p axis (1: scalar observation -- the p == 1 block kernels' adjoint; 8: up to eight).
// Persistent pieces, the loop's two vectors, and ONE scratch region shared by the setup (T, M, T K, W Z, the inverse's work space)
// and the epilogue (K_b, M_b, W_b, F_b, M, F_b Z, F_b Z P).  Round 3's layout took 14 KB per wavefront: 11 wavefronts per CU, so a
// 4096-draw launch ran as 1.45 rounds (headline: 1.20 ms); PS = 1 takes 5.6 KB, PS = 8 12.2 KB.
template <int PS>
struct SaLds {
  static constexpr int OFF_P = 0;                          // P            m x m  (stride 16)
  static constexpr int OFF_Z = 256;                        // Z            p x 16
  static constexpr int OFF_K = OFF_Z + 16 * PS;            // K            16 x PS (m x p)
  static constexpr int OFF_W = OFF_K + 16 * PS;            // W = F^-1     PS x PS
  static constexpr int OFF_F = OFF_W + PS * PS;            // F            PS x PS
  static constexpr int OFF_V = (OFF_F + PS * PS + 1) & ~1; // x = lam 16 | a 16 | y~ PS (padded to 48)
  static constexpr int OFF_O = OFF_V + 48;                 // stage outputs: lam_t 16 | af 16 | mu 16 | v 8 | vb 8
  static constexpr int OFF_S = OFF_O + 64;                 // scratch of the setup / the epilogue
  // setup:    T 256 | M 16 PS | T K 16 PS | W Z PS 16 | Gauss-Jordan work space 2 x 64 (PS > 1)
  static constexpr int S_T = OFF_S, S_M = S_T + 256, S_TK = S_M + 16 * PS, S_WZ = S_TK + 16 * PS, S_GJ = S_WZ + 16 * PS;
  static constexpr int SETUP_END = S_GJ + (PS > 1 ? 128 : 0);
  // epilogue: K_b 16 PS | M_b 16 PS | W_b PS PS | F_b PS PS | M 16 PS | F_b Z PS 16 | F_b Z P PS 16
  static constexpr int E_KB = OFF_S, E_MB = E_KB + 16 * PS, E_WB = E_MB + 16 * PS, E_FB = E_WB + PS * PS, E_M = E_FB + PS * PS,
                       E_FZ = E_M + 16 * PS, E_FZP = E_FZ + 16 * PS;
  static constexpr int EPI_END = E_FZP + 16 * PS;
  static constexpr int TOTAL = SETUP_END > EPI_END ? SETUP_END : EPI_END;
  static constexpr int ZERO_END = OFF_S;                   // zeroed before the setup (rows / columns beyond m, p stay zero)
};

// MS: k_states rounded up (12 / 16) -- the x vector a lane multiplies is lam (MS) | a (MS) | y~ (PS): at m <= 12 its coefficients take
// 26 register pairs instead of 34 and the kernel fits four wavefronts per SIMD without spilling (PS == 1)
template <int PS, int MS>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(PS == 1 ? 4 : 3))) void kf_bwd_steady_kernel(const KfBwdArgs A, const int32_t* __restrict__ ts_all,
                                                                 double* __restrict__ out_ab, double* __restrict__ out_Pb,
                                                                 double* __restrict__ out_P, const double* __restrict__ P_conv) {
  using L = SaLds<PS>;
  __shared__ __attribute__((aligned(16))) double smem[L::TOTAL];
  const int lane = threadIdx.x;
  const long b = A.list ? (long)A.list[blockIdx.x] : (long)blockIdx.x;
  const int n = A.T, m = A.m, p = A.p, mm = m * m, r_ = A.r;
  const int ts = ts_all[b] < n ? ts_all[b] : n;
  double* sP = smem + L::OFF_P;
  double* sZ = smem + L::OFF_Z;
  double* sK = smem + L::OFF_K;
  double* sW = smem + L::OFF_W;
  double* sF = smem + L::OFF_F;
  double* XV = smem + L::OFF_V;    // x: lam (MS) | a (MS) | y~ (PS)
  double* OUT = smem + L::OFF_O;   // lam_t (16) | af (16) | mu (16) | v (8) | vb (8)
  double* sT = smem + L::S_T;
  double* sM = smem + L::S_M;
  double* TK = smem + L::S_TK;
  double* WZ = smem + L::S_WZ;

  const double* gpa = A.pred_a + b * (long)(A.tape_T + 1) * m;
  // the converged predicted covariance: slot ts of the predicted tape, or (P_conv: the p == 1 block kernel keeps no compact
  // covariance tape) the [B][m][m] array the forward kernel left
  const double* gpP = P_conv ? P_conv + b * (long)mm : A.pred_P + b * (long)(A.tape_T + 1) * mm;
  const double* gT = A.Tm + b * (long)mm;
  const double* gZ = A.Z + b * (long)p * m;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool ypd = (A.flags & KF_FLAG_Y_PER_DRAW) != 0;
  const double* ybase = A.y + (ypd ? b * (long)A.y_T * p : 0L);
  const double g_all = A.g_logp ? A.g_logp[b] : 0.0;

  for (int e = lane; e < L::TOTAL; e += KF_WAVE) smem[e] = 0.0;
  wave_lds_fence();
  const int tsc = P_conv ? 0 : (ts < n ? ts : n);  // (a draw that never converged: P of slot n, nothing to sweep)
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    const double pij = 0.5 * (gpP[(long)tsc * mm + e] + gpP[(long)tsc * mm + j * m + i]);
    sP[i * 16 + j] = pij;
    sT[i * 16 + j] = gT[e];
    out_P[b * (long)mm + e] = pij;
  }
  for (int e = lane; e < p * m; e += KF_WAVE) sZ[(e / m) * 16 + e % m] = gZ[e];
  wave_lds_fence();
  // M = P Z' (m x p), F = Z M + diag(H)
  for (int e = lane; e < m * p; e += KF_WAVE) {
    const int i = e / p, q = e - i * p;
    double s = 0.0;
    for (int j = 0; j < m; ++j) s = fma(sP[i * 16 + j], sZ[q * 16 + j], s);
    sM[i * PS + q] = s;
  }
  wave_lds_fence();
  for (int e = lane; e < p * p; e += KF_WAVE) {
    const int q = e / p, r2 = e - q * p;
    double s = (q == r2) ? A.H[b * (long)p * p + q * p + q] : 0.0;
    for (int i = 0; i < m; ++i) s = fma(sZ[q * 16 + i], sM[i * PS + r2], s);
    sF[q * PS + r2] = s;
  }
  wave_lds_fence();
  // W = F^-1
  if constexpr (PS == 1) {
    if (lane == 0) sW[0] = 1.0 / sF[0];
    wave_lds_fence();
  } else {
    // Gauss-Jordan on (F | I), lane = entry of the 8 x 8 work matrices, pivots in order (F is SPD)
    double* X1 = smem + L::S_GJ;  // working copy of F
    double* X2 = X1 + 64;         // inverse being built
    {
      const int q = lane >> 3, r2 = lane & 7;
      X1[lane] = (q < p && r2 < p) ? 0.5 * (sF[q * PS + r2] + sF[r2 * PS + q]) : (q == r2 ? 1.0 : 0.0);
      X2[lane] = (q == r2) ? 1.0 : 0.0;
    }
    wave_lds_fence();
    for (int k = 0; k < p; ++k) {
      const int q = lane >> 3, r2 = lane & 7;
      const double piv = X1[k * 8 + k];
      const double ip = 1.0 / piv;
      const double fk = X1[q * 8 + k];            // column k of row q
      const double rowk1 = X1[k * 8 + r2] * ip;   // normalised pivot row
      const double rowk2 = X2[k * 8 + r2] * ip;
      wave_lds_fence();
      if (q == k) {
        X1[lane] = rowk1;
        X2[lane] = rowk2;
      } else {
        X1[lane] = fma(-fk, rowk1, X1[lane]);
        X2[lane] = fma(-fk, rowk2, X2[lane]);
      }
      wave_lds_fence();
    }
    sW[lane] = ((lane >> 3) < p && (lane & 7) < p) ? 0.5 * (X2[lane] + X2[(lane & 7) * 8 + (lane >> 3)]) : 0.0;
    wave_lds_fence();
  }
  // K = M W
  for (int e = lane; e < m * p; e += KF_WAVE) {
    const int i = e / p, q = e - i * p;
    double s = 0.0;
    for (int r2 = 0; r2 < p; ++r2) s = fma(sM[i * PS + r2], sW[r2 * PS + q], s);
    sK[i * PS + q] = s;
  }
  wave_lds_fence();

  // ---- one step = ONE matrix-vector stage.  Everything a step needs is linear in x = (lam_{t+1} | a_t | y~_t):
  //   lanes  0..15  lam_t[j] = sum_i Abar[i][j] lam_i - gl sum_l (Z' W Z)[j][l] a_l + gl sum_q (Z' W)[j][q] y~_q     (Abar = T - T K Z)
  //   lanes 16..31  af[j]    = sum_l (I - K Z)[j][l] a_l + sum_q K[j][q] y~_q
  //   lanes 32..47  mu[j]    = sum_i T[i][j] lam_i
  //   lanes 48..55  v[q]     = y~_q - sum_l Z[q][l] a_l
  //   lanes 56..63  vb[q]    = sum_i (T K)[i][q] lam_i + gl sum_l (W Z)[q][l] a_l - gl sum_r W[q][r] y~_r
  // each lane holds its 32 + PS coefficients in registers and reads x from LDS: one LDS round trip per step.
  // (gl is the draw's cotangent of logp: kf_filter_value_and_grad has no per-step cotangents.)
  for (int e = lane; e < 16 * PS; e += KF_WAVE) {  // T K
    const int i = e / PS, q = e - i * PS;
    double s2 = 0.0;
    for (int j = 0; j < 16; ++j) s2 = fma(sT[i * 16 + j], sK[j * PS + q], s2);
    TK[e] = s2;
  }
  for (int e = lane; e < 16 * PS; e += KF_WAVE) {  // W Z
    const int q = e >> 4, l = e & 15;
    double s2 = 0.0;
    for (int r2 = 0; r2 < PS; ++r2) s2 = fma(sW[q * PS + r2], sZ[r2 * 16 + l], s2);
    WZ[e] = s2;
  }
  wave_lds_fence();
  const double gl = g_all;
  constexpr int NX = 2 * MS + (PS == 1 ? 2 : PS);   // x entries a lane multiplies (padded to whole 16-byte reads)
  double cf[NX];
  {
    const int j = lane & 15, q = lane & 7;
    const bool qok = q < PS;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      double v = 0.0;
      const int l = k - MS;        // index into a (MS <= k < 2 MS)
      const int yq = k - 2 * MS;   // index into y~ (k >= 2 MS)
      if (lane < 16) {
        if (k < MS) {  // Abar[k][j] = T[k][j] - sum_q TK[k][q] Z[q][j]
          v = sT[k * 16 + j];
          for (int q2 = 0; q2 < PS; ++q2) v = fma(-TK[k * PS + q2], sZ[q2 * 16 + j], v);
        } else if (k < 2 * MS) {  // -gl (Z' W Z)[j][l]
          for (int q2 = 0; q2 < PS; ++q2) v = fma(sZ[q2 * 16 + j], WZ[q2 * 16 + l], v);
          v *= -gl;
        } else if (yq < PS) {  // gl (Z' W)[j][q]
          for (int r2 = 0; r2 < PS; ++r2) v = fma(sZ[r2 * 16 + j], sW[r2 * PS + yq], v);
          v *= gl;
        }
      } else if (lane < 32) {
        if (k >= MS && k < 2 * MS) {  // (I - K Z)[j][l]
          v = (l == j) ? 1.0 : 0.0;
          for (int q2 = 0; q2 < PS; ++q2) v = fma(-sK[j * PS + q2], sZ[q2 * 16 + l], v);
        } else if (k >= 2 * MS && yq < PS) {
          v = sK[j * PS + yq];
        }
      } else if (lane < 48) {
        if (k < MS) v = sT[k * 16 + j];
      } else if (lane < 56) {
        if (qok) {
          if (k >= MS && k < 2 * MS) v = -sZ[q * 16 + l];
          else if (k >= 2 * MS && yq < PS) v = (yq == q) ? 1.0 : 0.0;
        }
      } else if (qok) {
        if (k < MS) v = TK[k * PS + q];
        else if (k < 2 * MS) v = gl * WZ[q * 16 + l];
        else if (yq < PS) v = -gl * sW[q * PS + yq];
      }
      cf[k] = v;
    }
  }
  wave_lds_fence();
  for (int e = lane; e < 48; e += KF_WAVE) XV[e] = 0.0;

  // ---- accumulators (registers; lanes over matrix entries) ----------------------------------------------------------------------
  const int tj = lane & 15, ti4 = lane >> 4;   // T_b: column tj, rows ti4 + 4 s
  const int kq = lane & 7, ki = lane >> 3;     // K_b: column kq, rows ki + 8 s
  const int zj = lane & 15, zq = lane >> 4;    // Z_b: column zj, rows zq + 4 s
  const int vq = lane >> 3, vr = lane & 7;     // sum gl v v'
  double Tb[4] = {0.0, 0.0, 0.0, 0.0}, Kb[2] = {0.0, 0.0}, Zb[2] = {0.0, 0.0}, VV = 0.0, cb = 0.0, db = 0.0, GLs = 0.0;
  double GYv = 0.0;   // (g_yy requested: the dense-H gradient by whitening, kf_api.hip) sum_t vb_t y~_t' at (vq, vr)
  const bool want_yy = A.g_yy != nullptr;
  // lanes 16 .. 16+m-1 fetch a_t, lanes 32 .. 32+p-1 fetch y~_t, one step ahead
  const bool is_a = lane >= 16 && lane < 16 + m, is_y = lane >= 32 && lane < 32 + p;
  const double dq = (is_y && A.d) ? A.d[b * (long)p + (lane - 32)] : 0.0;
  auto fetch = [&](int t) -> double {
    if (is_a) return gpa[(long)t * m + (lane - 16)];
    if (is_y) return ybase[(long)t * p + (lane - 32)] + dsign * dq;
    return 0.0;
  };
  double nxt = ts < n ? fetch(n - 1) : 0.0;
  for (int t = n - 1; t >= ts; --t) {
    if (is_a) XV[MS + lane - 16] = nxt;
    if (is_y) XV[2 * MS + lane - 32] = nxt;
    wave_lds_fence();
    if (t > ts) nxt = fetch(t - 1);
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int k = 0; k < NX; k += 2) {
      const double2_t x01 = *reinterpret_cast<const double2_t*>(XV + k);
      acc0 = fma(cf[k], x01.x, acc0);
      acc1 = fma(cf[k + 1], x01.y, acc1);
    }
    const double o = acc0 + acc1;
    OUT[lane] = o;
    wave_lds_fence();
    // accumulate with lam_{t+1} (still in x), then lam_t replaces it
    {
      const double afj = OUT[16 + tj];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)   // (rows >= MS read past lam: finite values, the rows are never stored)
        Tb[s2] = fma(XV[ti4 + 4 * s2], afj, Tb[s2]);
      const double vqv = OUT[48 + kq];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) Kb[s2] = fma(OUT[32 + ki + 8 * s2], vqv, Kb[s2]);
      const double azj = XV[MS + (zj < MS ? zj : 0)];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) Zb[s2] = fma(-OUT[56 + zq + 4 * s2], azj, Zb[s2]);
      VV = fma(gl * OUT[48 + vq], OUT[48 + vr], VV);
      if (want_yy) GYv = fma(OUT[56 + vq], XV[2 * MS + vr], GYv);
      if (lane < 16) cb += XV[lane];
      if (lane >= 56) db = fma(dsign, o, db);
      GLs += gl;
    }
    if (lane < MS) XV[lane] = o;
  }
  wave_lds_fence();
  {
    const double dbv = __shfl(db, 56 + (lane & 7));
    db = dbv;
  }
  const double lam_fin = (lane < MS) ? XV[lane] : 0.0;
  wave_lds_fence();

  // ---- epilogue: (K_b, W_b) -> (P_b, Z_b, H_b); its buffers take the setup's scratch -------------------------------------------------
  double* eKb = smem + L::E_KB;    // 16 x PS
  double* eMb = smem + L::E_MB;    // 16 x PS
  double* eWb = smem + L::E_WB;    // PS x PS
  double* eFb = smem + L::E_FB;    // PS x PS
  double* eM = smem + L::E_M;      // 16 x PS: M = P Z'
  double* eFZ = smem + L::E_FZ;    // PS x 16: F_b Z
  double* eFZP = smem + L::E_FZP;  // PS x 16: F_b Z P
  const bool vok = vq < PS && vr < PS;
#pragma unroll
  for (int s = 0; s < 2; ++s)
    if (kq < PS) eKb[(ki + 8 * s) * PS + kq] = Kb[s];
  wave_lds_fence();
  for (int e = lane; e < m * p; e += KF_WAVE) {
    const int i = e / p, q = e - i * p;
    double sm = 0.0;
    for (int j = 0; j < m; ++j) sm = fma(sP[i * 16 + j], sZ[q * 16 + j], sm);
    eM[i * PS + q] = sm;
  }
  wave_lds_fence();
  if (vok) {  // W_b = (GL F - VV) / 2 + M' K_b, symmetrised
    double s = 0.5 * (GLs * sF[vq * PS + vr] - VV);
    for (int i = 0; i < m; ++i) s = fma(eM[i * PS + vq], eKb[i * PS + vr], s);
    eWb[vq * PS + vr] = (vq < p && vr < p) ? s : 0.0;
  }
  for (int e = lane; e < m * p; e += KF_WAVE) {  // M_b = K_b W
    const int i = e / p, q = e - i * p;
    double s = 0.0;
    for (int r2 = 0; r2 < p; ++r2) s = fma(eKb[i * PS + r2], sW[r2 * PS + q], s);
    eMb[i * PS + q] = s;
  }
  wave_lds_fence();
  if (vok) {  // F_b = -W sym(W_b) W
    double s = 0.0;
    for (int r2 = 0; r2 < p; ++r2) {
      double t2 = 0.0;
      for (int r3 = 0; r3 < p; ++r3) t2 = fma(0.5 * (eWb[r2 * PS + r3] + eWb[r3 * PS + r2]), sW[r3 * PS + vr], t2);
      s = fma(sW[vq * PS + r2], t2, s);
    }
    eFb[vq * PS + vr] = -s;
  }
  wave_lds_fence();
  for (int e = lane; e < p * m; e += KF_WAVE) {  // F_b Z
    const int q = e / m, j = e - q * m;
    double s = 0.0;
    for (int r2 = 0; r2 < p; ++r2) s = fma(eFb[q * PS + r2], sZ[r2 * 16 + j], s);
    eFZ[q * 16 + j] = s;
  }
  wave_lds_fence();
  for (int e = lane; e < p * m; e += KF_WAVE) {  // F_b Z P
    const int q = e / m, j = e - q * m;
    double s = 0.0;
    for (int l = 0; l < m; ++l) s = fma(eFZ[q * 16 + l], sP[l * 16 + j], s);
    eFZP[q * 16 + j] = s;
  }
  wave_lds_fence();
  // ---- outputs -----------------------------------------------------------------------------------------------------------------------
  for (int e = lane; e < mm; e += KF_WAVE) {  // P_b = M_b Z + Z' F_b Z
    const int i = e / m, j = e - i * m;
    double s = 0.0;
    for (int q = 0; q < p; ++q) s = fma(eMb[i * PS + q], sZ[q * 16 + j], fma(sZ[q * 16 + i], eFZ[q * 16 + j], s));
    out_Pb[b * (long)mm + e] = s;
    if (A.g_P0) A.g_P0[b * (long)mm + e] = 0.0;
  }
  if (A.g_T) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int i = ti4 + 4 * s;
      if (i < m && tj < m) A.g_T[b * (long)mm + i * m + tj] = Tb[s];
    }
  }
  if (A.g_Z) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = zq + 4 * s;
      if (q < p && zj < m) {
        double v = Zb[s] + 2.0 * eFZP[q * 16 + zj];  // + M_b' P + 2 F_b Z P
        for (int i = 0; i < m; ++i) v = fma(eMb[i * PS + q], sP[i * 16 + zj], v);
        A.g_Z[b * (long)p * m + q * m + zj] = v;
      }
    }
  }
  if (A.g_H)
    for (int e = lane; e < p * p; e += KF_WAVE) A.g_H[b * (long)p * p + e] = (e / p == e % p) ? eFb[(e / p) * PS + e % p] : 0.0;
  if (lane < m) {
    if (A.g_c) A.g_c[b * (long)m + lane] = cb;
    if (A.g_a0) A.g_a0[b * (long)m + lane] = 0.0;
    out_ab[b * (long)m + lane] = lam_fin;
  }
  if (lane < p && A.g_d) A.g_d[b * (long)p + lane] = db;
  if (want_yy && vq < p && vr < p) A.g_yy[b * (long)p * p + vq * p + vr] = GYv;
  if (A.g_R)
    for (int e = lane; e < m * r_; e += KF_WAVE) A.g_R[b * (long)m * r_ + e] = 0.0;
  if (A.g_Q)
    for (int e = lane; e < r_ * r_; e += KF_WAVE) A.g_Q[b * (long)r_ * r_ + e] = 0.0;
}


// ---- the same adjoint, TIME-PARALLEL (round 4; p == 1, m <= 10) ---------------------------------------------------------------------
// kf_bwd_steady_kernel walks the steps one at a time and is LDS-bound: every lane reads the whole x vector back per step (13 broadcast
// ds_read_b128 + 13 ds_read_b64 for the accumulations: headline 0.75 ms, C3 3.4 ms -- the largest kernel of C3's value-and-gradient).
// But behind the convergence point nothing is nonlinear:
//     v_t = y~_t - z a_t   (a_t is on the forward kernel's tape)          w_t = gl W v_t
//     lam_t = Abar' lam_{t+1} + z' w_t                                     (Abar = T - T K z)
// is an affine recursion in lam with DATA inputs -- a prefix scan over time, ONE STEP PER LANE, 64 steps at a time, latest step in lane 0:
// six Hillis-Steele levels, each a matrix-vector product with (Abar')^(2^k) read from LDS as broadcasts (the forward mean scan of
// kf_forward_b4_steady.hip, reversed) -- and every accumulation is a sum over time of an outer product of lam_{t+1} with (a_t | y~_t | 1):
//     S1 = sum lam_{t+1} a_t'     S2 = sum lam_{t+1} y~_t     S3 = sum lam_{t+1}     (lanes over the entries, 64 steps from two LDS row blocks)
//     Sva = sum v_t a_t           Sv = sum v_t                Svv = sum v_t^2        (per lane, one wave reduction per draw)
// from which, with sigma = S2 - S1 z' = sum lam_{t+1} v_t and tk = T K:
//     T_b = S1 + sigma K'    K_b = T' sigma    Z_b = gl W Sva' - tk' S1    c_b = S3    d_b = dsign (tk' S3 - gl W Sv)    VV = gl Svv
// -- the sums kf_bwd_steady_kernel accumulates step by step; the epilogue (K_b, W_b) -> (P_b, Z_b, H_b) is that kernel's at p = 1.
// ~35 instructions and ~14 LDS operations per draw-step instead of ~80 and 26.  Results agree with the sequential kernel to rounding
// (1e-13 of a gradient's largest entry; KF_VG_STEADY_SCAN=0 keeps the sequential kernel).
template <int M>
__global__ __launch_bounds__(KF_WAVE) void kf_bwd_steady_scan_kernel(const KfBwdArgs A, const int32_t* __restrict__ ts_all,
                                                                      double* __restrict__ out_ab, double* __restrict__ out_Pb,
                                                                      double* __restrict__ out_P, const double* __restrict__ P_conv) {
  constexpr int MM = M * M, NC = M + 2;          // columns of the right rows: a (M) | y~ | 1
  constexpr int NPAIR = M * NC;                  // entries of (S1 | S2 | S3)
  __shared__ __attribute__((aligned(16))) double sAP[6 * MM + 2];        // (Abar')^(2^k), k = 0 .. 5, row-major
  __shared__ __attribute__((aligned(16))) double rowsL[KF_WAVE * M];      // lam_{t+1} of the block's steps, one row per lane
  __shared__ __attribute__((aligned(16))) double rowsA[KF_WAVE * NC];     // (a_t | y~_t | 1)
  __shared__ __attribute__((aligned(16))) double sP[MM], sT[MM], sS[NPAIR + 2];
  __shared__ __attribute__((aligned(16))) double sV[10 * M + 8];
  const int lane = threadIdx.x;
  const long b = A.list ? (long)A.list[blockIdx.x] : (long)blockIdx.x;
  const int n = A.T, r_ = A.r;
  const int ts = ts_all[b] < n ? ts_all[b] : n;
  const double* gpa = A.pred_a + b * (long)(A.tape_T + 1) * M;
  // the converged covariance: handed over (block kernel), or slot ts of the compact predicted tape (lane kernel: every slot written)
  const double* gpP = P_conv ? P_conv + b * (long)MM : A.pred_P + (b * (long)(A.tape_T + 1) + (ts < n ? ts : n)) * MM;
  const double* gT = A.Tm + b * (long)MM;
  const double* gZ = A.Z + b * (long)M;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)A.y_T : 0L);
  const double d1 = A.d ? A.d[b] : 0.0;
  const double gl = A.g_logp ? A.g_logp[b] : 0.0;
  double* vz = sV;            // z
  double* vK = sV + M;        // K = P z' / F
  double* vM = sV + 2 * M;    // M = P z'
  double* vTK = sV + 3 * M;   // T K
  double* vX = sV + 4 * M;    // scratch vectors of the epilogue (6 M)
  for (int e = lane; e < MM; e += KF_WAVE) {
    const int i = e / M, j = e - i * M;
    const double pij = 0.5 * (gpP[e] + gpP[j * M + i]);
    sP[e] = pij;
    sT[e] = gT[e];
    out_P[b * (long)MM + e] = pij;
  }
  if (lane < M) vz[lane] = gZ[lane];
  wave_lds_fence();
  if (lane < M) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) s = fma(sP[lane * M + j], vz[j], s);
    vM[lane] = s;
  }
  wave_lds_fence();
  double F = A.H[b];
#pragma unroll
  for (int i = 0; i < M; ++i) F = fma(vz[i], vM[i], F);
  const double W = 1.0 / F;
  if (lane < M) vK[lane] = vM[lane] * W;
  wave_lds_fence();
  if (lane < M) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) s = fma(sT[lane * M + j], vK[j], s);
    vTK[lane] = s;
  }
  wave_lds_fence();
  for (int x = lane; x < MM; x += KF_WAVE) {   // (Abar')[i][l] = Abar[l][i] = T[l][i] - tk[l] z[i]
    const int i = x / M, l = x - i * M;
    sAP[x] = fma(-vTK[l], vz[i], sT[l * M + i]);
  }
  wave_lds_fence();
#pragma unroll 1
  for (int k = 1; k < 6; ++k) {
    const double* src = sAP + (k - 1) * MM;
    for (int x = lane; x < MM; x += KF_WAVE) {
      const int i = x / M, j = x - i * M;
      double v = 0.0;
#pragma unroll
      for (int l = 0; l < M; ++l) v = fma(src[i * M + l], src[l * M + j], v);
      sAP[k * MM + x] = v;
    }
    wave_lds_fence();
  }
  double z[M];
#pragma unroll
  for (int i = 0; i < M; ++i) z[i] = vz[i];
  // ---- forward: the means of the steps behind ts (A.fwd_logp; the block kernel stopped at ts) -------------------------------------------
  // a_{t+1} = Abar a_t + tk y~_t + c, 63 steps per pass: lane 0 carries a_tb, lane j >= 1 the forcing of step tb + j - 1, six levels with
  // Abar^(2^k) = the stored matrices read by columns (kf_b4_steady_scan_kernel's scan, kf_forward_b4_steady.hip); then every lane finishes
  // its own step's log-likelihood term and the rows leave through LDS as whole lines
  if (A.fwd_logp) {
    if (ts < n) {
      constexpr int STEPS = KF_WAVE - 1;
      double* wpa = A.fwd_pred_a + b * (long)(A.tape_T + 1) * M;
      double gk[M], cc[M], acur[M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        gk[i] = vTK[i];
        cc[i] = A.c ? A.c[b * (long)M + i] : 0.0;
        acur[i] = wpa[(long)ts * M + i];
      }
      const bool quirk = A.filter == KF_FILTER_UNIVARIATE && !(A.flags & KF_FLAG_MVN_CONST_EXACT) && F == 1.0;  // :497-500
      const double llconst = -0.5 * (KF_LOG_2PI + log(F));
      double acc = 0.0;
      for (int tb = ts; tb < n; tb += STEPS) {
        const int t = tb + lane;
        const int cnt = (n - tb) < STEPS ? (n - tb) : STEPS;
        const double yt = t < n ? ybase[t] + dsign * d1 : 0.0;
        const double yprev = __shfl_up(yt, 1);
        double x[M], xs[M];
#pragma unroll
        for (int i = 0; i < M; ++i) x[i] = lane == 0 ? acur[i] : fma(gk[i], yprev, cc[i]);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int s = 1 << k;
#pragma unroll
          for (int i = 0; i < M; ++i) {
            const double up = __shfl_up(x[i], s);
            xs[i] = lane >= s ? up : 0.0;
          }
#pragma unroll
          for (int l = 0; l < M; ++l)
#pragma unroll
            for (int i = 0; i < M; ++i) x[i] = fma(sAP[k * MM + l * M + i], xs[l], x[i]);
        }
        double za = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) za = fma(z[i], x[i], za);
        const double v = yt - za;
        const double quad = v * v * W;
        if (lane < cnt) acc += (quirk && quad == 0.0) ? 0.0 : llconst - 0.5 * quad;
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < M; ++i) rowsL[lane * M + i] = x[i];
        wave_lds_fence();
        // lanes 1 .. cnt hold a_{tb+1} .. a_{tb+cnt}
        for (int pos = lane; pos < cnt * M; pos += KF_WAVE) wpa[(long)(tb + 1) * M + pos] = rowsL[M + pos];
#pragma unroll
        for (int i = 0; i < M; ++i) acur[i] = __shfl(x[i], KF_WAVE - 1);
      }
#pragma unroll
      for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
      if (lane == 0) {
        A.fwd_logp[b] += acc;
        if (!(F > 0.0) && A.fwd_info && A.fwd_info[b] == 0) A.fwd_info[b] = ((ts + 1) << 8) | 1;
      }
    }
    __threadfence();   // the rows are read back below (other lanes, other cache lines of this CU's L1)
  }
  // this lane's entries of (S1 | S2 | S3): e = lane and lane + 64
  const int e0 = lane, e1 = lane + KF_WAVE;
  const bool p0 = e0 < NPAIR, p1 = e1 < NPAIR;
  const int i0 = p0 ? e0 / NC : 0, j0 = p0 ? e0 - i0 * NC : 0;
  const int i1 = p1 ? e1 / NC : 0, j1 = p1 ? e1 - i1 * NC : 0;
  double acc0 = 0.0, acc1 = 0.0;
  double sva[M], sv = 0.0, svv = 0.0, carry[M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    sva[i] = 0.0;
    carry[i] = 0.0;
  }
  for (int te = n; te > ts; te -= KF_WAVE) {
    const int tb = (te - KF_WAVE) > ts ? te - KF_WAVE : ts;
    const int cnt = te - tb;
    const bool act = lane < cnt;
    const int t = act ? te - 1 - lane : ts;
    double a[M];
#pragma unroll
    for (int i = 0; i < M; ++i) a[i] = act ? gpa[(long)t * M + i] : 0.0;
    const double yt = act ? ybase[t] + dsign * d1 : 0.0;
    double za = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) za = fma(z[i], a[i], za);
    const double v = yt - za;
    const double w = gl * W * v;
    double x[M], xs[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double u = z[i] * w;
      if (lane == 0) {   // the block behind this one enters through the latest step
#pragma unroll
        for (int l = 0; l < M; ++l) u = fma(sAP[i * M + l], carry[l], u);
      }
      x[i] = u;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int s = 1 << k;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const double up = __shfl_up(x[i], s);
        xs[i] = lane >= s ? up : 0.0;
      }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double vv = x[i];
#pragma unroll
        for (int l = 0; l < M; ++l) vv = fma(sAP[k * MM + i * M + l], xs[l], vv);
        x[i] = vv;
      }
    }
    // lane j holds lam of step te - 1 - j; lam_{t+1} of that step is the lane before it (the carry for lane 0)
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const double up = __shfl_up(x[i], 1);
      const double l1 = lane == 0 ? carry[i] : up;
      rowsL[lane * M + i] = act ? l1 : 0.0;
      rowsA[lane * NC + i] = a[i];
      sva[i] = fma(v, a[i], sva[i]);
    }
    rowsA[lane * NC + M] = yt;
    rowsA[lane * NC + M + 1] = act ? 1.0 : 0.0;
    sv += v;
    svv = fma(v, v, svv);
    wave_lds_fence();
#pragma unroll 8
    for (int s2 = 0; s2 < KF_WAVE; ++s2) {
      acc0 = fma(rowsL[s2 * M + i0], rowsA[s2 * NC + j0], acc0);
      acc1 = fma(rowsL[s2 * M + i1], rowsA[s2 * NC + j1], acc1);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) carry[i] = __shfl(x[i], cnt - 1);
  }
  // ---- sums -> LDS; wave reductions of the per-lane data sums ----------------------------------------------------------------------
  wave_lds_fence();
  if (p0) sS[e0] = acc0;
  if (p1) sS[e1] = acc1;
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    sv += __shfl_xor(sv, off);
    svv += __shfl_xor(svv, off);
#pragma unroll
    for (int i = 0; i < M; ++i) sva[i] += __shfl_xor(sva[i], off);
  }
  wave_lds_fence();
  // sigma = S2 - S1 z,  K_b = T' sigma
  double* vSig = vX;
  double* vKb = vX + M;
  double* vMb = vX + 2 * M;
  double* vFZ = vX + 3 * M;
  double* vFZP = vX + 4 * M;
  if (lane < M) {
    double s = sS[lane * NC + M];
#pragma unroll
    for (int j = 0; j < M; ++j) s = fma(-sS[lane * NC + j], z[j], s);
    vSig[lane] = s;
  }
  wave_lds_fence();
  if (lane < M) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < M; ++l) s = fma(sT[l * M + lane], vSig[l], s);
    vKb[lane] = s;
  }
  wave_lds_fence();
  const double GLs = gl * (double)(n - ts);
  double Wb = 0.5 * (GLs * F - gl * svv);   // W_b = (GL F - VV) / 2 + M' K_b
#pragma unroll
  for (int i = 0; i < M; ++i) Wb = fma(vM[i], vKb[i], Wb);
  const double Fb = -W * Wb * W;
  if (lane < M) {
    vMb[lane] = vKb[lane] * W;   // M_b = K_b W
    vFZ[lane] = Fb * z[lane];    // F_b Z
  }
  wave_lds_fence();
  if (lane < M) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < M; ++l) s = fma(vFZ[l], sP[l * M + lane], s);
    vFZP[lane] = s;              // F_b Z P
  }
  wave_lds_fence();
  for (int e = lane; e < MM; e += KF_WAVE) {
    const int i = e / M, j = e - i * M;
    out_Pb[b * (long)MM + e] = fma(vMb[i], z[j], z[i] * vFZ[j]);   // P_b = M_b Z + Z' F_b Z
    if (A.g_P0) A.g_P0[b * (long)MM + e] = 0.0;
    if (A.g_T) A.g_T[b * (long)MM + e] = fma(vSig[i], vK[j], sS[i * NC + j]);   // T_b = S1 + sigma K'
  }
  if (lane < M) {
    if (A.g_Z) {
      double zb = gl * W * sva[lane];       // Z_b = gl W Sva - tk' S1 ...
      double mp = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        zb = fma(-vTK[i], sS[i * NC + lane], zb);
        mp = fma(vMb[i], sP[i * M + lane], mp);
      }
      A.g_Z[b * (long)M + lane] = zb + 2.0 * vFZP[lane] + mp;   // ... + M_b' P + 2 F_b Z P
    }
    if (A.g_c) A.g_c[b * (long)M + lane] = sS[lane * NC + M + 1];
    if (A.g_a0) A.g_a0[b * (long)M + lane] = 0.0;
    out_ab[b * (long)M + lane] = carry[lane < M ? lane : 0];
  }
  if (lane == 0) {
    if (A.g_H) A.g_H[b] = Fb;
    if (A.g_d) {
      double s = -gl * W * sv;
#pragma unroll
      for (int i = 0; i < M; ++i) s = fma(vTK[i], sS[i * NC + M + 1], s);
      A.g_d[b] = dsign * s;
    }
  }
  if (A.g_R)
    for (int e = lane; e < M * r_; e += KF_WAVE) A.g_R[b * (long)M * r_ + e] = 0.0;
  if (A.g_Q)
    for (int e = lane; e < r_ * r_; e += KF_WAVE) A.g_Q[b * (long)r_ * r_ + e] = 0.0;
}

}  // namespace

bool kf_backward_steady_scan_ok(int m, int p) {
  const char* e = kf_env_switch("KF_VG_STEADY_SCAN");
  return p == 1 && m >= 1 && m <= 10 && !(e && e[0] == '0');
}

bool kf_backward_steady_ok(int m, int p, int r) { return m >= 1 && m <= SA_M && p >= 1 && p <= SA_P && r >= 1; }

// doubles of workspace: lam_ts [B][m] | P_b [B][m][m] | P [B][m][m]  (+ the int32 ts array, B entries, kept by the caller)
size_t kf_backward_steady_workspace_bytes(long B, int m) {
  return sizeof(double) * (size_t)B * (size_t)(m + 2 * m * m) + 4 * (size_t)B + 256;
}

int kf_launch_backward_steady(kf_handle* h, const KfBwdArgs& a, const int32_t* ts, double* out_ab, double* out_Pb, double* out_P,
                              hipStream_t stream, const double* P_conv) {
  if (!kf_backward_steady_ok(a.m, a.p, a.r) || (a.flags & KF_FLAG_TV_ANY) || a.g_ll_obs) {  // (one cotangent per draw: g_logp)
    KF_SET_ERR(h, "backward(steady): needs m <= %d, p <= %d, static matrices (got m=%d p=%d)", SA_M, SA_P, a.m, a.p);
    return KF_ERR_UNSUPPORTED;
  }
  // p == 1, m <= 10 with the converged covariance handed over (the p == 1 block kernel's value-and-gradient): the time-parallel form
  {
    if ((P_conv || a.pred_P) && kf_backward_steady_scan_ok(a.m, a.p)) {
#define KF_SS_LAUNCH(M_) \
  case M_: hipLaunchKernelGGL((kf_bwd_steady_scan_kernel<M_>), dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, ts, out_ab, out_Pb, out_P, P_conv); break;
      switch (a.m) {
        KF_SS_LAUNCH(1) KF_SS_LAUNCH(2) KF_SS_LAUNCH(3) KF_SS_LAUNCH(4) KF_SS_LAUNCH(5) KF_SS_LAUNCH(6) KF_SS_LAUNCH(7) KF_SS_LAUNCH(8)
        KF_SS_LAUNCH(9) KF_SS_LAUNCH(10)
      }
#undef KF_SS_LAUNCH
      KF_HIP_CHECK(h, hipGetLastError());
      return KF_OK;
    }
    if (a.fwd_logp) {
      KF_SET_ERR(h, "backward(steady): the forward hand-over needs the time-parallel kernel (m=%d p=%d)", a.m, a.p);
      return KF_ERR_UNSUPPORTED;
    }
  }
#define KF_SA_LAUNCH(PS_, MS_) \
  hipLaunchKernelGGL((kf_bwd_steady_kernel<PS_, MS_>), dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, ts, out_ab, out_Pb, out_P, P_conv)
  if (a.p == 1 && a.m <= 12) KF_SA_LAUNCH(1, 12);
  else if (a.p == 1) KF_SA_LAUNCH(1, 16);
  else if (a.m <= 12) KF_SA_LAUNCH(8, 12);
  else KF_SA_LAUNCH(8, 16);
#undef KF_SA_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
