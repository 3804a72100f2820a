// kf_forward_row.hip -- row-owner forward Kalman filter for small state dimension
// (compile-time M = k_states <= 10, p = k_endog = 1): the speed path for BASELINE configs
// C1-C3 and the north-star headline (4096 draws x T=1000 x m=10).
//
// Same contract as kf_fwd_lds_kernel (reference pymc_statespace/filters/kalman_filter.py:
// 126-252, 255-351, 444-505); different mapping:
//   * M lanes per draw, floor(64/M) draws per wavefront, one wavefront per workgroup.
//     Lane i of a draw OWNS row i of the covariance P, of T and of R Q R' (VGPRs), a_i, and
//     a copy of Z.  The draw's whole T also sits in LDS (static for all T steps).
//   * Three wave-local LDS exchanges per time step:
//       (1) all-gather of M = P Z' and of the partial innovations  -> F, v, K in every lane
//       (2) all-gather of the P_f rows (and a_f)                   -> W[i,:] = T[i,:] P_f
//           then S[i,:] = W[i,:] T' + RQR'[i,:] streams T from LDS (static: no exchange)
//       (3) transpose of S for the symmetrisation P_hat = (S + S')/2 (:221)
//   * Tape stores are staged through the same LDS tiles: a lane never stores its own row
//     (80-byte pieces 80 bytes apart would shred every 128-byte line into partial writes);
//     instead the wave streams the draws' contiguous m*m blocks with 16 bytes per lane,
//     consecutive lanes on consecutive addresses (non-temporal, write-once data).
//   * All loops are fully unrolled (compile-time M): LDS offsets are immediates, reads are
//     broadcast within a draw and hit distinct 16-byte bank slots across the draws of a wave
//     (region stride = 2 mod 32 doubles).
// Arithmetic is that of the LDS kernel (expanded Joseph form, symmetrised P_hat).
#include "kf_internal.h"

#ifndef KF_ROW_M
#error "compile with -DKF_ROW_M=<k_states>"
#endif

namespace {

template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int M>
struct RowCfg {
  static constexpr int MM = M * M;
  static constexpr int DPW = KF_WAVE / M;                       // draws per wavefront
  static constexpr int RAW = 3 * MM + 2 * M + 2;                // TL, X, Y, V (+pad)
  static constexpr int STRIDE = ((RAW + 29) / 32) * 32 + 2;     // == 2 (mod 32), >= RAW
  static constexpr int LDS_DOUBLES = DPW * STRIDE;
  static constexpr int TILE = DPW * MM;                         // doubles of one tape step, whole wave
  static constexpr int NST = (TILE + 2 * KF_WAVE - 1) / (2 * KF_WAVE);  // 16-byte stores per lane
  static constexpr bool PAIRS = (MM % 2) == 0;                  // 16-byte staging needs even m*m
};

template <int M>
__global__ __launch_bounds__(KF_WAVE) void kf_fwd_row_kernel(const KfFwdArgs A) {
  using C = RowCfg<M>;
  constexpr int MM = C::MM;
  __shared__ __attribute__((aligned(16))) double smem[C::LDS_DOUBLES];
  const int lane = threadIdx.x;
  int grp = lane / M;
  const int i = lane - grp * M;  // row owned by this lane
  const bool lane_ok = grp < C::DPW;
  if (!lane_ok) grp = C::DPW - 1;  // spare lanes shadow the last draw of the wave, store nothing
  const long b0 = (long)blockIdx.x * C::DPW;
  long b = b0 + grp;
  const bool live = lane_ok && b < A.B;
  if (b >= A.B) b = A.B - 1;
  const int n = A.T, r = A.r;

  double* TL = smem + grp * C::STRIDE;  // static T (row-major)
  double* X = TL + MM;                  // exchange tile 1 (P_f rows / predicted P rows)
  double* Y = X + MM;                   // exchange tile 2 (S rows)
  double* V = Y + MM;                   // 2M exchange vector

  // ---- per-draw constants ------------------------------------------------------------------------
  double Trow[M], RQRrow[M], Zr[M], Prow[M];
  {
    const double* gT = A.Tm + b * MM;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      Trow[k] = gT[i * M + k];
      TL[i * M + k] = Trow[k];
    }
    const double* gZ = A.Z + b * M;
#pragma unroll
    for (int k = 0; k < M; ++k) Zr[k] = gZ[k];
    const double* gP = A.P0 + b * MM;
#pragma unroll
    for (int k = 0; k < M; ++k) Prow[k] = (k >= i) ? gP[i * M + k] : gP[k * M + i];  // upper triangle
    const double* gR = A.R + b * (long)M * r;
    const double* gQ = A.Q + b * (long)r * r;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      double s = 0.0;
      for (int l = 0; l < r; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
        s += rq * gR[j * r + l];
      }
      RQRrow[j] = s;
    }
  }
  const double Zi = A.Z[b * M + i];
  const double Hs = A.H[b];
  const double ds = A.d ? A.d[b] : 0.0;
  const double ci = A.c ? A.c[b * M + i] : 0.0;
  double ai = A.a0[b * M + i];

  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);

  double* o_fa = (A.filt_a && live) ? A.filt_a + b * (long)n * M + i : nullptr;
  double* o_pa = (A.pred_a && live) ? A.pred_a + b * (long)(n + 1) * M + i : nullptr;
  double* o_ll = (A.ll_obs && live && i == 0) ? A.ll_obs + b * (long)n : nullptr;

  // ---- coalesced tape staging: slot q of this lane = doubles (e, e+1) of the wave's tile -----------
  // e = 2*lane + 128*q ; draw = e / MM ; off = e % MM   (compile-time M: cheap, done once)
  const bool stage_fP = A.filt_P != nullptr, stage_pP = A.pred_P != nullptr;
  int st_lds[C::NST];   // LDS offset (doubles) of the slot inside tile X of its draw
  double* gfP[C::NST];  // global address of the slot in the filtered / predicted covariance tape
  double* gpP[C::NST];  // (advance by MM per time step); nullptr = dead slot
#pragma unroll
  for (int q = 0; q < C::NST; ++q) {
    const int e = 2 * lane + 2 * KF_WAVE * q;
    const int dr = e / MM, off = e - dr * MM;
    const bool okslot = C::PAIRS && e < C::TILE && (b0 + dr) < A.B;
    st_lds[q] = (dr < C::DPW ? dr : 0) * C::STRIDE + MM + off;
    gfP[q] = (okslot && stage_fP) ? A.filt_P + ((b0 + dr) * (long)n) * MM + off : nullptr;
    gpP[q] = (okslot && stage_pP) ? A.pred_P + ((b0 + dr) * (long)(n + 1)) * MM + off : nullptr;
  }

  // predicted tape slot 0 = (a0, P0)
  if (o_pa) { *o_pa = ai; o_pa += M; }
  if (stage_pP) {
    const double* gP = A.P0 + b * MM + i * M;
#pragma unroll
    for (int k = 0; k < M; ++k) X[i * M + k] = gP[k];
  }
  __syncthreads();
  if (stage_pP) {
    if (C::PAIRS) {
#pragma unroll
      for (int q = 0; q < C::NST; ++q)
        if (gpP[q]) {
          st_stream(reinterpret_cast<double2_t*>(gpP[q]),
                    *reinterpret_cast<const double2_t*>(smem + st_lds[q]));
          gpP[q] += MM;
        }
    } else if (live) {
      double* o = A.pred_P + b * (long)(n + 1) * MM + i * M;
#pragma unroll
      for (int k = 0; k < M; ++k) o[k] = X[i * M + k];
    }
  }
  double* o_fP_row = (!C::PAIRS && stage_fP && live) ? A.filt_P + b * (long)n * MM + i * M : nullptr;
  double* o_pP_row = (!C::PAIRS && stage_pP && live) ? A.pred_P + (b * (long)(n + 1) + 1) * MM + i * M : nullptr;

  double logp = 0.0;
  int info = 0;
  bool pending_pP = false;  // predicted rows of the previous step sit in X, not yet stored

  for (int t = 0; t < n; ++t) {
    const double yv = ybase[t];
    const bool obs = !(yv != yv);
    // ---- (1) M_i = P[i,:] Z' ; partial innovation ------------------------------------------------
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      m0 = fma(Prow[k], Zr[k], m0);
      if (k + 1 < M) m1 = fma(Prow[k + 1], Zr[k + 1], m1);
    }
    const double Mi = m0 + m1;
    V[2 * i] = Mi;
    V[2 * i + 1] = Zi * ai;
    __syncthreads();  // also publishes the predicted rows written to X at the end of step t-1
    if (pending_pP) {
      if (C::PAIRS) {
#pragma unroll
        for (int q = 0; q < C::NST; ++q)
          if (gpP[q]) {
            st_stream(reinterpret_cast<double2_t*>(gpP[q]),
                      *reinterpret_cast<const double2_t*>(smem + st_lds[q]));
            gpP[q] += MM;
          }
      }
    }
    double Mall[M];
    double F0 = Hs, F1 = 0.0, z0 = 0.0, z1 = 0.0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      Mall[k] = V[2 * k];
      z0 += V[2 * k + 1];
      F0 = fma(Zr[k], Mall[k], F0);
      if (k + 1 < M) {
        Mall[k + 1] = V[2 * k + 2];
        z1 += V[2 * k + 3];
        F1 = fma(Zr[k + 1], Mall[k + 1], F1);
      }
    }
    double F = F0 + F1;
    double v = (yv + dsign * ds) - (z0 + z1);
    bool degenerate = !obs;
    if (univariate) {  // :468-471
      degenerate = degenerate || (F == 0.0);
      if (degenerate) F += 1e-8;
    } else if (!obs) {
      F = 1.0;  // all-missing step: F + I*flag with Z masked to 0 (:268)
    }
    if (!(F > 0.0) && info == 0 && !degenerate) info = ((t + 1) << 8) | 1;
    const double invF = degenerate ? 0.0 : 1.0 / F;
    if (degenerate) v = 0.0;
    const double Ki = Mi * invF;
    const double afi = fma(Ki, v, ai);
    // ---- P_f row: expanded Joseph form (Univariate: P - K K' F, :476) ----------------------------------
    double Pf[M];
    const double FKi = F * Ki;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double Kk = Mall[k] * invF;
      double s;
      if (univariate) {
        s = Prow[k] - Ki * Kk * F;
      } else {
        s = Prow[k] - Ki * Mall[k];
        s = fma(-Mi, Kk, s);
        s = fma(FKi, Kk, s);
      }
      Pf[k] = s;
    }
    // ---- (2) all-gather P_f rows and a_f ------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < M; ++k) X[i * M + k] = Pf[k];
    V[i] = afi;
    __syncthreads();
    const double ll = degenerate ? 0.0 : -0.5 * (KF_LOG_2PI + log(F) + v * v * invF);
    logp += ll;
    if (o_fa) { *o_fa = afi; o_fa += M; }
    if (o_ll) o_ll[t] = ll;
    if (stage_fP) {
      if (C::PAIRS) {
#pragma unroll
        for (int q = 0; q < C::NST; ++q)
          if (gfP[q]) {
            st_stream(reinterpret_cast<double2_t*>(gfP[q]),
                      *reinterpret_cast<const double2_t*>(smem + st_lds[q]));
            gfP[q] += MM;
          }
      } else if (o_fP_row) {
#pragma unroll
        for (int k = 0; k < M; ++k) o_fP_row[k] = Pf[k];
        o_fP_row += MM;
      }
    }
    // W[i,:] = T[i,:] P_f ;  a_hat_i = T[i,:] a_f + c_i
    double Wr[M];
#pragma unroll
    for (int j = 0; j < M; ++j) Wr[j] = 0.0;
    double an = ci;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double tk = Trow[k];
      an = fma(tk, V[k], an);
#pragma unroll
      for (int j = 0; j < M; ++j) Wr[j] = fma(tk, X[k * M + j], Wr[j]);
    }
    ai = an;
    // S[i,:] = W[i,:] T' + RQR'[i,:]   (T streamed from LDS: static, no exchange)
    double Sr[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
      double s0 = RQRrow[j], s1 = 0.0;
#pragma unroll
      for (int k = 0; k < M; k += 2) {
        s0 = fma(Wr[k], TL[j * M + k], s0);
        if (k + 1 < M) s1 = fma(Wr[k + 1], TL[j * M + k + 1], s1);
      }
      Sr[j] = s0 + s1;
    }
    // ---- (3) transpose S: P_hat[i,:] = (S[i,:] + S[:,i]) / 2 ---------------------------------------------------------
#pragma unroll
    for (int j = 0; j < M; ++j) Y[i * M + j] = Sr[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M; ++j) Prow[j] = 0.5 * (Sr[j] + Y[j * M + i]);
    if (o_pa) { *o_pa = ai; o_pa += M; }
    if (stage_pP) {
      if (C::PAIRS) {
        // rows go to X (all lanes finished reading P_f from X before the sync above);
        // they are stored after the next sync (top of the next step / after the loop)
#pragma unroll
        for (int k = 0; k < M; ++k) X[i * M + k] = Prow[k];
        pending_pP = true;
      } else if (o_pP_row) {
#pragma unroll
        for (int k = 0; k < M; ++k) o_pP_row[k] = Prow[k];
        o_pP_row += MM;
      }
    }
  }
  if (pending_pP && C::PAIRS) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::NST; ++q)
      if (gpP[q])
        st_stream(reinterpret_cast<double2_t*>(gpP[q]),
                  *reinterpret_cast<const double2_t*>(smem + st_lds[q]));
  }
  if (live && i == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

#define KF_CAT2(a, b) a##b
#define KF_CAT(a, b) KF_CAT2(a, b)
int KF_CAT(kf_launch_forward_row_m, KF_ROW_M)(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  constexpr int M = KF_ROW_M;
  using C = RowCfg<M>;
  long blocks = (a.B + C::DPW - 1) / C::DPW;
  hipLaunchKernelGGL(kf_fwd_row_kernel<M>, dim3((unsigned)blocks), dim3(KF_WAVE), 0, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
