// kf_forward_lane.hip -- one LANE per draw, everything in registers (compile-time M <= 4, p = 1).
//
// For tiny state dimension (BASELINE config C2: ARMA(1,1), m = 2; C1: m = 1) the row-owner
// kernels are bound by their per-step LDS exchanges (~2000 cycles of latency per time step
// whatever m is).  With m <= 4 a whole draw fits one lane's registers (P, T, RQR': 3 m^2
// doubles), so a time step is a short straight-line block of FMAs with NO exchange and NO sync:
// 64 draws per wavefront, step latency = its own issue time.
//
// Same contract and arithmetic as the other forward kernels (reference
// pymc_statespace/filters/kalman_filter.py:126-252, 255-351, 444-505): expanded Joseph form,
// P_hat = (S + S')/2 computed pairwise.  Tapes: each lane writes its draw's rows directly
// (m <= 4: 8..128 contiguous bytes per draw-step; consecutive time steps are adjacent in the
// draw-major tapes, so plain write-back stores let L2 assemble full lines).
#include "kf_internal.h"
#include "kf_tile.h"

#ifndef KF_LANE_M
#error "compile with -DKF_LANE_M=<k_states>"
#endif

namespace {

template <int M>
__global__ __launch_bounds__(KF_WAVE) void kf_fwd_lane_kernel(const KfFwdArgs A) {
  constexpr int MM = M * M;
  long b = (long)blockIdx.x * KF_WAVE + threadIdx.x;
  const bool live = b < A.B;
  if (!live) b = A.B - 1;
  const int n = A.T, r = A.r;

  double T[M][M], P[M][M], RQR[M][M], Z[M], a[M], c[M];
  {
    const double* gT = A.Tm + b * MM;
    const double* gP = A.P0 + b * MM;
    const double* gR = A.R + b * (long)M * r;
    const double* gQ = A.Q + b * (long)r * r;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      Z[i] = A.Z[b * M + i];
      a[i] = A.a0[b * M + i];
      c[i] = A.c ? A.c[b * M + i] : 0.0;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        T[i][j] = gT[i * M + j];
        P[i][j] = (j >= i) ? gP[i * M + j] : gP[j * M + i];  // upper triangle
        double s = 0.0;
        for (int l = 0; l < r; ++l) {
          double rq = 0.0;
          for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
          s += rq * gR[j * r + l];
        }
        RQR[i][j] = s;
      }
    }
  }
  const double Hs = A.H[b];
  const double ds = A.d ? A.d[b] : 0.0;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);

  double* o_fa = (A.filt_a && live) ? A.filt_a + b * (long)n * M : nullptr;
  double* o_pa = (A.pred_a && live) ? A.pred_a + b * (long)(n + 1) * M : nullptr;
  double* o_fP = (A.filt_P && live) ? A.filt_P + b * (long)n * MM : nullptr;
  double* o_pP = (A.pred_P && live) ? A.pred_P + b * (long)(n + 1) * MM : nullptr;
  double* o_ll = (A.ll_obs && live) ? A.ll_obs + b * (long)n : nullptr;
  if (o_pa) {
#pragma unroll
    for (int i = 0; i < M; ++i) o_pa[i] = a[i];
    o_pa += M;
  }
  if (o_pP) {
    const double* gP = A.P0 + b * MM;
#pragma unroll
    for (int e = 0; e < MM; ++e) o_pP[e] = gP[e];
    o_pP += MM;
  }

  double logp = 0.0;
  int info = 0;
  double ynext = ybase[0];
  // without the per-step tape the log is taken once per 16 steps: mantissa product, exponent sum
  const bool want_ll = A.ll_obs != nullptr;  // wave-uniform
  double mant = 1.0, qsum = 0.0;
  int ex = 0, cnt = 0;
  for (int t = 0; t < n; ++t) {
    const double yv = ynext;
    if (t + 1 < n) ynext = ybase[t + 1];
    const bool obs = !(yv != yv);
    double Mv[M];
    double F = Hs, za = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < M; ++k) s = fma(P[i][k], Z[k], s);
      Mv[i] = s;
      F = fma(Z[i], s, F);
      za = fma(Z[i], a[i], za);
    }
    double v = (yv + dsign * ds) - za;
    bool degenerate = !obs;
    if (univariate) {
      degenerate = degenerate || (F == 0.0);
      if (degenerate) F += 1e-8;
    } else if (!obs) {
      F = 1.0;
    }
    if (!(F > 0.0) && info == 0 && !degenerate) info = ((t + 1) << 8) | 1;
    const double invF = degenerate ? 0.0 : fast_rcp(F);
    if (degenerate) v = 0.0;
    double ll = 0.0;
    if (want_ll) {
      ll = degenerate ? 0.0 : -0.5 * (KF_LOG_2PI + log(F) + v * v * invF);
      logp += ll;
    } else {
      if (!degenerate) {
        int e2;
        const double fm = frexp(F, &e2);
        mant *= (F > 0.0) ? fm : __builtin_nan("");  // not PD: NaN, as log(F) would give
        ex += e2;
        qsum = fma(v * v, invF, qsum);
        ++cnt;
      }
      if ((t & 15) == 15 || t == n - 1) {
        logp += -0.5 * ((double)cnt * KF_LOG_2PI + log(mant) + 0.69314718055994530942 * (double)ex + qsum);
        mant = 1.0;
        qsum = 0.0;
        ex = 0;
        cnt = 0;
      }
    }
    double K[M], af[M], Pf[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      K[i] = Mv[i] * invF;
      af[i] = fma(K[i], v, a[i]);
    }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = i; j < M; ++j) {
        double s;
        if (univariate) {
          s = P[i][j] - K[i] * K[j] * F;
        } else {
          s = P[i][j] - K[i] * Mv[j];
          s = fma(-Mv[i], K[j], s);
          s = fma(F * K[i], K[j], s);
        }
        Pf[i][j] = s;
        Pf[j][i] = s;
      }
    if (o_fa) {
#pragma unroll
      for (int i = 0; i < M; ++i) o_fa[i] = af[i];
      o_fa += M;
    }
    if (o_fP) {
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) o_fP[i * M + j] = Pf[i][j];
      o_fP += MM;
    }
    if (o_ll) o_ll[t] = ll;
    // predict
    double W[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = c[i];
#pragma unroll
      for (int k = 0; k < M; ++k) s = fma(T[i][k], af[k], s);
      a[i] = s;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        double w = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) w = fma(T[i][k], Pf[k][j], w);
        W[i][j] = w;
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = i; j < M; ++j) {
        double sij = RQR[i][j], sji = RQR[j][i];
#pragma unroll
        for (int k = 0; k < M; ++k) {
          sij = fma(W[i][k], T[j][k], sij);
          sji = fma(W[j][k], T[i][k], sji);
        }
        const double s = 0.5 * (sij + sji);
        P[i][j] = s;
        P[j][i] = s;
      }
    if (o_pa) {
#pragma unroll
      for (int i = 0; i < M; ++i) o_pa[i] = a[i];
      o_pa += M;
    }
    if (o_pP) {
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) o_pP[i * M + j] = P[i][j];
      o_pP += MM;
    }
  }
  if (live) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

#define KF_CAT2(a, b) a##b
#define KF_CAT(a, b) KF_CAT2(a, b)
int KF_CAT(kf_launch_forward_lane_m, KF_LANE_M)(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  constexpr int M = KF_LANE_M;
  long blocks = (a.B + KF_WAVE - 1) / KF_WAVE;
  hipLaunchKernelGGL(kf_fwd_lane_kernel<M>, dim3((unsigned)blocks), dim3(KF_WAVE), 0, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
