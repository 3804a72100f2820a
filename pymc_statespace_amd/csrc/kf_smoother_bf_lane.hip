// kf_smoother_bf_lane.hip -- smoothed moments from the FILTER's predicted tapes, one LANE per draw (p == 1, m <= 4: C1, C2).
//
// The lane-per-draw form of kf_smoother_bf.hip (same recursion, same entry point: kf_smoother_from_filter), the backward counterpart
// of kf_forward_lane.hip.  What it replaces: KalmanSmoother.build_graph (pymc_statespace/filters/kalman_smoother.py:56-104) when its
// tapes come from this library's filter on the same model -- the only way the reference calls it (core/statespace.py:176-197).
// BayesianARMA has k_posdef = 1 and no measurement error (models/SARIMAX.py:23-49): the prediction covariance T P_f T' + R Q R' has
// rank one at almost every step, the reference's pinv(P_hat) sits on its cut-off, and the RTS lane kernel (kf_smoother_lane.hip)
// runs a Jacobi eigen-decomposition per step -- 0.76 ms at C2 and smoothed covariances that agree with statsmodels only to 1e-2
// (tests/test_gpu_parity.py: arma11.npz).  The recursion of de Jong (1989) / Durbin & Koopman (2012, 4.4) needs no m x m inverse
// (p = 1: F is a scalar) and IS what statsmodels computes:
//     v = y - Z a_t - d     F = Z P_t Z' + H     K = P_t Z' / F     a_f = a_t + K v     P_f = P_t - K (Z P_t)
//     u = T' r_t            M = T' N_t T
//     a_s[t] = a_f + P_f u                      P_s[t] = P_f - P_f M P_f
//     r_{t-1} = Z' v / F + (I - K Z)' u         N_{t-1} = Z' Z / F + (I - K Z)' M (I - K Z)         (r_{n-1} = 0, N_{n-1} = 0)
// (the form of kf_smoother_bf.hip: P_t - P_t N_{t-1} P_t = P_f - P_f M P_f; a missing observation, or F <= 0: K = 0, 1 / F := 0.)
// Every draw costs the same -- no data-dependent branch -- and nothing but the two predicted tapes is read: 2 (m + m^2) doubles per
// draw and step in and out.
//
// Only (r, N) is a recursion.  The first line above depends on the TAPE alone and the third on (u, M) of its own step, so a block of
// S steps is done in three phases and only the middle one is sequential (first version of this file: everything in one loop, 1000+
// cycles per step -- v_rcp_f64 + two Newton steps, two LDS round trips and ~85 fp64 instructions on one dependency chain; C2 0.57 ms):
//   1. side by side, one (draw slot, step) pair per lane: v / F, 1 / F, T K, a_f, P_f  -> LDS (P_f, a_f over the tape rows they came from)
//   2. step after step, every lane of a slot the same arithmetic: u, M -> LDS; k = (T K)' r, N (T K), r_{t-1}, N_{t-1} -- the carried
//      chain is N -> N (T K) -> T' N (T K) -> N_{t-1}: 2 m + 3 dependent FMAs
//   3. side by side again: a_s, P_s from (a_f, P_f, u, M), written over a_f, P_f; the block leaves as whole lines
// DPW in {4, 16, 64} draws per wavefront from the batch, so that the launch covers the SIMDs: lane l works on draw slot l % DPW and,
// in phases 1 and 3, on the steps l / DPW, l / DPW + 64 / DPW, ... of the block.  The next block's tape rows are fetched into
// registers before phase 1, so that HBM latency runs beside the arithmetic (kf_forward_lane.hip's y_fetch / y_commit).
#include <type_traits>

#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int bfl_rec_doubles(int M) { return (2 * M * M + 3 * M + 2 + (M * (M + 1)) / 2) | 1; }  // odd: steps on different banks
constexpr int bfl_block_steps(int M, int DPW) {
  int s = 16;
  while (s > 1 && DPW * (bfl_rec_doubles(M) * s + 1) * 8 > 40 * 1024) s >>= 1;
  return s;
}

template <int M, int DPW>
struct BflLds {
  static constexpr int MM = M * M, NTRI = (M * (M + 1)) / 2;
  static constexpr int S = bfl_block_steps(M, DPW);
  // one record per (slot, step): P_t -> P_f -> P_s | a_t -> a_f -> a_s | T K | y -> v / F | 1 / F | u | M (upper triangle)
  static constexpr int O_P = 0, O_A = MM, O_TK = MM + M, O_VF = MM + 2 * M, O_RF = O_VF + 1, O_U = O_RF + 1, O_M = O_U + M;
  static constexpr int W = bfl_rec_doubles(M);
  static_assert(O_M + NTRI <= W, "record layout");
  static constexpr int SLOT = S * W + ((S * W) % 2 == 0 ? 1 : 0);  // odd slot stride
  static constexpr int TOTAL = DPW * SLOT;
};

template <int M, int DPW>
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_bf_lane_kernel(const KfSmoothBfArgs A) {
  using L = BflLds<M, DPW>;
  constexpr int MM = M * M, S = L::S, W = L::W, LPS = KF_WAVE / DPW;
  constexpr int NP = (DPW * S * MM + KF_WAVE - 1) / KF_WAVE, NA = (DPW * S * M + KF_WAVE - 1) / KF_WAVE,
                NY = (DPW * S + KF_WAVE - 1) / KF_WAVE;
  __shared__ double smem[L::TOTAL];
  const int lane = threadIdx.x;
  const int d = lane % DPW, jj = lane / DPW;
  const long b0 = (long)blockIdx.x * DPW;
  long b = b0 + d;
  if (b >= A.B) b = A.B - 1;
  const int n = A.T;

  double T[M][M], Z[M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    Z[i] = A.Z[b * M + i];
#pragma unroll
    for (int j = 0; j < M; ++j) T[i][j] = A.Tm[b * MM + i * M + j];
  }
  const double Hs = A.H[b];
  const double ds = A.d ? A.d[b] : 0.0;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // kalman_filter.py:335 quirk: y_hat = Z a - d
  const bool y_per_draw = (A.flags & KF_FLAG_Y_PER_DRAW) != 0;
  double* rec0 = smem + d * L::SLOT;

  // ---- tape rows [t0, t0 + cnt) of the wavefront's draws <-> the records: a slot's rows of a block are adjacent in the draw-major
  // tapes, so consecutive lanes move consecutive doubles (whole lines)
  double preP[NP], preA[NA], preY[NY];
  auto fetch = [&](int t0, int cnt) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / (S * MM), rem = e - dd * (S * MM), st = rem / MM;
      const bool ok = ((DPW * S * MM) % KF_WAVE == 0 || dd < DPW) && st < cnt && b0 + dd < A.B;
      preP[k] = ok ? A.pred_P[((b0 + dd) * (long)(n + 1) + t0) * MM + rem] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / (S * M), rem = e - dd * (S * M), st = rem / M;
      const bool ok = ((DPW * S * M) % KF_WAVE == 0 || dd < DPW) && st < cnt && b0 + dd < A.B;
      preA[k] = ok ? A.pred_a[((b0 + dd) * (long)(n + 1) + t0) * M + rem] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NY; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / S, st = e - dd * S;
      const bool ok = ((DPW * S) % KF_WAVE == 0 || dd < DPW) && st < cnt && b0 + dd < A.B;
      preY[k] = ok ? A.y[(y_per_draw ? (b0 + dd) * (long)n : 0L) + t0 + st] : 0.0;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / (S * MM), rem = e - dd * (S * MM), st = rem / MM, el = rem - st * MM;
      if ((DPW * S * MM) % KF_WAVE == 0 || dd < DPW) smem[dd * L::SLOT + st * W + L::O_P + el] = preP[k];
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / (S * M), rem = e - dd * (S * M), st = rem / M, el = rem - st * M;
      if ((DPW * S * M) % KF_WAVE == 0 || dd < DPW) smem[dd * L::SLOT + st * W + L::O_A + el] = preA[k];
    }
#pragma unroll
    for (int k = 0; k < NY; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / S, st = e - dd * S;
      if ((DPW * S) % KF_WAVE == 0 || dd < DPW) smem[dd * L::SLOT + st * W + L::O_VF] = preY[k];
    }
    wave_lds_fence();
  };
  auto copy_out = [&](int t0, int cnt) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / (S * MM), rem = e - dd * (S * MM), st = rem / MM, el = rem - st * MM;
      if (((DPW * S * MM) % KF_WAVE == 0 || dd < DPW) && st < cnt && b0 + dd < A.B)
        A.smooth_P[((b0 + dd) * (long)n + t0) * MM + rem] = smem[dd * L::SLOT + st * W + L::O_P + el];
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int e = lane + KF_WAVE * k, dd = e / (S * M), rem = e - dd * (S * M), st = rem / M, el = rem - st * M;
      if (((DPW * S * M) % KF_WAVE == 0 || dd < DPW) && st < cnt && b0 + dd < A.B)
        A.smooth_a[((b0 + dd) * (long)n + t0) * M + rem] = smem[dd * L::SLOT + st * W + L::O_A + el];
    }
    wave_lds_fence();
  };

  double r[M], N[M][M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    r[i] = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) N[i][j] = 0.0;
  }
  const int t_last = ((n - 1) / S) * S;
  fetch(t_last, n - t_last);
  for (int t0 = t_last; t0 >= 0; t0 -= S) {
    const int cnt = (n - t0) < S ? (n - t0) : S;
    commit();
    if (t0 > 0) fetch(t0 - S, S);
    // ---- phase 1: the filter's quantities of every step of the block (kalman_filter.py:255-284 at p = 1), side by side --------------
    for (int ss = jj; ss < cnt; ss += LPS) {
      double* rc = rec0 + ss * W;
      double a[M], P[M][M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        a[i] = rc[L::O_A + i];
#pragma unroll
        for (int j = 0; j < M; ++j) P[i][j] = rc[L::O_P + i * M + j];
      }
      const double yv = rc[L::O_VF];
      double Mv[M], F = Hs, za = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double sm = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) sm = fma(P[i][k], Z[k], sm);
        Mv[i] = sm;
        F = fma(Z[i], sm, F);
        za = fma(Z[i], a[i], za);
      }
      const bool upd = !(yv != yv) && F > 0.0;  // (F <= 0: the filter flagged the draw; UnivariateFilter skips F == 0, :468-471)
      const double rF = upd ? fast_rcp(F) : 0.0;
      const double v = upd ? (yv + dsign * ds) - za : 0.0;
      double K[M];
#pragma unroll
      for (int i = 0; i < M; ++i) K[i] = Mv[i] * rF;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double tk = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) tk = fma(T[i][k], K[k], tk);
        rc[L::O_TK + i] = tk;
        rc[L::O_A + i] = fma(K[i], v, a[i]);
      }
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
          const double sm = fma(-K[i], Mv[j], P[i][j]);
          rc[L::O_P + i * M + j] = sm;
          if (j != i) rc[L::O_P + j * M + i] = sm;
        }
      rc[L::O_VF] = v * rF;
      rc[L::O_RF] = rF;
    }
    wave_lds_fence();
    // ---- phase 2: the recursion, step after step (every lane of a slot: same values to the same addresses) ---------------------------
    for (int s = cnt - 1; s >= 0; --s) {
      double* rc = rec0 + s * W;
      double TK[M];
#pragma unroll
      for (int i = 0; i < M; ++i) TK[i] = rc[L::O_TK + i];
      const double vF = rc[L::O_VF], rF = rc[L::O_RF];
      // u = T' r_t,  Mm = T' N_t T -> the record (phase 3);  N (T K), k = (T K)' r,  (T K)' N (T K)
      double u[M], NT[M][M], NTK[M], ku = 0.0, kmk = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double sm = 0.0, w = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
          sm = fma(T[k][i], r[k], sm);
          w = fma(N[i][k], TK[k], w);
        }
        u[i] = sm;
        NTK[i] = w;
        ku = fma(TK[i], r[i], ku);
#pragma unroll
        for (int j = 0; j < M; ++j) {
          double x = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) x = fma(N[i][k], T[k][j], x);
          NT[i][j] = x;
        }
      }
      double Mm[M][M], MK[M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        kmk = fma(TK[i], NTK[i], kmk);
        double w = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) w = fma(T[k][i], NTK[k], w);
        MK[i] = w;
        rc[L::O_U + i] = u[i];
      }
      {
        int tri = 0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i; j < M; ++j) {
            double sm = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(T[k][i], NT[k][j], sm);
            Mm[i][j] = sm;
            rc[L::O_M + tri] = sm;
            ++tri;
          }
      }
      // r_{t-1} = Z' v / F + (I - K Z)' u,  N_{t-1} = Z' Z / F + (I - K Z)' Mm (I - K Z)
      const double zz = rF + kmk;  // coefficient of Z' Z:  1 / F + K' Mm K
#pragma unroll
      for (int i = 0; i < M; ++i) r[i] = fma(Z[i], vF - ku, u[i]);
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
          double sm = Mm[i][j];
          sm = fma(-MK[i], Z[j], sm);
          sm = fma(-Z[i], MK[j], sm);
          sm = fma(Z[i] * zz, Z[j], sm);
          N[i][j] = sm;
          N[j][i] = sm;
        }
    }
    wave_lds_fence();
    // ---- phase 3: a_s = a_f + P_f u,  P_s = P_f - P_f Mm P_f, side by side -----------------------------------------------------------
    for (int ss = jj; ss < cnt; ss += LPS) {
      double* rc = rec0 + ss * W;
      double af[M], Pf[M][M], u[M], Mm[M][M];
      {
        int tri = 0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          af[i] = rc[L::O_A + i];
          u[i] = rc[L::O_U + i];
#pragma unroll
          for (int j = 0; j < M; ++j) Pf[i][j] = rc[L::O_P + i * M + j];
#pragma unroll
          for (int j = i; j < M; ++j) {
            const double x = rc[L::O_M + tri];
            Mm[i][j] = x;
            Mm[j][i] = x;
            ++tri;
          }
        }
      }
      double Y[M][M];
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          double sm = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) sm = fma(Mm[i][k], Pf[k][j], sm);
          Y[i][j] = sm;
        }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double sm = af[i];
#pragma unroll
        for (int k = 0; k < M; ++k) sm = fma(Pf[i][k], u[k], sm);
        rc[L::O_A + i] = sm;
      }
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
          double sm = Pf[i][j];
#pragma unroll
          for (int k = 0; k < M; ++k) sm = fma(-Pf[i][k], Y[k][j], sm);
          rc[L::O_P + i * M + j] = sm;
          if (j != i) rc[L::O_P + j * M + i] = sm;
        }
    }
    wave_lds_fence();
    copy_out(t0, cnt);
  }
}

}  // namespace

bool kf_smoother_bf_lane_ok(int m, int p, unsigned flags) { return p == 1 && m >= 1 && m <= 4 && !(flags & KF_FLAG_TV_ANY); }

int kf_launch_smoother_bf_lane(kf_handle* h, const KfSmoothBfArgs& a, hipStream_t stream) {
  if (!kf_smoother_bf_lane_ok(a.m, 1, a.flags)) {
    KF_SET_ERR(h, "smoother(from filter, lane): needs p == 1, m <= 4 and static matrices");
    return KF_ERR_UNSUPPORTED;
  }
  const long simds = (h && h->n_cu > 0 ? h->n_cu : 256) * 4L;
  const int dpw = a.B <= 4 * simds ? 4 : (a.B <= 16 * simds ? 16 : 64);
#define KF_BFL2(M_, D_) \
  hipLaunchKernelGGL((kf_smooth_bf_lane_kernel<M_, D_>), dim3((unsigned)((a.B + D_ - 1) / D_)), dim3(KF_WAVE), 0, stream, a)
#define KF_BFL(M_)                       \
  do {                                   \
    if (dpw == 4) KF_BFL2(M_, 4);        \
    else if (dpw == 16) KF_BFL2(M_, 16); \
    else KF_BFL2(M_, 64);                \
  } while (0)
  switch (a.m) {
    case 1: KF_BFL(1); break;
    case 2: KF_BFL(2); break;
    case 3: KF_BFL(3); break;
    default: KF_BFL(4); break;
  }
#undef KF_BFL
#undef KF_BFL2
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
