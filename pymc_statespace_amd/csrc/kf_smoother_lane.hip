// kf_smoother_lane.hip -- RTS smoother, one LANE per draw, everything in registers (m <= 4: C1, C2).
//
// The backward counterpart of kf_forward_lane.hip (reference kalman_smoother.py:56-104, same order of operations as
// kf_smoother_lds.hip: no intercept, no symmetrisation of P_hat):
//    a_hat = T a_f ;  P_hat = T P_f T' + R Q R' ;  X = pinv(P_hat) (T P_f)
//    a_s   = a_f + X' (a_s' - a_hat) ;  P_s = P_f + X' (P_s' - P_hat) X
// pinv (kalman_smoother.py:92, numpy.linalg.pinv): while the Cholesky pivots of P_hat stay above 1e-12 x its largest diagonal
// entry the pseudo-inverse IS the inverse and X comes from two triangular solves in registers; otherwise -- config C2, an
// ARMA(1,1) without measurement noise: P_f -> 0 and P_hat -> R Q R' has rank one at almost every step -- the lane runs a
// Jacobi eigen-decomposition of P_hat in its own registers (m = 2: one rotation, exact; m = 3, 4: cyclic sweeps) and builds the
// Moore-Penrose inverse with numpy's cut-off (eigenvalues <= 1e-15 max |eigenvalue| dropped), the semantics of
// kf_smoother_lds.hip's fallback -- which every draw of C2 took at every step (3.4 ms; round 2 parked this kernel for that reason).
// DPW in {4, 16, 64} draws per wavefront picked from the batch so that the launch covers the chip's SIMDs (lane l works on draw slot
// l % DPW; replicas write the same values to the same LDS address).  The tapes move through LDS blocks of S steps: the filtered
// rows of a block are copied in and the smoothed rows copied out by the whole wavefront, 8 contiguous bytes per lane (a draw's S
// rows are adjacent in the draw-major tapes: whole lines).
#include <type_traits>

#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int sl_block_steps(int M, int DPW) {
  int s = 16;
  while (s > 2 && DPW * (2 * (M + M * M) * s + 4) * 8 > 48 * 1024) s >>= 1;
  return s;
}

template <int M, int DPW>
struct SlLds {
  static constexpr int MM = M * M;
  static constexpr int S = sl_block_steps(M, DPW);
  static constexpr int W_A = S * M + 1;  // odd chunk strides: the DPW slots hit different banks
  static constexpr int W_P = S * MM + 1;
  static constexpr int OFF_FA = 0;
  static constexpr int OFF_FP = OFF_FA + DPW * W_A;
  static constexpr int OFF_SA = OFF_FP + DPW * W_P;
  static constexpr int OFF_SP = OFF_SA + DPW * W_A;
  static constexpr int TOTAL = OFF_SP + DPW * W_P;
};

// Moore-Penrose inverse of the symmetric matrix D (its upper triangle; destroyed) in one lane's registers
template <int M>
__device__ __forceinline__ void pinv_sym_lane(double (&D)[M][M], double (&Pi)[M][M]) {
  double V[M][M];
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      V[i][j] = i == j ? 1.0 : 0.0;
      if (j < i) D[i][j] = D[j][i];
    }
  for (int sweep = 0; sweep < (M <= 2 ? 1 : 40); ++sweep) {
    if (M > 2) {
      double off = 0.0, dg = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          if (i == j) dg = fma(D[i][j], D[i][j], dg);
          else off = fma(D[i][j], D[i][j], off);
        }
      if (!__any(off > 1e-30 * dg && off > 0.0)) break;  // (the lanes in this branch only; converged ones do identity rotations)
    }
#pragma unroll
    for (int p_ = 0; p_ < M - 1; ++p_)
#pragma unroll
      for (int q_ = p_ + 1; q_ < M; ++q_) {
        const double apq = D[p_][q_];
        double c = 1.0, s = 0.0;
        if (apq != 0.0) {
          const double tau = (D[q_][q_] - D[p_][p_]) / (2.0 * apq);
          const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(fma(tau, tau, 1.0)));
          c = 1.0 / sqrt(fma(t, t, 1.0));
          s = t * c;
        }
#pragma unroll
        for (int k = 0; k < M; ++k) {
          const double dkp = D[k][p_], dkq = D[k][q_];
          D[k][p_] = c * dkp - s * dkq;
          D[k][q_] = s * dkp + c * dkq;
          const double vkp = V[k][p_], vkq = V[k][q_];
          V[k][p_] = c * vkp - s * vkq;
          V[k][q_] = s * vkp + c * vkq;
        }
#pragma unroll
        for (int k = 0; k < M; ++k) {
          const double dpk = D[p_][k], dqk = D[q_][k];
          D[p_][k] = c * dpk - s * dqk;
          D[q_][k] = s * dpk + c * dqk;
        }
      }
  }
  double smax = 0.0, ev[M];
#pragma unroll
  for (int k = 0; k < M; ++k) smax = fmax(smax, fabs(D[k][k]));
#pragma unroll
  for (int k = 0; k < M; ++k) ev[k] = (fabs(D[k][k]) > 1e-15 * smax) ? 1.0 / D[k][k] : 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      double sm = 0.0;
#pragma unroll
      for (int k = 0; k < M; ++k) sm = fma(V[i][k] * V[j][k], ev[k], sm);
      Pi[i][j] = sm;
    }
}

template <int M, int DPW>
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_lane_kernel(const KfSmoothArgs A) {
  using L = SlLds<M, DPW>;
  constexpr int MM = M * M, S = L::S;
  __shared__ double smem[L::TOTAL];
  const int lane = threadIdx.x;
  const int d = lane % DPW;
  const long b0 = (long)blockIdx.x * DPW;
  long b = b0 + d;
  const bool live = b < A.B;
  if (!live) b = A.B - 1;
  const int n = A.T, r = A.r;

  double T[M][M], RQR[M][M];
  {
    const double* gT = A.Tm + b * MM;
    const double* gR = A.R + b * (long)M * r;
    const double* gQ = A.Q + b * (long)r * r;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        T[i][j] = gT[i * M + j];
        double s = 0.0;
        for (int l = 0; l < r; ++l) {
          double rq = 0.0;
          for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
          s += rq * gR[j * r + l];
        }
        RQR[i][j] = s;
      }
  }
  double* c_fa = smem + L::OFF_FA + d * L::W_A;
  double* c_fP = smem + L::OFF_FP + d * L::W_P;
  double* c_sa = smem + L::OFF_SA + d * L::W_A;
  double* c_sP = smem + L::OFF_SP + d * L::W_P;

  // rows [row0, row0 + cnt) of a draw-major tape <-> the per-slot LDS chunks, 8 contiguous bytes per lane
  auto copy = [&](auto in_tag, const double* tape_in, double* tape_out, double* lds, int stride, auto wtag, int row0, int cnt) {
    constexpr bool IN = decltype(in_tag)::value;
    constexpr int W = decltype(wtag)::value;
    const int cw = cnt * W;
    for (int e = lane; e < DPW * cw; e += KF_WAVE) {
      const int dd = e / cw, off = e - dd * cw;
      if (b0 + dd < A.B) {
        const long g_ = ((b0 + dd) * (long)n + row0) * W + off;
        if (IN) lds[dd * stride + off] = tape_in[g_];
        else tape_out[g_] = lds[dd * stride + off];
      }
    }
  };

  double as[M], Ps[M][M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    as[i] = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) Ps[i][j] = 0.0;
  }
  // m <= 2: where the filtered covariance of EVERY draw of the wavefront is bit for bit the previous step's (the constant tapes a
  // steady-state forward writes; an every-step tape at its floating-point fixpoint) the last X and P_hat are reused -- no Cholesky, no
  // Jacobi rotation; the results are those of the full step (kf_smoother_b4.hip does the same per draw).  KF_FLAG_EXACT_RECURSION: never.
  constexpr bool REUSE = M <= 2;
  double Xk[REUSE ? M : 1][REUSE ? M : 1], Phk[REUSE ? M : 1][REUSE ? M : 1], Pfk[REUSE ? M : 1][REUSE ? M : 1];
  bool have_k = false;
  for (int t0 = ((n - 1) / S) * S; t0 >= 0; t0 -= S) {
    const int cnt = (n - t0) < S ? (n - t0) : S;
    copy(std::true_type{}, A.filt_a, nullptr, smem + L::OFF_FA, L::W_A, std::integral_constant<int, M>{}, t0, cnt);
    copy(std::true_type{}, A.filt_P, nullptr, smem + L::OFF_FP, L::W_P, std::integral_constant<int, MM>{}, t0, cnt);
    __syncthreads();
    for (int s = cnt - 1; s >= 0; --s) {
      double af[M], Pf[M][M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        af[i] = c_fa[s * M + i];
#pragma unroll
        for (int j = 0; j < M; ++j) Pf[i][j] = c_fP[s * MM + i * M + j];
      }
      if (t0 + s == n - 1) {  // the last step: smoothed = filtered
#pragma unroll
        for (int i = 0; i < M; ++i) {
          as[i] = af[i];
#pragma unroll
          for (int j = 0; j < M; ++j) Ps[i][j] = Pf[i][j];
        }
      } else {
        double W[M][M], Ph[M][M], ah[M], X[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
          double sm = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) sm = fma(T[i][k], af[k], sm);
          ah[i] = sm;
        }
        bool reuse = false;
        if constexpr (REUSE) {
          if (have_k && !A.exact) {
            bool same = true;
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
              for (int j = 0; j < M; ++j) same = same && (Pf[i][j] == Pfk[i][j]);
            reuse = __all(same || !live);
          }
#pragma unroll
          for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) Pfk[i][j] = Pf[i][j];
        }
        if (reuse) {
          if constexpr (REUSE) {
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
              for (int j = 0; j < M; ++j) {
                X[i][j] = Xk[i][j];
                Ph[i][j] = Phk[i][j];
              }
          }
        } else {
#pragma unroll
        for (int i = 0; i < M; ++i) {
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double w = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) w = fma(T[i][k], Pf[k][j], w);
            W[i][j] = w;
          }
        }
        double dmax = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double sm = RQR[i][j];
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(W[i][k], T[j][k], sm);
            Ph[i][j] = sm;
            if (i == j) dmax = fmax(dmax, sm);
          }
        // Cholesky P_hat = L L' (lower triangle of P_hat), 1 / L_ii kept
        double Lc[M][M], il[M];
        const double piv_min = 1e-12 * dmax;
        bool bad = !(dmax > 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) {
          double dj = Ph[j][j];
#pragma unroll
          for (int k = 0; k < j; ++k) dj = fma(-Lc[j][k], Lc[j][k], dj);
          if (!(dj > piv_min)) bad = true;
          const double ok = dj > 0.0 ? dj : 1.0;
          const double rl = 1.0 / sqrt(ok);
          il[j] = rl;
          Lc[j][j] = ok * rl;
#pragma unroll
          for (int i = j + 1; i < M; ++i) {
            double v = Ph[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = fma(-Lc[i][k], Lc[j][k], v);
            Lc[i][j] = v * rl;
          }
        }
        // X = P_hat^-1 W: L Y = W, L' X = Y (column by column of W)
        if (bad) {  // singular (or indefinite by rounding) P_hat: X = pinv(P_hat) W
          double Dm[M][M], Pi[M][M];
#pragma unroll
          for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) Dm[i][j] = Ph[i][j];
          pinv_sym_lane<M>(Dm, Pi);
#pragma unroll
          for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) {
              double sm = 0.0;
#pragma unroll
              for (int k = 0; k < M; ++k) sm = fma(Pi[i][k], W[k][j], sm);
              X[i][j] = sm;
            }
        } else {
#pragma unroll
        for (int cidx = 0; cidx < M; ++cidx) {
          double yv[M];
#pragma unroll
          for (int i = 0; i < M; ++i) {
            double v = W[i][cidx];
#pragma unroll
            for (int k = 0; k < i; ++k) v = fma(-Lc[i][k], yv[k], v);
            yv[i] = v * il[i];
          }
#pragma unroll
          for (int i = M - 1; i >= 0; --i) {
            double v = yv[i];
#pragma unroll
            for (int k = i + 1; k < M; ++k) v = fma(-Lc[k][i], X[k][cidx], v);
            X[i][cidx] = v * il[i];
          }
        }
        }
        if constexpr (REUSE) {
          have_k = true;
#pragma unroll
          for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) {
              Xk[i][j] = X[i][j];
              Phk[i][j] = Ph[i][j];
            }
        }
        }
        // a_s = a_f + X' (a_s' - a_hat);  P_s = P_f + X' (P_s' - P_hat) X
        double da[M], E[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i) da[i] = as[i] - ah[i];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double sm = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(Ps[i][k] - Ph[i][k], X[k][j], sm);
            E[i][j] = sm;
          }
#pragma unroll
        for (int i = 0; i < M; ++i) {
          double sm = af[i];
#pragma unroll
          for (int k = 0; k < M; ++k) sm = fma(X[k][i], da[k], sm);
          as[i] = sm;
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double sm = Pf[i][j];
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(X[k][i], E[k][j], sm);
            Ps[i][j] = sm;
          }
      }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        c_sa[s * M + i] = as[i];
#pragma unroll
        for (int j = 0; j < M; ++j) c_sP[s * MM + i * M + j] = Ps[i][j];
      }
    }
    __syncthreads();
    copy(std::false_type{}, nullptr, A.smooth_a, smem + L::OFF_SA, L::W_A, std::integral_constant<int, M>{}, t0, cnt);
    copy(std::false_type{}, nullptr, A.smooth_P, smem + L::OFF_SP, L::W_P, std::integral_constant<int, MM>{}, t0, cnt);
    __syncthreads();
  }
}

}  // namespace

bool kf_smoother_lane_ok(int m) { return m >= 1 && m <= 4; }

int kf_launch_smoother_lane(kf_handle* h, const KfSmoothArgs& a, hipStream_t stream) {
  if (!kf_smoother_lane_ok(a.m) || a.flags) {
    KF_SET_ERR(h, "smoother(lane): needs m <= 4 and static matrices");
    return KF_ERR_UNSUPPORTED;
  }
  const long simds = (h && h->n_cu > 0 ? h->n_cu : 256) * 4L;
  const int dpw = a.B <= 4 * simds ? 4 : (a.B <= 16 * simds ? 16 : 64);
#define KF_SL2(M_, D_) \
  hipLaunchKernelGGL((kf_smooth_lane_kernel<M_, D_>), dim3((unsigned)((a.B + D_ - 1) / D_)), dim3(KF_WAVE), 0, stream, a)
#define KF_SL(M_)                 \
  do {                            \
    if (dpw == 4) KF_SL2(M_, 4);  \
    else if (dpw == 16) KF_SL2(M_, 16); \
    else KF_SL2(M_, 64);          \
  } while (0)
  switch (a.m) {
    case 1: KF_SL(1); break;
    case 2: KF_SL(2); break;
    case 3: KF_SL(3); break;
    default: KF_SL(4); break;
  }
#undef KF_SL
#undef KF_SL2
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
