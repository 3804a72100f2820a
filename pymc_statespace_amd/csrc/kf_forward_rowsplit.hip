// kf_forward_rowsplit.hip -- row-owner forward Kalman filter with the two m x m x m products
// split over H lanes per row (compile-time even M = k_states <= 10, p = 1).
//
// Why: at the north-star shape (4096 draws x T=1000 x m=10) the batch is too small to fill the
// chip with one-wave-per-SIMD row-owner waves (6 draws per wave -> 683 waves for 1024 SIMDs),
// and a lone wave exposes every LDS round trip.  Splitting each row's W/S columns over H lanes
// shortens the per-step instruction stream of a wave by ~H for the O(m^3) part and raises the
// wave count by H, so two waves share a SIMD and hide each other's latency.
//
// Mapping: M*H lanes per draw, floor(64/(M*H)) draws per wavefront, one wavefront per
// workgroup.  Lane (i,h): i = row, h = column-pair slice.  Every lane of row i keeps the FULL
// row i of P (cheap O(m) work is done redundantly: M_i, F, v, K_i, a_f) but computes
// P_f, W = T P_f and S = W T' + RQR' only for its own column pairs c = h, h+H, h+2H, ...
// Per step, wave-local LDS exchanges:
//   (1) all-gather of M = P Z' and of the partial innovations
//   (2) all-gather of P_f (tile X; also the staging tile of the filtered-covariance tape)
//   (3) H > 1 only: the row's W entries are reassembled from its H lanes (tile Y)
//   (4) transpose of S (tile X) for the symmetrisation P_hat = (S + S')/2 (:221); the new rows
//       are written to tile Y, the staging tile of the predicted-covariance tape
// T is streamed from LDS (static).  Tape stores: 16 bytes per lane, consecutive lanes on
// consecutive addresses (see kf_forward_row.hip).  Arithmetic = kf_fwd_lds_kernel's.
#include "kf_internal.h"

#if !defined(KF_ROW_M) || !defined(KF_ROW_H)
#error "compile with -DKF_ROW_M=<even k_states> -DKF_ROW_H=<lanes per row>"
#endif

namespace {

template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  __builtin_nontemporal_store(v, p);
}

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int M, int H>
struct SplitCfg {
  static_assert(M % 2 == 0, "row-split kernel needs an even state dimension");
  static constexpr int MM = M * M;
  static constexpr int NPAIR = M / 2;                           // column pairs per row
  static constexpr int NP = (NPAIR + H - 1) / H;                // pairs per lane (last may be a clone)
  static constexpr int LPD = M * H;                             // lanes per draw
  static_assert(LPD <= KF_WAVE, "M*H must fit a wavefront");
  static constexpr int DPW = KF_WAVE / LPD;                     // draws per wavefront
  static constexpr int RING = 64;                               // time steps per ll batch
  static constexpr int RAW = 3 * MM + 2 * M + 2 * RING + 2;     // TLh, X, Y, V, ring(F, q)
  static constexpr int STRIDE = ((RAW + 29) / 32) * 32 + 2;     // == 2 (mod 32)
  static constexpr int LDS_DOUBLES = DPW * STRIDE;
  static constexpr int TILE = DPW * MM;
  static constexpr int NST = (TILE + 2 * KF_WAVE - 1) / (2 * KF_WAVE);
};

template <int M, int H>
__global__ __launch_bounds__(KF_WAVE) void kf_fwd_rowsplit_kernel(const KfFwdArgs A) {
  using C = SplitCfg<M, H>;
  constexpr int MM = C::MM, NP = C::NP;
  __shared__ __attribute__((aligned(16))) double smem[C::LDS_DOUBLES];
  const int lane = threadIdx.x;
  int grp = lane / C::LPD;
  const int lid = lane - grp * C::LPD;
  const int i = lid / H;       // row
  const int h = lid - i * H;   // column-pair slice
  const bool lane_ok = grp < C::DPW;
  if (!lane_ok) grp = C::DPW - 1;  // spare lanes shadow rows of the last draw, store nothing
  const long b0 = (long)blockIdx.x * C::DPW;
  long b = b0 + grp;
  const bool live = lane_ok && b < A.B;
  if (b >= A.B) b = A.B - 1;
  const int n = A.T, r = A.r;

  double* TLh = smem + grp * C::STRIDE;  // static T/2 (exact scaling; folds the 1/2 of (S+S')/2)
  double* X = TLh + MM;                 // P_f rows, then S rows
  double* Y = X + MM;                   // W rows, then predicted P rows
  double* V = Y + MM;                   // 2M exchange vector
  double* RG = V + 2 * M;               // ring: (F_t, v_t^2/F_t) of the last <= 64 steps

  // own column pairs: first column j0 = 2*(h + pp*H), clamped (surplus slots clone the last pair)
  int j0[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    int c = h + pp * H;
    if (c > C::NPAIR - 1) c = C::NPAIR - 1;
    j0[pp] = 2 * c;
  }

  // ---- per-draw constants ------------------------------------------------------------------------
  double Trow[M], Zr[M], Prow[M];
  double RQRo[NP][2], Po[NP][2];
  {
    const double* gT = A.Tm + b * MM;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      Trow[k] = gT[i * M + k];
      TLh[i * M + k] = 0.5 * Trow[k];
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
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = j0[pp] + e;
        double s = 0.0;
        for (int l = 0; l < r; ++l) {
          double rq = 0.0;
          for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
          s += rq * gR[j * r + l];
        }
        RQRo[pp][e] = 0.5 * s;
        Po[pp][e] = (j >= i) ? gP[i * M + j] : gP[j * M + i];
      }
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

  const bool own = live && h == 0;  // one lane per row writes the vector tapes
  double* o_fa = (A.filt_a && own) ? A.filt_a + b * (long)n * M + i : nullptr;
  double* o_pa = (A.pred_a && own) ? A.pred_a + b * (long)(n + 1) * M + i : nullptr;

  // ---- coalesced tape staging slots (see kf_forward_row.hip) ----------------------------------------
  const bool stage_fP = A.filt_P != nullptr, stage_pP = A.pred_P != nullptr;
  int st_lds[C::NST];
  bool st_on[C::NST];
  double* gfP[C::NST];
  double* gpP[C::NST];
#pragma unroll
  for (int q = 0; q < C::NST; ++q) {
    int e = 2 * lane + 2 * KF_WAVE * q;
    st_on[q] = e < C::TILE;              // surplus slots of the last round store nothing
    if (e >= C::TILE) e -= C::TILE;
    int dr = e / MM;
    const int off = e - dr * MM;
    st_lds[q] = dr * C::STRIDE + MM + off;  // tile X of draw dr; tile Y = + MM
    long bd = b0 + dr;
    if (bd >= A.B) bd = A.B - 1;         // tail draws are clones of the last draw: identical bytes
    gfP[q] = stage_fP ? A.filt_P + (bd * (long)n) * MM + off : nullptr;
    gpP[q] = stage_pP ? A.pred_P + (bd * (long)(n + 1)) * MM + off : nullptr;
  }

  // predicted tape slot 0 = (a0, P0), staged through tile Y
  if (o_pa) { *o_pa = ai; o_pa += M; }
  if (stage_pP) {
    const double* gP = A.P0 + b * MM + i * M;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      Y[i * M + j0[pp]] = gP[j0[pp]];
      Y[i * M + j0[pp] + 1] = gP[j0[pp] + 1];
    }
  }
  bool pending_pP = stage_pP;  // predicted rows sit in tile Y, stored after the next sync

  double logp = 0.0;   // lane `lane` accumulates the ll of the ring slots it evaluates
  int info = 0;
  double ynext = ybase[0];

  // evaluate the ll of the steps [tb, tb+cnt) held in the ring: one (draw, step) per lane
  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < C::DPW; ++dd) {
      const double* rg = smem + dd * C::STRIDE + 3 * MM + 2 * M;
      const double2_t fq = *reinterpret_cast<const double2_t*>(rg + 2 * lane);
      const bool act = lane < cnt;
      const double Fv = act ? fq.x : -1.0;
      // -1 marks a step without likelihood contribution; F <= 0 otherwise (not PD) gives NaN
      const double llv = (Fv == -1.0) ? 0.0 : -0.5 * (KF_LOG_2PI + log(Fv) + fq.y);
      const long bd = b0 + dd;
      if (act && bd < A.B && A.ll_obs) st_stream(A.ll_obs + bd * (long)n + tb + lane, llv);
      // wave reduction -> every lane holds the sum; lane dd*LPD keeps draw dd's total
      double ssum = llv;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
      if (lane == dd * C::LPD) logp += ssum;
    }
  };

  for (int t = 0; t < n; ++t) {
    const double yv = ynext;
    if (t + 1 < n) ynext = ybase[t + 1];
    const bool obs = !(yv != yv);
    // ---- (1) M_i = P[i,:] Z' ; partial innovation ------------------------------------------------
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      m0 = fma(Prow[k], Zr[k], m0);
      m1 = fma(Prow[k + 1], Zr[k + 1], m1);
    }
    const double Mi = m0 + m1;
    V[2 * i] = Mi;
    V[2 * i + 1] = Zi * ai;
    __syncthreads();  // also publishes the predicted rows written to tile Y at the end of step t-1
    if (pending_pP) {
#pragma unroll
      for (int q = 0; q < C::NST; ++q) {
        if (st_on[q])
          st_stream(reinterpret_cast<double2_t*>(gpP[q]),
                    *reinterpret_cast<const double2_t*>(smem + st_lds[q] + MM));
        gpP[q] += MM;
      }
    }
    double F0 = Hs, F1 = 0.0, z0 = 0.0, z1 = 0.0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      F0 = fma(Zr[k], V[2 * k], F0);
      z0 += V[2 * k + 1];
      F1 = fma(Zr[k + 1], V[2 * k + 2], F1);
      z1 += V[2 * k + 3];
    }
    double Mo[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      Mo[pp][0] = V[2 * j0[pp]];
      Mo[pp][1] = V[2 * j0[pp] + 2];
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
    // ---- P_f, own columns: expanded Joseph form (Univariate: P - K K' F, :476) ----------------------------
    const double FKi = F * Ki;
    double Pf[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double Mj = Mo[pp][e];
        const double Kj = Mj * invF;
        double s;
        if (univariate) {
          s = Po[pp][e] - Ki * Kj * F;
        } else {
          s = Po[pp][e] - Ki * Mj;
          s = fma(-Mi, Kj, s);
          s = fma(FKi, Kj, s);
        }
        Pf[pp][e] = s;
      }
    }
    // ---- (2) all-gather P_f and a_f ------------------------------------------------------------------------------
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      double2_t t2 = {Pf[pp][0], Pf[pp][1]};
      *reinterpret_cast<double2_t*>(X + i * M + j0[pp]) = t2;
    }
    V[i] = afi;
    __syncthreads();
    if (lid == 0) {  // F <= 0 marks "no likelihood contribution" (missing / degenerate / not PD)
      double2_t fq = {degenerate ? -1.0 : F, v * v * invF};
      *reinterpret_cast<double2_t*>(RG + 2 * (t & (C::RING - 1))) = fq;
    }
    if (o_fa) { *o_fa = afi; o_fa += M; }
    if (stage_fP) {
#pragma unroll
      for (int q = 0; q < C::NST; ++q) {
        if (st_on[q])
          st_stream(reinterpret_cast<double2_t*>(gfP[q]),
                    *reinterpret_cast<const double2_t*>(smem + st_lds[q]));
        gfP[q] += MM;
      }
    }
    // W[i, own] = T[i,:] P_f[:, own] ;  a_hat_i = T[i,:] a_f + c_i
    double Wo[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) Wo[pp][0] = Wo[pp][1] = 0.0;
    double an = ci;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double tk = Trow[k];
      an = fma(tk, V[k], an);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const double2_t x2 = *reinterpret_cast<const double2_t*>(X + k * M + j0[pp]);
        Wo[pp][0] = fma(tk, x2.x, Wo[pp][0]);
        Wo[pp][1] = fma(tk, x2.y, Wo[pp][1]);
      }
    }
    ai = an;
    // ---- (3) reassemble the row of W from its H lanes ----------------------------------------------------------------
    double Wr[M];
    if (H > 1) {
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        double2_t t2 = {Wo[pp][0], Wo[pp][1]};
        *reinterpret_cast<double2_t*>(Y + i * M + j0[pp]) = t2;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < M; ++k) Wr[k] = Y[i * M + k];
    } else {
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        Wr[2 * pp] = Wo[pp][0];      // H == 1: j0[pp] == 2*pp, the lane owns the whole row
        Wr[2 * pp + 1] = Wo[pp][1];
      }
    }
    // S[i, own] = W[i,:] T[own,:]' + RQR'[i, own]   (T streamed from LDS: static)
    double So[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double* Tj = TLh + (j0[pp] + e) * M;
        double s0 = RQRo[pp][e], s1 = 0.0;
#pragma unroll
        for (int k = 0; k < M; k += 2) {
          s0 = fma(Wr[k], Tj[k], s0);
          s1 = fma(Wr[k + 1], Tj[k + 1], s1);
        }
        So[pp][e] = s0 + s1;
      }
    }
    // ---- (4) transpose S through tile X: P_hat = (S + S') / 2 -------------------------------------------------------------
    // (every lane finished reading P_f from X before sync (3); with H == 1 the S rows go to
    //  tile Y instead, which nobody reads between the top-of-step stores and here)
    double* Ttile = (H > 1) ? X : Y;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      double2_t t2 = {So[pp][0], So[pp][1]};
      *reinterpret_cast<double2_t*>(Ttile + i * M + j0[pp]) = t2;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < M; ++k) Prow[k] = Ttile[i * M + k] + Ttile[k * M + i];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      Po[pp][0] = So[pp][0] + Ttile[j0[pp] * M + i];
      Po[pp][1] = So[pp][1] + Ttile[(j0[pp] + 1) * M + i];
    }
    if (o_pa) { *o_pa = ai; o_pa += M; }
    if ((t & (C::RING - 1)) == C::RING - 1) flush_ring(t - (C::RING - 1), C::RING);
    if (stage_pP) {
      // predicted rows -> tile Y (H > 1: W rows in Y were consumed before sync (4);
      // H == 1: the S rows in Y were consumed just above by this wave in program order --
      // other lanes read S[k][i] from Y, so a sync is needed before overwriting)
      if (H == 1) __syncthreads();
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        double2_t t2 = {Po[pp][0], Po[pp][1]};
        *reinterpret_cast<double2_t*>(Y + i * M + j0[pp]) = t2;
      }
    }
  }
  if (pending_pP) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::NST; ++q)
      if (st_on[q])
        st_stream(reinterpret_cast<double2_t*>(gpP[q]),
                  *reinterpret_cast<const double2_t*>(smem + st_lds[q] + MM));
  }
  if (n & (C::RING - 1)) flush_ring(n & ~(C::RING - 1), n & (C::RING - 1));
  if (live && lid == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

#define KF_CAT3_(a, b, c, d) a##b##c##d
#define KF_CAT3(a, b, c, d) KF_CAT3_(a, b, c, d)
int KF_CAT3(kf_launch_forward_rowsplit_m, KF_ROW_M, _h, KF_ROW_H)(kf_handle* h, const KfFwdArgs& a,
                                                                 hipStream_t stream) {
  constexpr int M = KF_ROW_M, HH = KF_ROW_H;
  using C = SplitCfg<M, HH>;
  long blocks = (a.B + C::DPW - 1) / C::DPW;
  hipLaunchKernelGGL((kf_fwd_rowsplit_kernel<M, HH>), dim3((unsigned)blocks), dim3(KF_WAVE), 0, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
