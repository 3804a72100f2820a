// kf_forward_lane.hip -- one LANE per draw, everything in registers (compile-time M <= 4, p = 1).
//
// For tiny state dimension (BASELINE config C2: ARMA(1,1), m = 2; C1: m = 1) the row-owner
// kernels are bound by their per-step LDS exchanges (~2000 cycles of latency per time step
// whatever m is).  With m <= 4 a whole draw fits one lane's registers (P, T, RQR': 3 m^2
// doubles), so a time step is a short straight-line block of FMAs with NO exchange and NO sync.
//
// Round 2 (C2 ran on 64 wavefronts, each lane storing 13 scattered doubles per step):
//   * DPW draws per wavefront, DPW in {4, 16, 64} chosen from the batch so that the launch covers the
//     chip's 1024 SIMDs (4096 draws: 1024 wavefronts of 4 draws).  All 64 lanes run the recursion --
//     lane l that of draw slot l % DPW, replicas are free -- and lanes < DPW own the results.
//   * Tapes leave through LDS: the owners append each step's rows to per-draw chunks, and every S steps
//     the whole wavefront copies the chunks out, 8 contiguous bytes per lane: a draw's S rows are adjacent
//     in the draw-major tapes, so the stores are whole 128-byte lines instead of 64 scattered 8-byte pieces.
//   * The logarithm is off the recursion: (F, v^2/F, observed) go to the LDS block and the S x DPW
//     log-likelihood terms are evaluated side by side at the flush, one per lane; ll_obs is stored
//     coalesced from there and logp is the sum of the lanes' partial sums (a different summation order
//     than the sequential kernels: differences at rounding level, tests: 1e-8).
//
// Same contract and arithmetic as the other forward kernels (reference
// pymc_statespace/filters/kalman_filter.py:126-252, 255-351, 444-505): expanded Joseph form,
// P_hat = (S + S')/2 computed pairwise.
#include <stdlib.h>

#include <type_traits>

#include "kf_internal.h"
#include "kf_tile.h"

#ifndef KF_LANE_M
#error "compile with -DKF_LANE_M=<k_states>"
#endif

namespace {

// steps per LDS block: the largest power of two <= 16 that keeps a wavefront's block under 40 KiB
constexpr int lane_block_steps(int M, int DPW) {
  int s = 16;
  while (s > 2 && DPW * ((2 * M + 2 * M * M + 4) * s + 8) * 8 > 40 * 1024) s >>= 1;
  return s;
}

template <int M, int DPW>
struct LaneLds {
  static constexpr int MM = M * M;
  static constexpr int S = lane_block_steps(M, DPW);
  static constexpr int W_A = S * M + 1;   // chunk strides are odd: the DPW owners hit different banks
  static constexpr int W_P = S * MM + 1;
  static constexpr int W_L = S + 1;
  static constexpr int OFF_FA = 0;
  static constexpr int OFF_PA = OFF_FA + DPW * W_A;
  static constexpr int OFF_FP = OFF_PA + DPW * W_A;
  static constexpr int OFF_PP = OFF_FP + DPW * W_P;
  static constexpr int OFF_LF = OFF_PP + DPW * W_P;
  static constexpr int OFF_LQ = OFF_LF + DPW * W_L;
  static constexpr int OFF_LO = OFF_LQ + DPW * W_L;
  static constexpr int OFF_YB = OFF_LO + DPW * W_L;   // y of the block, per draw slot (shared y: replicated)
  static constexpr int OFF_ACC = OFF_YB + DPW * W_L;
  static constexpr int TOTAL = OFF_ACC + DPW;
};

// TM: which tapes are written, known at compile time (bit 0 filtered states, 1 predicted states, 2 filtered
// covariances, 3 predicted covariances) -- a taken scalar branch costs about as much as four dependent FMAs and the
// step had a dozen of them; TM < 0: decided at run time (the uncommon combinations)
// SS: with the steady-state phase (KF_FLAG_EXACT_RECURSION runs the instantiation without it: the round-2 kernel as it was)
template <int M, int DPW, int TM, bool SS>
__global__ __launch_bounds__(KF_WAVE) void kf_fwd_lane_kernel(const KfFwdArgs A) {
  using L = LaneLds<M, DPW>;
  constexpr int MM = M * M, S = L::S;
  constexpr int NK = (DPW * S + KF_WAVE - 1) / KF_WAVE;  // log-likelihood terms per lane and block
  __shared__ double smem[L::TOTAL];
  // (round 6) stage of the chunked first 256 steps: per lane four steps of the five outputs, odd strides
  constexpr bool PRE_OK = SS && M <= 2 && DPW == 4 && S == 16;
  __shared__ double pstage[PRE_OK ? KF_WAVE * (2 * (4 * M + 1) + 2 * (4 * MM + 1) + 5) : 1];
  const int lane = threadIdx.x;
  // Round 6, DPW == 4 ("ROWMAP"): a draw slot is one DPP ROW of sixteen lanes (slot = lane / 16, replica / chunk / step index = lane % 16),
  // so that what the sixteen lanes of a slot exchange -- the prefix scan of the frozen phase, the chained means of the chunked first
  // 256 steps -- are row_shr / row_newbcast moves instead of ds_bpermute round trips (the frozen phase's block of sixteen steps was six
  // dependent LDS-crossbar stages).  The other widths keep slot = lane % DPW.
  constexpr bool ROWMAP = DPW == 4;
  const int d = ROWMAP ? lane / 16 : lane % DPW;
  const int jrep = ROWMAP ? lane % 16 : lane / DPW;   // which of the slot's lanes this is
  const bool owner = jrep == 0;
  // the value of the slot's lane K places down / one place up / of its last lane (ROWMAP: lanes without a source keep their own value)
  auto slot_up = [&](double x, auto kc) -> double {
    constexpr int K = decltype(kc)::value;
    if constexpr (ROWMAP) return dpp_mov<0x110 + K>(x);   // row_shr:K
    else return __shfl_up(x, K * DPW, KF_WAVE);
  };
  auto slot_next = [&](double x) -> double {
    if constexpr (ROWMAP) return dpp_mov<0x101>(x);       // row_shl:1
    else return __shfl_down(x, DPW, KF_WAVE);
  };
  auto slot_last = [&](double x) -> double {
    if constexpr (ROWMAP) return dpp_mov<0x15F>(x);       // row_newbcast:15
    else return __shfl(x, d + DPW * (KF_WAVE / DPW - 1), KF_WAVE);
  };
  const long b0 = (long)blockIdx.x * DPW;
  long b = b0 + d;
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
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const bool w_fa = TM < 0 ? A.filt_a != nullptr : (TM & 1) != 0, w_pa = TM < 0 ? A.pred_a != nullptr : (TM & 2) != 0,
             w_fP = TM < 0 ? A.filt_P != nullptr : (TM & 4) != 0, w_pP = TM < 0 ? A.pred_P != nullptr : (TM & 8) != 0;

  // predicted slot 0 = (a0, P0 as given)
  if (owner && live) {
    if (w_pa) {
#pragma unroll
      for (int i = 0; i < M; ++i) A.pred_a[b * (long)(n + 1) * M + i] = a[i];
    }
    if (w_pP) {
      const double* gP = A.P0 + b * MM;
#pragma unroll
      for (int e = 0; e < MM; ++e) A.pred_P[b * (long)(n + 1) * MM + e] = gP[e];
    }
  }
  if (owner) smem[L::OFF_ACC + d] = 0.0;

  double* s_fa = smem + L::OFF_FA + d * L::W_A;
  double* s_pa = smem + L::OFF_PA + d * L::W_A;
  double* s_fP = smem + L::OFF_FP + d * L::W_P;
  double* s_pP = smem + L::OFF_PP + d * L::W_P;
  double* s_lf = smem + L::OFF_LF + d * L::W_L;
  double* s_lq = smem + L::OFF_LQ + d * L::W_L;
  double* s_lo = smem + L::OFF_LO + d * L::W_L;

  // chunk [draw slot][cnt rows of W doubles] in LDS -> rows [row0, row0 + cnt) of a draw-major tape with `rows`
  // rows per draw: 8 contiguous bytes per lane
  auto copy_out = [&](double* tape, const double* lds, int stride, auto wtag, int rows, int row0, int cnt) {
    constexpr int W = decltype(wtag)::value;
    if (cnt == S) {
      // (round 6: every LDS read of the block first, then the stores -- with the bounds test inside, each of the up to twelve copies
      // of a flush was a basic block of its own: read, wait for it, store; one wavefront per SIMD has nothing else to run meanwhile)
      // ... and sixteen bytes per lane: a slot's S rows are S * W consecutive doubles on both sides (S * W is even)
      constexpr int CW = S * W;
      constexpr int NP = DPW * CW / 2;                          // pairs of doubles
      constexpr int NI = (NP + KF_WAVE - 1) / KF_WAVE;
      static_assert(CW % 2 == 0, "pairs never straddle two slots");
      double2_t vals[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        int pi = k * KF_WAVE + lane;
        if (NP % KF_WAVE != 0 && pi >= NP) pi = NP - 1;
        const int e = 2 * pi;
        const int dd = e / CW, off = e - dd * CW;
        vals[k] = double2_t{lds[dd * stride + off], lds[dd * stride + off + 1]};
      }
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        asm volatile("" : "+v"(vals[k]));   // (pinned here: the compiler sinks each load into its store's branch)
      }
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int pi = k * KF_WAVE + lane;
        const int e = 2 * pi;
        const int dd = e / CW, off = e - dd * CW;
        if ((NP % KF_WAVE == 0 || pi < NP) && b0 + dd < A.B) {
          double* dst = tape + ((b0 + dd) * (long)rows + row0) * W + off;
          if ((W % 2 == 0) || ((reinterpret_cast<unsigned long long>(dst) & 15ull) == 0ull)) {
            *reinterpret_cast<double2_t*>(dst) = vals[k];
          } else {   // (W odd: an odd row start is 8-byte aligned only)
            dst[0] = vals[k].x;
            dst[1] = vals[k].y;
          }
        }
      }
    } else {
      const int cw = cnt * W;
      for (int e = lane; e < DPW * cw; e += KF_WAVE) {
        const int dd = e / cw, off = e - dd * cw;
        if (b0 + dd < A.B) tape[((b0 + dd) * (long)rows + row0) * W + off] = lds[dd * stride + off];
      }
    }
  };

  // y reaches the recursion through LDS, a block of S steps at a time: the loads of block k + 1 are issued when block k
  // is flushed and land in LDS after its copy-out (a per-step global load would put L2 latency on every step)
  const bool y_per_draw = (A.flags & KF_FLAG_Y_PER_DRAW) != 0;
  // (round 6: TWO blocks ahead -- the loads of block k + 2 are issued at the flush of block k and committed at the flush of block
  // k + 1: with the covariance frozen a block of sixteen steps is ~1.4 us, less than a load from HBM under load takes)
  double ynx[NK], yn2[NK];
  auto y_fetch = [&](int t0) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int e = lane + KF_WAVE * k;
      const int dd = e / S, ss = e - dd * S;
      long bb = b0 + dd;
      if (bb >= A.B) bb = A.B - 1;
      const int tt = (t0 + ss < n) ? t0 + ss : n - 1;
      yn2[k] = ((DPW * S) % KF_WAVE == 0 || e < DPW * S) ? A.y[(y_per_draw ? bb * (long)n : 0L) + tt] : 0.0;
    }
  };
  auto y_commit = [&]() {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int e = lane + KF_WAVE * k;
      const int dd = e / S, ss = e - dd * S;
      if ((DPW * S) % KF_WAVE == 0 || e < DPW * S) smem[L::OFF_YB + dd * L::W_L + ss] = ynx[k];
    }
    wave_lds_fence();
  };
  auto y_shift = [&]() {
#pragma unroll
    for (int k = 0; k < NK; ++k) ynx[k] = yn2[k];
  };
  const double* s_yb = smem + L::OFF_YB + d * L::W_L;
  double lsum[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) lsum[k] = 0.0;
  auto flush = [&](int t0, int cnt) {
    wave_lds_fence();
    if (t0 + 2 * cnt < n) y_fetch(t0 + 2 * cnt);   // (a partial block is the last one: cnt == S whenever another block follows)
    if (w_fa) copy_out(A.filt_a, smem + L::OFF_FA, L::W_A, std::integral_constant<int, M>{}, n, t0, cnt);
    if (w_pa) copy_out(A.pred_a, smem + L::OFF_PA, L::W_A, std::integral_constant<int, M>{}, n + 1, t0 + 1, cnt);
    if (w_fP) copy_out(A.filt_P, smem + L::OFF_FP, L::W_P, std::integral_constant<int, MM>{}, n, t0, cnt);
    if (w_pP) copy_out(A.pred_P, smem + L::OFF_PP, L::W_P, std::integral_constant<int, MM>{}, n + 1, t0 + 1, cnt);
    // the block's log-likelihood terms, one (draw slot, step) pair per lane and round
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int e = lane + KF_WAVE * k;
      const int dd = e / S, ss = e - dd * S;
      if (((DPW * S) % KF_WAVE == 0 || e < DPW * S) && ss < cnt) {
        const double Fv = smem[L::OFF_LF + dd * L::W_L + ss];
        const double quad = smem[L::OFF_LQ + dd * L::W_L + ss];
        const bool observed = smem[L::OFF_LO + dd * L::W_L + ss] != 0.0;
        const bool skipped = observed && univariate && Fv == 0.0;  // :468-471
        double llv;
        if (!observed) llv = 0.0;
        else if (skipped) llv = exact_const ? -0.5 * KF_LOG_2PI : 0.0;
        else if (univariate && !exact_const && Fv == 1.0 && quad == 0.0) llv = 0.0;  // :497-500
        else llv = -0.5 * (KF_LOG_2PI + log(Fv) + quad);
        if (A.ll_obs && b0 + dd < A.B) A.ll_obs[(b0 + dd) * (long)n + t0 + ss] = llv;
        lsum[k] += llv;
      }
    }
    wave_lds_fence();
    if (t0 + cnt < n) {
      y_commit();    // block t0 + cnt, fetched one flush ago
      y_shift();
    }
  };

  int info = 0;
  int t = 0;
  constexpr bool SCAN = SS && M <= 2 && DPW == 4 && S == 16;
  // ---- round 6: the first 256 steps as SIXTEEN CHUNKS of 16 steps on the slot's sixteen lanes --------------------------------------------
  // The launch lasts as long as its slowest wavefront, and that is one whose draw needs ~300 sequential steps before its covariance
  // stops moving (C2: median 16 steps, slowest of 4096 draws 312; 0.5 us per step with one wavefront on the SIMD) -- the lanes of a
  // slot replicate that chain sixteen times.  For a static model without missing observations the covariance sequence does not depend
  // on y: P_f(16 j - 1) = E16^(j - 1)(P_f(15)) with E16 the fourth square of the one-step element (A, C, J) of the filtered
  // covariance's Riccati map (kf_forward_b4c.hip has the derivation; here m <= 2 and everything is a register).  Lane j of the slot
  //   pass 1: runs ITS 16 steps from its own predicted covariance and a zero mean: transition Phi_j, response z_j, every F > 0, and
  //           its last predicted covariance against lane j + 1's start (1e-12 of the largest entry);
  //   the means are chained over the lanes (a_{16 (j + 1)} = Phi_j a_{16 j} + z_j, fifteen shuffles);
  //   pass 2: the same 16 steps from its true start, every output written straight from the registers (a lane's rows are
  //           consecutive: the lines complete in L2), the log-likelihood terms summed over the slot.
  // Chunk 0 is the sequential kernel's arithmetic bit for bit; the others agree with it to ~1e-14 (the doubling).  Any doubt -- a
  // missing observation in the first 256 steps, F <= 0, S0 <= 0, a failed check, in ANY slot of the wavefront -- and nothing has
  // been written: the loop below starts at step 0 as before.  KF_LANE_PRECHUNK=0 switches it off.
  if constexpr (SCAN) {
    constexpr int PL = 16, PN = 16 * PL;
    // (not ahead of the steady-state adjoint, KfFwdArgs::hs_ts: a slot then freezes behind step 256 at the earliest and the exact
    // adjoint sweeps >= 288 steps of EVERY draw -- measured at C2: forward 222 -> 128 us, kf_bwd_b4_kernel 203 -> 351)
    if (A.lane_pre && n >= 2 * PN && !A.hs_ts) {
      const int jj = jrep;
      const double* yb = A.y + (y_per_draw ? b * (long)n : 0L);
      double yc[PL];
      bool okl = true;
#pragma unroll
      for (int ss = 0; ss < PL; ++ss) {
        yc[ss] = yb[PL * jj + ss];
        okl = okl && !(yc[ss] != yc[ss]);
      }
      // one step of the recursion in the main loop's own arithmetic; returns F as computed (before any degenerate-step handling)
      auto kstep = [&](double (&Pq)[M][M], double (&aq)[M], double yv, double (&af)[M], double (&Pf)[M][M], double& quad,
                       double (&Kq)[M]) -> double {
        double Mv[M];
        double F = Hs, za = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          double sm = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) sm = fma(Pq[i][k], Z[k], sm);
          Mv[i] = sm;
          F = fma(Z[i], sm, F);
          za = fma(Z[i], aq[i], za);
        }
        const double v = (yv + dsign * ds) - za;
        const double invF = fast_rcp(F);
        quad = v * v * invF;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          Kq[i] = Mv[i] * invF;
          af[i] = fma(Kq[i], v, aq[i]);
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i; j < M; ++j) {
            double sm;
            if (univariate) {
              sm = Pq[i][j] - Kq[i] * Kq[j] * F;
            } else {
              sm = Pq[i][j] - Kq[i] * Mv[j];
              sm = fma(-Mv[i], Kq[j], sm);
              sm = fma(F * Kq[i], Kq[j], sm);
            }
            Pf[i][j] = sm;
            Pf[j][i] = sm;
          }
        double Wq[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
          double sm = c[i];
#pragma unroll
          for (int k = 0; k < M; ++k) sm = fma(T[i][k], af[k], sm);
          aq[i] = sm;
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double w = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) w = fma(T[i][k], Pf[k][j], w);
            Wq[i][j] = w;
          }
        }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i; j < M; ++j) {
            double sij = RQR[i][j], sji = RQR[j][i];
#pragma unroll
            for (int k = 0; k < M; ++k) {
              sij = fma(Wq[i][k], T[j][k], sij);
              sji = fma(Wq[j][k], T[i][k], sji);
            }
            const double sm = 0.5 * (sij + sji);
            Pq[i][j] = sm;
            Pq[j][i] = sm;
          }
        return F;
      };
      // small dense helpers (M <= 2)
      auto mul = [&](const double (&X)[M][M], const double (&Y)[M][M], double (&O)[M][M]) {   // O = X Y
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double sm = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(X[i][k], Y[k][j], sm);
            O[i][j] = sm;
          }
      };
      auto mul_t = [&](const double (&X)[M][M], const double (&Y)[M][M], double (&O)[M][M]) {   // O = X Y'
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double sm = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(X[i][k], Y[j][k], sm);
            O[i][j] = sm;
          }
      };
      auto tmul = [&](const double (&X)[M][M], const double (&Y)[M][M], double (&O)[M][M]) {   // O = X' Y
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double sm = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(X[k][i], Y[k][j], sm);
            O[i][j] = sm;
          }
      };
      auto inv_ipxy = [&](const double (&X)[M][M], const double (&Y)[M][M], double (&D)[M][M]) -> bool {   // D = (I + X Y)^-1
        double Wm[M][M];
        mul(X, Y, Wm);
#pragma unroll
        for (int i = 0; i < M; ++i) Wm[i][i] += 1.0;
        if constexpr (M == 1) {
          D[0][0] = 1.0 / Wm[0][0];
          return Wm[0][0] > 0.0;
        } else {
          const double det = Wm[0][0] * Wm[1][1] - Wm[0][1] * Wm[1][0];
          const double rd = 1.0 / det;
          D[0][0] = Wm[1][1] * rd;
          D[0][1] = -Wm[0][1] * rd;
          D[1][0] = -Wm[1][0] * rd;
          D[1][1] = Wm[0][0] * rd;
          return det > 0.0;   // (X, Y positive semi-definite: every eigenvalue of I + X Y is >= 1)
        }
      };
      auto symm = [&](double (&X)[M][M]) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i + 1; j < M; ++j) {
            const double sm = 0.5 * (X[i][j] + X[j][i]);
            X[i][j] = sm;
            X[j][i] = sm;
          }
      };
      // (i) P_f(15): the covariance recursion of steps 0 .. 15 (every lane of the slot, redundantly: 16 of the 256 steps)
      double a_in[M], P_in[M][M], Pf15[M][M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        a_in[i] = a[i];
#pragma unroll
        for (int j = 0; j < M; ++j) P_in[i][j] = P[i][j];
      }
      {
        double Pq[M][M], aq[M], af[M], Kq[M], quad;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          aq[i] = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) Pq[i][j] = P_in[i][j];
        }
        for (int ss = 0; ss < PL; ++ss) {
          const double F = kstep(Pq, aq, 0.0, af, Pf15, quad, Kq);
          okl = okl && (F > 0.0);
        }
      }
      // (ii) the one-step element, squared up; P_f(16 jj - 1) by the bits of jj - 1
      double Ae[M][M], Ce[M][M], Je[M][M];
      {
        double qz[M], zt[M];
        double S0 = Hs;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int k = 0; k < M; ++k) {
            s1 = fma(0.5 * (RQR[i][k] + RQR[k][i]), Z[k], s1);
            s2 = fma(Z[k], T[k][i], s2);
          }
          qz[i] = s1;
          zt[i] = s2;
          S0 = fma(Z[i], s1, S0);
        }
        okl = okl && (S0 > 0.0);
        const double rS0 = 1.0 / S0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            Ae[i][j] = fma(-qz[i] * rS0, zt[j], T[i][j]);
            Ce[i][j] = fma(-qz[i] * rS0, qz[j], 0.5 * (RQR[i][j] + RQR[j][i]));
            Je[i][j] = zt[i] * zt[j] * rS0;
          }
      }
      auto square = [&]() {
        double D[M][M], X[M][M], Y[M][M], V[M][M], Wm[M][M], N1[M][M];
        okl = inv_ipxy(Ce, Je, D) && okl;
        mul(Ae, D, X);          // A D
        mul(X, Ce, Y);          // A D C
        mul_t(Y, Ae, N1);       // A D C A'
        mul(Je, D, V);          // J D
        tmul(Ae, V, Wm);        // A' J D
        double N2[M][M], N3[M][M];
        mul(Wm, Ae, N2);        // A' J D A
        mul(X, Ae, N3);         // A D A
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            Ce[i][j] += N1[i][j];
            Je[i][j] += N2[i][j];
            Ae[i][j] = N3[i][j];
          }
        symm(Ce);
        symm(Je);
      };
      double Pfs[M][M];   // P_f(16 jj - 1)
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) Pfs[i][j] = Pf15[i][j];
      for (int q = 0; q < 4; ++q) square();   // E16
      const int pw = jj > 0 ? jj - 1 : 0;
      for (int q = 0; q < 4; ++q) {
        double D[M][M], Psi[M][M], Y[M][M], Nn[M][M];
        const bool dok = inv_ipxy(Pfs, Je, D);
        mul(Ae, D, Psi);
        mul(Psi, Pfs, Y);
        mul_t(Y, Ae, Nn);
        const bool bit = ((pw >> q) & 1) != 0;
        okl = okl && (dok || !bit);
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) Pfs[i][j] = bit ? Nn[i][j] + Ce[i][j] : Pfs[i][j];
        symm(Pfs);
        if (q < 3) square();   // E32, E64, E128
      }
      double Pst[M][M];   // the predicted covariance this lane's chunk starts from
      {
        double Wq[M][M];
        mul(T, Pfs, Wq);
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i; j < M; ++j) {
            double sij = RQR[i][j], sji = RQR[j][i];
#pragma unroll
            for (int k = 0; k < M; ++k) {
              sij = fma(Wq[i][k], T[j][k], sij);
              sji = fma(Wq[j][k], T[i][k], sji);
            }
            const double sm = 0.5 * (sij + sji);
            Pst[i][j] = (jj == 0) ? P_in[i][j] : sm;
            Pst[j][i] = (jj == 0) ? P_in[j][i] : sm;
          }
      }
      // (iii) pass 1: transition and zero-start response of the lane's 16 steps
      double Phi[M][M], zr[M];
      {
        double Pq[M][M], aq[M], af[M], Pf[M][M], Kq[M], quad;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          aq[i] = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) {
            Pq[i][j] = Pst[i][j];
            Phi[i][j] = (i == j) ? 1.0 : 0.0;
          }
        }
#pragma unroll
        for (int ss = 0; ss < PL; ++ss) {
          const double F = kstep(Pq, aq, yc[ss], af, Pf, quad, Kq);
          okl = okl && (F > 0.0);
          // a_{t+1} = (T - T K Z) a_t + ...: the transition of this step
          double TK[M], At[M][M], Nn[M][M];
#pragma unroll
          for (int i = 0; i < M; ++i) {
            double sm = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) sm = fma(T[i][k], Kq[k], sm);
            TK[i] = sm;
          }
#pragma unroll
          for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) At[i][j] = fma(-TK[i], Z[j], T[i][j]);
          mul(At, Phi, Nn);
#pragma unroll
          for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) Phi[i][j] = Nn[i][j];
        }
#pragma unroll
        for (int i = 0; i < M; ++i) zr[i] = aq[i];
        // the chunk's last predicted covariance against the next lane's start
        double pmax = 0.0, dmax = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            const double nx = slot_next(Pst[i][j]);
            pmax = fmax(pmax, fabs(Pq[i][j]));
            dmax = fmax(dmax, fabs(Pq[i][j] - nx));
          }
        if (jj < 15) okl = okl && (dmax <= 1e-12 * pmax);
      }
      if (__all(okl)) {   // (wave-uniform)
        // (iv) the predicted mean each chunk starts from
        double cur[M];
#pragma unroll
        for (int i = 0; i < M; ++i) cur[i] = a_in[i];
        for (int k = 0; k < 15; ++k) {
          double nx[M];
#pragma unroll
          for (int i = 0; i < M; ++i) {
            double sm = zr[i];
#pragma unroll
            for (int l = 0; l < M; ++l) sm = fma(Phi[i][l], cur[l], sm);
            nx[i] = sm;
          }
#pragma unroll
          for (int i = 0; i < M; ++i) {
            const double rv = slot_up(nx[i], std::integral_constant<int, 1>{});
            cur[i] = (jj == k + 1) ? rv : cur[i];
          }
        }
        // (v) pass 2: the lane's 16 steps for real.  Rows leave through an LDS stage four steps at a time: a lane's four rows of a
        // tape are adjacent in memory, so sixteen (covariances) or eight (states) consecutive lanes of a copy write one whole line --
        // straight from the registers every store instruction touched 64 lines (0.08 ms of partial-line traffic at C2)
        constexpr int PG = 4;                                  // steps per stage
        constexpr int ST_A = PG * M + 1, ST_P = PG * MM + 1, ST_L = PG + 1;   // odd lane strides: no bank conflicts
        double* g_fa = pstage;
        double* g_pa = g_fa + KF_WAVE * ST_A;
        double* g_fP = g_pa + KF_WAVE * ST_A;
        double* g_pP = g_fP + KF_WAVE * ST_P;
        double* g_ll = g_pP + KF_WAVE * ST_P;
        auto stage_out = [&](double* tape, const double* lds, int lstride, auto wtag, int rows, int rshift, int ss0) {
          constexpr int W = decltype(wtag)::value;   // doubles per row
          constexpr int CW = PG * W, NI = CW;        // 64 lane-chunks x CW doubles = CW rounds of 64
          double vals[NI];
#pragma unroll
          for (int k = 0; k < NI; ++k) {
            const int e = k * KF_WAVE + lane;
            const int cl = e / CW, wi = e - cl * CW;
            vals[k] = lds[cl * lstride + wi];
          }
#pragma unroll
          for (int k = 0; k < NI; ++k) {
            const int e = k * KF_WAVE + lane;
            const int cl = e / CW, wi = e - cl * CW;   // the stage of lane cl: slot cl / 16, chunk cl % 16 (ROWMAP)
            const long bb = b0 + (cl / 16);
            if (bb < A.B) tape[(bb * (long)rows + PL * (cl % 16) + ss0 + rshift) * W + wi] = vals[k];
          }
        };
        double Pq[M][M], aq[M], lacc = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          aq[i] = cur[i];
#pragma unroll
          for (int j = 0; j < M; ++j) Pq[i][j] = Pst[i][j];
        }
#pragma unroll
        for (int ss = 0; ss < PL; ++ss) {
          double af[M], Pf[M][M], Kq[M], quad;
          const double F = kstep(Pq, aq, yc[ss], af, Pf, quad, Kq);
          const double llv = (univariate && !exact_const && F == 1.0 && quad == 0.0) ? 0.0 : -0.5 * (KF_LOG_2PI + log(F) + quad);  // :497-500
          lacc += llv;
          const int sg = ss % PG;
          g_ll[lane * ST_L + sg] = llv;
#pragma unroll
          for (int i = 0; i < M; ++i) {
            g_fa[lane * ST_A + sg * M + i] = af[i];
            g_pa[lane * ST_A + sg * M + i] = aq[i];
#pragma unroll
            for (int j = 0; j < M; ++j) {
              g_fP[lane * ST_P + sg * MM + i * M + j] = Pf[i][j];
              g_pP[lane * ST_P + sg * MM + i * M + j] = Pq[i][j];
            }
          }
          if (sg == PG - 1) {
            const int ss0 = ss - (PG - 1);
            wave_lds_fence();
            if (A.ll_obs) stage_out(A.ll_obs, g_ll, ST_L, std::integral_constant<int, 1>{}, n, 0, ss0);
            if (w_fa) stage_out(A.filt_a, g_fa, ST_A, std::integral_constant<int, M>{}, n, 0, ss0);
            if (w_pa) stage_out(A.pred_a, g_pa, ST_A, std::integral_constant<int, M>{}, n + 1, 1, ss0);
            if (w_fP) stage_out(A.filt_P, g_fP, ST_P, std::integral_constant<int, MM>{}, n, 0, ss0);
            if (w_pP) stage_out(A.pred_P, g_pP, ST_P, std::integral_constant<int, MM>{}, n + 1, 1, ss0);
            wave_lds_fence();
          }
        }
        lacc = row_allsum(lacc);   // (ROWMAP: the slot's sixteen lanes)
        if (owner) smem[L::OFF_ACC + d] += lacc;
        // (vi) the loop below goes on from step 256 with the last lane's predicted moments
#pragma unroll
        for (int i = 0; i < M; ++i) {
          a[i] = slot_last(aq[i]);
#pragma unroll
          for (int j = 0; j < M; ++j) P[i][j] = slot_last(Pq[i][j]);
        }
        t = PN;
      }
    }
  }
  y_fetch(t);
  y_shift();
  y_commit();
  if (t + S < n) {
    y_fetch(t + S);
    y_shift();
  }
  double ycur = s_yb[0];
  // ---- steady state (kf_forward_b4.hip has the long version): once every draw of the wavefront has a predicted covariance that
  // stopped moving (3e-16 of its largest entry, tested at the end of each block of S steps, no missing observation ahead) the
  // gain, F and both covariances are constants and the loop at the end advances only the mean
  // Each LANE decides for its own draw (and from then on keeps its covariance where it is, so that the results of a draw do not
  // depend on its neighbours in the wavefront -- permuting a batch permutes its outputs bit for bit); the WAVEFRONT leaves the
  // full loop when all of its lanes have.
  const bool steady_en = SS && n >= 64;
  int t_free = 0;  // first step behind the last missing observation of this lane's draw
  if (steady_en) {
    constexpr int LPD = KF_WAVE / DPW;  // lanes that share a draw slot: lane, lane + DPW, ...
    int tn = -1;
    const double* yb = A.y + (y_per_draw ? b * (long)n : 0L);
    for (int tt = jrep; tt < n; tt += LPD) {
      const double yv = yb[tt];
      if (yv != yv) tn = tt;
    }
#pragma unroll
    for (int off = ROWMAP ? 1 : DPW; off < (ROWMAP ? 16 : KF_WAVE); off <<= 1) tn = max(tn, __shfl_xor(tn, off));
    t_free = tn + 1;
  }
  bool fz = false;
  int ts_own = n;   // first step of this lane's draw in the steady-state phase (KfFwdArgs::hs_ts: the steady-state adjoint's steps)
  double Pold[M][M];
  // DPW == 4 (SCAN): a frozen draw slot takes the sixteen steps of a block SIDE BY SIDE on its sixteen lanes.  The steady recursion
  // is affine,  a_{t+1} = A a_t + w_t,  A = T - (T K) Z,  w_t = (T K)(y_t -+ d) + c,  so  a_{t+j} = A^j a_t + sum_{i<j} A^{j-1-i} w_{t+i}:
  // a 4-level prefix scan over the lanes (shuffles within the slot; A^1, A^2, A^4, A^8 in registers), then every lane runs its own
  // step from its own a_{t+j} and writes its row of the LDS block.  (A is the closed-loop matrix of the converged filter: stable.)
  // A slot changes over at the block boundary behind ITS OWN convergence, whatever the other slots of the wavefront do (they keep
  // stepping with the frozen slots' lanes idle), so a draw's bits do not depend on its neighbours.
  const int jj = jrep;
  double zK[M], zTK[M], zAp[4][M][M], zAj[M][M], zInvF = 0.0, zFraw = 0.0;
  bool zDegen = false;
  auto freeze_consts = [&](int t_at) {
    double Mv[M], Pf[M][M];
    double F = Hs;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double sm = 0.0;
#pragma unroll
      for (int k = 0; k < M; ++k) sm = fma(P[i][k], Z[k], sm);
      Mv[i] = sm;
      F = fma(Z[i], sm, F);
    }
    zFraw = F;
    zDegen = univariate && F == 0.0;  // :468-471
    F = zDegen ? F + 1e-8 : F;
    if (!(F > 0.0) && info == 0 && !zDegen) info = ((t_at + 1) << 8) | 1;
    zInvF = zDegen ? 0.0 : fast_rcp(F);
#pragma unroll
    for (int i = 0; i < M; ++i) zK[i] = Mv[i] * zInvF;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = i; j < M; ++j) {
        double sm;
        if (univariate) {
          sm = P[i][j] - zK[i] * zK[j] * F;
        } else {
          sm = P[i][j] - zK[i] * Mv[j];
          sm = fma(-Mv[i], zK[j], sm);
          sm = fma(F * zK[i], zK[j], sm);
        }
        Pf[i][j] = sm;
        Pf[j][i] = sm;
      }
    // the covariance rows of a block never change again: written to the LDS chunks once
    if (w_fP || w_pP) {
      for (int ss = 0; ss < S; ++ss) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            if (w_fP) s_fP[ss * MM + i * M + j] = Pf[i][j];
            if (w_pP) s_pP[ss * MM + i * M + j] = P[i][j];
          }
      }
    }
    if constexpr (SCAN) {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double sm = 0.0;
#pragma unroll
        for (int l = 0; l < M; ++l) sm = fma(T[i][l], zK[l], sm);
        zTK[i] = sm;
      }
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int k = 0; k < M; ++k) {
          zAp[0][i][k] = fma(-zTK[i], Z[k], T[i][k]);
          zAj[i][k] = (i == k) ? 1.0 : 0.0;
        }
#pragma unroll
      for (int q = 1; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int k = 0; k < M; ++k) {
            double sm = 0.0;
#pragma unroll
            for (int l = 0; l < M; ++l) sm = fma(zAp[q - 1][i][l], zAp[q - 1][l][k], sm);
            zAp[q][i][k] = sm;
          }
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // A^j by the bits of j
        double Nx[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int k = 0; k < M; ++k) {
            double sm = 0.0;
#pragma unroll
            for (int l = 0; l < M; ++l) sm = fma(zAp[q][i][l], zAj[l][k], sm);
            Nx[i][k] = sm;
          }
        const bool bit = ((jj >> q) & 1) != 0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int k = 0; k < M; ++k) zAj[i][k] = bit ? Nx[i][k] : zAj[i][k];
      }
    }
  };
  // the `cnt` steps of the current block (its y is in the LDS chunk) for a frozen slot: rows -> LDS, a <- the state behind the block
  auto scan_block = [&](int cnt) {
    const double yt = s_yb[jj] + dsign * ds;
    double Sv[M];
#pragma unroll
    for (int i = 0; i < M; ++i) Sv[i] = fma(zTK[i], yt, c[i]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double rv[M];
#pragma unroll
      for (int i = 0; i < M; ++i)
        rv[i] = q == 0 ? slot_up(Sv[i], std::integral_constant<int, 1>{}) : q == 1 ? slot_up(Sv[i], std::integral_constant<int, 2>{})
              : q == 2 ? slot_up(Sv[i], std::integral_constant<int, 4>{}) : slot_up(Sv[i], std::integral_constant<int, 8>{});
      const bool has = jj >= (1 << q);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double sm = Sv[i];
#pragma unroll
        for (int l = 0; l < M; ++l) sm = fma(zAp[q][i][l], rv[l], sm);
        Sv[i] = has ? sm : Sv[i];
      }
    }
    double x[M], af[M], ap[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const double ev = slot_up(Sv[i], std::integral_constant<int, 1>{});
      double sm = (jj >= 1) ? ev : 0.0;
#pragma unroll
      for (int l = 0; l < M; ++l) sm = fma(zAj[i][l], a[l], sm);
      x[i] = sm;
    }
    double za = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) za = fma(Z[i], x[i], za);
    const double v = zDegen ? 0.0 : yt - za;
#pragma unroll
    for (int i = 0; i < M; ++i) af[i] = fma(zK[i], v, x[i]);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double sm = c[i];
#pragma unroll
      for (int k = 0; k < M; ++k) sm = fma(T[i][k], af[k], sm);
      ap[i] = sm;
    }
    if (jj < cnt) {
      s_lf[jj] = zFraw;
      s_lq[jj] = v * v * zInvF;
      s_lo[jj] = 1.0;
      if (w_fa) {
#pragma unroll
        for (int i = 0; i < M; ++i) s_fa[jj * M + i] = af[i];
      }
      if (w_pa) {
#pragma unroll
        for (int i = 0; i < M; ++i) s_pa[jj * M + i] = ap[i];
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) a[i] = slot_last(ap[i]);  // the next block starts from the last lane's prediction (S == 16 == the slot's lanes)
  };
  int npass = 0;  // consecutive convergence checks this lane's draw has passed
  for (; t < n; ++t) {
    const int s = t & (S - 1);
    if (steady_en && s == S - 1) {
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) Pold[i][j] = P[i][j];
    }
    const double yv = ycur;
    ycur = s_yb[s + 1];  // next step's y (slot S is the chunk's pad: re-read after the flush at a block boundary)
    const bool obs = !(yv != yv);
    double Mv[M];
    double F = Hs, za = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double sm = 0.0;
#pragma unroll
      for (int k = 0; k < M; ++k) sm = fma(P[i][k], Z[k], sm);
      Mv[i] = sm;
      F = fma(Z[i], sm, F);
      za = fma(Z[i], a[i], za);
    }
    const double Fraw = F;
    double v = (yv + dsign * ds) - za;
    // Univariate: a zero F skips the step (:468-471); the others process a missing step with F = 1, K = 0
    const bool degenerate = !obs || (univariate && F == 0.0);
    F = degenerate ? (univariate ? F + 1e-8 : 1.0) : F;
    if (!(F > 0.0) && info == 0 && !degenerate) info = ((t + 1) << 8) | 1;
    const double invF = degenerate ? 0.0 : fast_rcp(F);
    if (degenerate) v = 0.0;
    const double quad = v * v * invF;
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
        double sm;
        if (univariate) {
          sm = P[i][j] - K[i] * K[j] * F;
        } else {
          sm = P[i][j] - K[i] * Mv[j];
          sm = fma(-Mv[i], K[j], sm);
          sm = fma(F * K[i], K[j], sm);
        }
        Pf[i][j] = sm;
        Pf[j][i] = sm;
      }
    // predict
    double W[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double sm = c[i];
#pragma unroll
      for (int k = 0; k < M; ++k) sm = fma(T[i][k], af[k], sm);
      a[i] = (SCAN && fz) ? a[i] : sm;
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
        const double sm = (SS && fz) ? P[i][j] : 0.5 * (sij + sji);
        P[i][j] = sm;
        P[j][i] = sm;
      }
    // the step's rows and likelihood terms -> the LDS block (replicas write what the owner writes, to the same address)
    if (!(SCAN && fz)) {
      s_lf[s] = Fraw;
      s_lq[s] = quad;
      s_lo[s] = obs ? 1.0 : 0.0;
      if (w_fa) {
#pragma unroll
        for (int i = 0; i < M; ++i) s_fa[s * M + i] = af[i];
      }
      if (w_fP) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) s_fP[s * MM + i * M + j] = Pf[i][j];
      }
      if (w_pa) {
#pragma unroll
        for (int i = 0; i < M; ++i) s_pa[s * M + i] = a[i];
      }
      if (w_pP) {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) s_pP[s * MM + i * M + j] = P[i][j];
      }
    }
    if (s == S - 1) {
      if constexpr (SCAN) {
        if (__any(fz)) {
          if (fz) scan_block(S);
        }
      }
      flush(t - (S - 1), S);
      ycur = s_yb[0];
      if (steady_en && t + S < n) {
        double dmax = 0.0, pmax = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i; j < M; ++j) {
            dmax = fmax(dmax, fabs(P[i][j] - Pold[i][j]));
            pmax = fmax(pmax, fabs(P[i][j]));
          }
        bool still = true;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i; j < M; ++j) still = still && (P[i][j] == P[i][j]);  // (fmax drops NaNs)
        // (as in kf_forward_b4.hip: also every entry within 1e-12 of its own scale, and on two consecutive checks)
        bool each = true;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = i; j < M; ++j) {
            const double df = P[i][j] - Pold[i][j];
            each = each && (df * df <= A.ss_tol2 * fabs(P[i][i] * P[j][j]));
          }
        const bool pass = still && t >= t_free && dmax <= 3e-16 * pmax && each;
        npass = pass ? npass + 1 : 0;
        const bool newly = !fz && npass >= 2;
        if constexpr (SCAN) {
          if (newly) freeze_consts(t + 1);
        }
        fz = fz || newly;
        if (newly) ts_own = t + 1;
        if (__all(fz)) {
          ++t;
          break;
        }
      }
    }
  }
  if constexpr (SCAN) {
    if (t < n) {  // every slot of the wavefront is frozen (t is a multiple of S)
      for (; t < n; t += S) {
        const int cnt = (n - t) < S ? (n - t) : S;
        scan_block(cnt);
        if (cnt == S) flush(t, S);
      }
    } else if ((n % S) != 0 && __any(fz)) {  // the last, partial block of the slots that froze while others kept stepping
      if (fz) scan_block(n % S);
    }
  } else if (t < n) {
    // ---- steady-state steps t .. n-1 (t is a multiple of S), one after the other ----------------------------------------------------
    freeze_consts(t);
    ts_own = t;   // (every slot of the wavefront changes over here)
    const bool degenerate = zDegen;
    const double Fraw = zFraw, invF = zInvF;
    double K[M];
#pragma unroll
    for (int i = 0; i < M; ++i) K[i] = zK[i];
    for (; t < n; ++t) {
      const int s = t & (S - 1);
      const double yv = ycur;
      ycur = s_yb[s + 1];
      double za = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) za = fma(Z[i], a[i], za);
      const double v = degenerate ? 0.0 : (yv + dsign * ds) - za;
      double af[M];
#pragma unroll
      for (int i = 0; i < M; ++i) af[i] = fma(K[i], v, a[i]);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double sm = c[i];
#pragma unroll
        for (int k = 0; k < M; ++k) sm = fma(T[i][k], af[k], sm);
        a[i] = sm;
      }
      s_lf[s] = Fraw;
      s_lq[s] = v * v * invF;
      s_lo[s] = 1.0;
      if (w_fa) {
#pragma unroll
        for (int i = 0; i < M; ++i) s_fa[s * M + i] = af[i];
      }
      if (w_pa) {
#pragma unroll
        for (int i = 0; i < M; ++i) s_pa[s * M + i] = a[i];
      }
      if (s == S - 1) {
        flush(t - (S - 1), S);
        ycur = s_yb[0];
      }
    }
  }
  if (n % S) flush(n - n % S, n % S);

  // logp: the lanes' partial sums, S consecutive lanes per draw slot and round
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    double v = lsum[k];
#pragma unroll
    for (int off = S / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, KF_WAVE);
    const int e = lane + KF_WAVE * k;
    if ((lane % S) == 0 && ((DPW * S) % KF_WAVE == 0 || e < DPW * S)) smem[L::OFF_ACC + e / S] += v;
    wave_lds_fence();
  }
  if (owner && live) {
    A.logp[b] = smem[L::OFF_ACC + d];
    if (A.info) A.info[b] = info;
    if (A.hs_ts) A.hs_ts[b] = ts_own;
  }
}

}  // namespace

#define KF_CAT2(a, b) a##b
#define KF_CAT(a, b) KF_CAT2(a, b)
int KF_CAT(kf_launch_forward_lane_m, KF_LANE_M)(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  constexpr int M = KF_LANE_M;
  // draws per wavefront: enough wavefronts for every SIMD of the device first (the recursion is one
  // dependency chain per wavefront, replicas cost nothing), then wider wavefronts
  const long simds = (h && h->n_cu > 0 ? h->n_cu : 256) * 4L;
  const int dpw = a.B <= 4 * simds ? 4 : (a.B <= 16 * simds ? 16 : 64);
  const int mask = (a.filt_a ? 1 : 0) | (a.pred_a ? 2 : 0) | (a.filt_P ? 4 : 0) | (a.pred_P ? 8 : 0);
  KfFwdArgs a_ = a;
  {
    const char* sse = kf_env_switch("KF_STEADY_STATE");  // =0: step-by-step recursion (A/B measurements), as KF_FLAG_EXACT_RECURSION
    if (sse && *sse && atoi(sse) == 0) a_.flags |= KF_FLAG_EXACT_RECURSION;
  }
  const bool ss = !(a_.flags & KF_FLAG_EXACT_RECURSION) && a.T >= 64;
  {
    const char* e = kf_env_switch("KF_LANE_PRECHUNK");   // =0: every step one after the other from step 0 (A/B measurements)
    a_.lane_pre = (e && *e && atoi(e) == 0) ? 0 : 1;
  }
#define KF_LANE_LAUNCH2(D_, TM_)                                                                                             \
  do {                                                                                                                       \
    if (ss)                                                                                                                  \
      hipLaunchKernelGGL((kf_fwd_lane_kernel<M, D_, TM_, true>), dim3((unsigned)((a.B + D_ - 1) / D_)), dim3(KF_WAVE), 0, stream, a_);  \
    else                                                                                                                     \
      hipLaunchKernelGGL((kf_fwd_lane_kernel<M, D_, TM_, false>), dim3((unsigned)((a.B + D_ - 1) / D_)), dim3(KF_WAVE), 0, stream, a_); \
  } while (0)
#define KF_LANE_LAUNCH(D_)                                \
  do {                                                    \
    if (mask == 0) KF_LANE_LAUNCH2(D_, 0);                \
    else if (mask == 15) KF_LANE_LAUNCH2(D_, 15);         \
    else if (mask == 10) KF_LANE_LAUNCH2(D_, 10);         \
    else KF_LANE_LAUNCH2(D_, -1);                         \
  } while (0)
  if (dpw == 4) KF_LANE_LAUNCH(4);
  else if (dpw == 16) KF_LANE_LAUNCH(16);
  else KF_LANE_LAUNCH(64);
#undef KF_LANE_LAUNCH2
#undef KF_LANE_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
