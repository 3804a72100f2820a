// kf_smoother_b4.hip -- RTS smoother on v_mfma_f64_4x4x4_4b_f64 (one wavefront per draw, 5 <= k_states <= 15, static
// T, R, Q), with the inverse of the prediction covariance by WARM-STARTED Newton-Schulz iterations instead of a
// per-step elimination.  Companion of kf_forward_b4.hip / kf_backward_b4.hip (tile layout, MFMA conventions: kf_b4.h).
//
// Reference pymc_statespace/filters/kalman_smoother.py:56-104, backwards over t = n-2 .. 0:
//     P_hat = T P_f T' + R Q R'  (no symmetrisation, :99-104)       X = pinv(P_hat) T P_f
//     a_s[t] = a_f[t] + X' (a_s[t+1] - T a_f[t])    P_s[t] = P_f[t] + X' (P_s[t+1] - P_hat) X
// Why not eliminate: on register tiles a Gauss-Jordan pivot costs ~45 VALU instructions of row / column broadcasts,
// selects and a reciprocal around its six FMAs, ten pivots per step -- two thirds of kf_smoother_mfma.hip's 6.4 ms at the
// north-star shape.  But P_hat barely moves from one step to the next (the filter's covariance converges): with
// X_0 = the previous step's inverse,  R = I - P_hat X_0  is already below 1e-13 at most steps (median 0 iterations,
// mean 0.4 - 1.0 on the BASELINE workloads; scripts/r02/ns_smoother_proto.py), and where it is not
//     X <- X + X R,   R <- I - P_hat X        (Newton-Schulz / Ben-Israel-Cohen, quadratic)
// gets there in 1 - 5 steps; both are block-MFMA products.  A cold start (the first step, or ||R|| >= 1/2 -- the
// diffuse start of a series, reached last) uses X_0 = I / ||P_hat||_1 and takes ~10 iterations.  A draw whose
// iteration does not settle within 64 steps (P_hat singular or not safely positive definite: the reference's pinv
// case) is flagged and recomputed by kf_smooth_lds_kernel (Cholesky / Jacobi pinv) behind this kernel, as for
// kf_smoother_mfma.hip -- no host synchronisation.
//
// Tiles are N4 x N4, N4 = 4 NB >= m + 1: indices 0..m-1 states, JC = N4-1 a carrier of the MEAN (row / column = a,
// T_e[JC][JC] = 1), so that the four products also move the state vector:
//     U = T_e S_f            A = T_e U' + C_e             (row / column JC of A: a_hat = T a_f)
//     D = S_s[t+1] - A       X_e = blockdiag(P_hat^-1, 1/s) U^   (U^ = U on the state block, s at (JC, JC))
//     E = D X_e              S_s[t] = S_f + X_e' E        (row JC: a_s' = a_f' + (a_s[t+1] - a_hat)' X)
// The inverse runs on A^ = blockdiag(P_hat, s I) with s = ||P_hat||_1 of the last cold start, so the padding converges
// with the rest.  (Tried: a symmetric diagonal equilibration S P_hat S before the iteration -- +1.1 ms at the headline shape for
// its scale vectors in two layouts, and no better on a diffuse start, whose P_hat is ill-conditioned rather than badly scaled.)  Dynamic left factors (A^, X, D, X_e') get their replicated layout from one pass through an LDS tile.
#include "kf_b4.h"

namespace {

// LDS (doubles): five N4-row transpose tiles -- TH (A^: only its state block is rewritten per step, the padding at a cold
// start), TA (X_e), TX (the inverse, kept across steps, rewritten only when it changes), TU (U, parked while the inverse is
// iterated), TD (D, written in the same pass as A^) -- and the compact output image
template <int NB>
struct Sb4Lds {
  static constexpr int TILE_D = 4 * NB * TRS;
  static constexpr int IMG_D = NB <= 3 ? 128 : 240;  // m <= 11: 121 doubles; m <= 15: 225
  static constexpr int OFF_TX = TILE_D, OFF_TU = 2 * TILE_D, OFF_TD = 3 * TILE_D, OFF_TH = 4 * TILE_D, OFF_IMG = 5 * TILE_D,
                       OFF_IMA = OFF_IMG + IMG_D;
  static constexpr int TOTAL = OFF_IMA + 16 + (NB <= 3 ? 0 : 4 * TRS);  // (lanes of unused columns read up to 4 rows past a tile:
                                                                        //  into the next tile / the image, never used)
};
static_assert(Sb4Lds<3>::TOTAL <= 1280, "four wavefronts per SIMD need <= 10 KiB of LDS per wavefront");
constexpr int SB4_MAXIT = 64;

// max over the wavefront, in every lane: DPP row rotations, then the two lane-swap levels (no LDS)
__device__ __forceinline__ double wave_allmax_dpp(double x) {
  x = fmax(x, dpp_mov<0x128>(x));
  x = fmax(x, dpp_mov<0x124>(x));
  x = fmax(x, dpp_mov<0x122>(x));
  x = fmax(x, dpp_mov<0x121>(x));
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
  lo = __double2loint(x);
  hi = __double2hiint(x);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return fmax(__hiloint2double(d[0], c[0]), __hiloint2double(d[1], c[1]));
}
// The same for a convergence test, on the HIGH DWORDS of |x| only: sign cleared, the exponent and 20 mantissa bits of a
// double order like unsigned integers -- a norm to 1e-6 relative is all a threshold needs -- so each level is one
// v_max_u32 (DPP folded in) instead of two moves and a canonicalising v_max_f64, and Inf / NaN come out as the largest
// values (no separate NaN test).  Returns a wave-uniform lower bound of max |x| within 2^-20.
template <int NB>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(NB <= 3 ? 4 : 3))) void kf_smooth_b4_kernel(
    const KfSmoothArgs A, int32_t* flags, const int32_t* __restrict__ start, const int32_t* __restrict__ list) {
  constexpr int N4 = 4 * NB, JC = N4 - 1;
  using L = Sb4Lds<NB>;
  __shared__ __attribute__((aligned(16))) double smem[L::TOTAL];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15, q = lane & 3;
  const int fb = q * 4 + ((lane >> 2) & 3);
  // start / list (kf_smoother_b4_steady.hip): the steps from start[b] on are done -- the sweep resumes from the smoothed moments of
  // step start[b] (>= n: nothing was done) --, workgroup i takes draw list[i] (placement: the draws need start[b] steps each)
  const long b = list ? (long)list[blockIdx.x] : (long)blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m, r_ = A.r;
  const bool resume = start && start[b] < n;
  const int t0 = resume ? start[b] : n - 1;   // the step whose smoothed moments the sweep starts from
  if (resume && t0 == 0) {
    if (threadIdx.x == 0) flags[b] = 0;
    return;
  }
  double* TA = smem;
  double* TX = smem + L::OFF_TX;
  double* TU = smem + L::OFF_TU;
  double* TD = smem + L::OFF_TD;
  double* TH = smem + L::OFF_TH;
  double* IMG = smem + L::OFF_IMG;
  double* IMA = smem + L::OFF_IMA;

  const double* gT = A.Tm + b * (long)mm;
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  // ---- per-draw constants -------------------------------------------------------------------------------------------------
  double Ta[NB][NB];   // left-factor blocks of T_e: lane (g = k, any b, q = i) holds T_e[4I+q][4K+g]
  double Ce[NB];       // row-block registers of C_e = blockdiag(R Q R', 0)
  bool on_mm[NB];      // the state block
  bool on_dg[NB];      // the diagonal outside the state block (the padding of A^)
  bool on_tp[NB];      // tape image: where a tape value belongs
  const double* src[NB];
  int stp[NB];
  const double* gfa = A.filt_a + b * (long)n * m;
  const double* gfP = A.filt_P + b * (long)n * mm;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int rr = 4 * I + g;
    const bool on = rr < m && bq < m;
    on_mm[I] = on;
    on_dg[I] = rr == bq && rr >= m && bq < N4;
    double cv = 0.0;
    if (on) {
      for (int l = 0; l < r_; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r_; ++k) rq += gR[rr * r_ + k] * gQ[k * r_ + l];
        cv += rq * gR[bq * r_ + l];
      }
    }
    Ce[I] = cv;
#pragma unroll
    for (int K = 0; K < NB; ++K) {
      const int ti = 4 * I + q, tk = 4 * K + g;
      Ta[I][K] = (ti < m && tk < m) ? gT[ti * m + tk] : ((ti == JC && tk == JC) ? 1.0 : 0.0);
    }
    if (on) { src[I] = gfP + rr * m + bq; stp[I] = mm; on_tp[I] = true; }
    else if (rr < m && bq == JC) { src[I] = gfa + rr; stp[I] = m; on_tp[I] = true; }
    else if (rr == JC && bq < m) { src[I] = gfa + bq; stp[I] = m; on_tp[I] = true; }
    else { src[I] = gfa; stp[I] = 0; on_tp[I] = false; }  // (a finite value: such tile entries never reach an output, see below)
    src[I] += (long)(resume ? t0 - 1 : n - 1) * stp[I];
  }
  const bool on_jc = g == 3 && bq == JC;  // tile entry (JC, JC): DPP row 3 of the last register
  for (int e = lane; e < L::TOTAL; e += KF_WAVE) smem[e] = 0.0;
  wave_lds_fence();

  double PN[NB];
  auto prefetch = [&]() {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      PN[I] = *src[I];
      src[I] -= stp[I];
    }
  };
  // compact image of the tile's state block + its row JC -> the smoothed tapes, 16 bytes per lane
  const int nb128 = mm >> 1;
  auto store_moments = [&](int t, const double (&S)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g;
      if (rr < m && bq < m) IMG[rr * m + bq] = S[I];
      if (I == NB - 1 && rr == JC && bq < m) IMA[bq] = S[I];
    }
    wave_lds_fence();
    double* dst = A.smooth_P + (b * (long)n + t) * mm;
    const double2_t v2 = *reinterpret_cast<const double2_t*>(IMG + 2 * lane);
    if (lane < nb128) *(reinterpret_cast<double2_t*>(dst) + lane) = v2;
    else if ((mm & 1) && lane == nb128) dst[mm - 1] = v2.x;
    if (mm > 2 * KF_WAVE) {
      const double2_t w2 = *reinterpret_cast<const double2_t*>(IMG + 2 * (lane + KF_WAVE));
      if (lane + KF_WAVE < nb128) *(reinterpret_cast<double2_t*>(dst) + lane + KF_WAVE) = w2;
      else if ((mm & 1) && lane + KF_WAVE == nb128) dst[mm - 1] = w2.x;
    }
    if (lane < m) A.smooth_a[(b * (long)n + t) * m + lane] = IMA[lane];
    wave_lds_fence();
  };
  // row-block registers -> LDS tile (element (r, c) at r * TRS + 4 (c & 3) + (c >> 2))
  auto to_tile = [&](double* tile, const double (&X)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) tile[(4 * I + g) * TRS + fb] = X[I];
    wave_lds_fence();
  };
  // ---- last step: smoothed = filtered ----------------------------------------------------------------------------------------
  double Ps[NB];
  if (resume) {
    const double* gsa = A.smooth_a + (b * (long)n + t0) * m;
    const double* gsP = A.smooth_P + (b * (long)n + t0) * mm;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g;
      Ps[I] = (rr < m && bq < m) ? gsP[rr * m + bq] : ((rr < m && bq == JC) ? gsa[rr] : ((rr == JC && bq < m) ? gsa[bq] : 0.0));
    }
    prefetch();
  } else {
    prefetch();
#pragma unroll
    for (int I = 0; I < NB; ++I) Ps[I] = on_tp[I] ? PN[I] : 0.0;
    store_moments(n - 1, Ps);
    if (n > 1) prefetch();
  }

  double X[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) X[I] = 0.0;
  double s_pad = 1.0;
  bool have_x = false, bad = false;
  // The filter's covariance converges (kalman_filter.py:215-252 repeats one and the same update from then on); the tape then holds
  // the SAME P_f step after step, and with it P_hat, its inverse and X_e = P_hat^-1 T P_f are the last step's.  A step whose filtered
  // covariance is BIT FOR BIT the previous step's (the one after it in time) -- what a steady-state forward writes (C3, C4: constant
  // fills), and what an every-step recursion arrives at once it has reached its floating-point fixpoint -- reuses X_e (in TA, both
  // layouts) and the P_hat in TH: three products (U for T a_f, E, S_s) instead of six, no residual, no norm.  The full step would have
  // found its warm start's residual below 1e-13 and recomputed the same X_e from the same operands: the results do not change.
  // Any other step -- the start of a series, a missing observation, a tape that still moves in its last place -- takes the full path.
  // (Tolerances were tried first: 1e-15 of the largest entry, 32 steps in a row -- a fuzz case with cond(P_hat) ~ 1e3 came out 2.6e-11
  // away from every step in full, the inverse amplifies the drift; one unit in the last place of sqrt(P_ii P_jj), entry by entry -- its
  // LDS round trip for the diagonal cost the headline shape 8 %, where the tapes of an every-step forward rarely qualify anyway.)
  // KF_FLAG_EXACT_RECURSION: never.
  bool have_ref = false;
  bool st = false;

  for (int t = t0 - 1; t >= 0 && !bad; --t) {
    double Sf[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) Sf[I] = PN[I];  // (entries outside the tape image stay confined to unused rows / columns and (JC, JC))
    // ---- U = T_e S_f (parked in TU) ;  A = T_e U' + C_e ----------------------------------------------------------------------------
    double Ah[NB];
    {
      double U[NB];  // (first product of each chain: the literal 0 as C operand is an inline constant)
#pragma unroll
      for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][0], Sf[0], 0.0);
#pragma unroll
      for (int K = 1; K < NB; ++K)
#pragma unroll
        for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][K], Sf[K], U[I]);
      to_tile(TU, U);
      if (st) {
        // A without its product: the state block is the P_hat in TH, column JC is U's (T a_f), row JC its transpose
        const double urow = TU[bq * TRS + 4 * (JC & 3) + (JC >> 2)];  // U[bq][JC]
#pragma unroll
        for (int I = 0; I < NB; ++I) {
          const double ph = TH[(4 * I + g) * TRS + fb];
          Ah[I] = on_mm[I] ? ph : (bq == JC ? U[I] : ((I == NB - 1 && g == 3) ? urow : 0.0));
        }
      } else {
      const double* ys = TU + bq * TRS + 4 * g;
      const double2_t c01 = *reinterpret_cast<const double2_t*>(ys);
      double2_t c23 = {0.0, 0.0};
      if (NB > 2) c23 = *reinterpret_cast<const double2_t*>(ys + 2);
      const double ut[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
      for (int I = 0; I < NB; ++I) Ah[I] = mfma4(Ta[I][0], ut[0], Ce[I]);  // (C operand = C_e itself: no copy into the accumulator)
#pragma unroll
      for (int K = 1; K < NB; ++K)
#pragma unroll
        for (int I = 0; I < NB; ++I) Ah[I] = mfma4(Ta[I][K], ut[K], Ah[I]);
      }
    }
    // D = S_s[t+1] - A goes to its tile now: it rides along with the pass that replicates A^ (no round trip of its own later)
#pragma unroll
    for (int I = 0; I < NB; ++I) TD[(4 * I + g) * TRS + fb] = Ps[I] - Ah[I];
    // ---- X = A^^-1, A^ = blockdiag(P_hat, s I): residual of the previous inverse, Newton-Schulz where needed -------------------------
    double Rm[NB];
    double rnorm = 0.0;
    auto a_hat_tile = [&]() {  // the state block of A^ -> TH (its padding is written at a cold start, the rest stays zero)
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (on_mm[I]) TH[(4 * I + g) * TRS + fb] = Ah[I];
      wave_lds_fence();
    };
    auto residual = [&]() {  // R = I - A^ X (A^ as a left factor from TA, block row by block row); rnorm = max |R|
      unsigned mx = 0u;
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        const double* ls = TH + (4 * I + q) * TRS + 4 * g;
        const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
        double2_t d23 = {0.0, 0.0};
        if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
        const double ar[4] = {d01.x, d01.y, d23.x, d23.y};
        double rr_ = (4 * I + g == bq && bq < N4) ? 1.0 : 0.0;
#pragma unroll
        for (int K = 0; K < NB; ++K) rr_ = mfma4_nega(ar[K], X[K], rr_);
        Rm[I] = rr_;
        mx = max(mx, abs_hi(rr_));
      }
      rnorm = wave_absmax_hi(mx);  // (a NaN / Inf entry comes out as NaN / Inf: a diverged iteration cannot look converged)
    };
    auto rnorm_inf = [&]() {  // max absolute row sum of R: < 1 guarantees convergence (max |R_ij| does not)
      double rs_ = 0.0;
#pragma unroll
      for (int I = 0; I < NB; ++I) rs_ = fmax(rs_, row_allsum(fabs(Rm[I])));
      return uniform_f64(wave_allmax_dpp(rs_));
    };
    if (!st) {
    bool cold = !have_x;
    if (have_x) {
      a_hat_tile();
      residual();
      cold = !(rnorm < 1e-2) && !(rnorm_inf() < 0.7);
    }
    if (cold) {
      // X_0 = I / ||P_hat||_1 (P_hat is symmetric positive definite where this converges: the spectrum of I - A^ X_0 lies in
      // [0, 1 - lambda_min / ||P_hat||_1], about log2(cond) + 5 iterations -- half of what A^' / ||A^||^2 needs), s = ||P_hat||_1
      double rs_ = 0.0;
#pragma unroll
      for (int I = 0; I < NB; ++I) rs_ = fmax(rs_, row_allsum(on_mm[I] ? fabs(Ah[I]) : 0.0));
      const double n1 = uniform_f64(wave_allmax_dpp(rs_));
      if (!(n1 > 0.0) || !(n1 < 1e300)) bad = true;
      s_pad = n1;
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (on_dg[I]) TH[(4 * I + g) * TRS + fb] = s_pad;
      a_hat_tile();
      const double inv1 = 1.0 / n1;
#pragma unroll
      for (int I = 0; I < NB; ++I) X[I] = (4 * I + g == bq && bq < N4) ? inv1 : 0.0;
      to_tile(TX, X);
      residual();
    }
    int it = 0;
    while (!(rnorm <= 1e-13) && !bad) {
      // above 1e-7 one step of third order, X <- X (I + R + R^2) (residual R^3), for two products: where the filter has not
      // converged P_hat moves by 1e-6 .. 1e-2 of itself per step and the second-order step needed a second one, with a residual
      // product and a norm between them (measured at the headline shape: 2 - 4 steps at EVERY time step of the slowest draws,
      // which set the launch's duration).  TA is free here: X_e goes there at the end of the step.
      const bool third = !(rnorm <= 1e-7);
      if (third) {
        wave_lds_fence();
        to_tile(TA, Rm);
        double S2[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {  // S = R + R R (R as a left factor from TA)
          const double* ls = TA + (4 * I + q) * TRS + 4 * g;
          const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
          double2_t d23 = {0.0, 0.0};
          if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
          const double rl[4] = {d01.x, d01.y, d23.x, d23.y};
          double acc = Rm[I];
#pragma unroll
          for (int K = 0; K < NB; ++K) acc = mfma4(rl[K], Rm[K], acc);
          S2[I] = acc;
        }
#pragma unroll
        for (int I = 0; I < NB; ++I) Rm[I] = S2[I];
      }
#pragma unroll
      for (int I = 0; I < NB; ++I) {  // X <- X + X R (X as a left factor from TX)
        const double* ls = TX + (4 * I + q) * TRS + 4 * g;
        const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
        double2_t d23 = {0.0, 0.0};
        if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
        const double xr[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
        for (int K = 0; K < NB; ++K) X[I] = mfma4(xr[K], Rm[K], X[I]);
      }
      wave_lds_fence();
      to_tile(TX, X);
      if (rnorm <= 2e-5) break;  // the new residual is rnorm^2 (below 1e-7) or rnorm^3 (+ the rounding floor) -- not worth a product
      const double rprev = rnorm;
      residual();
      // the rounding floor of I - A^ X (~ cond(P_hat) eps) is reached: accepted up to cond ~ 1e8 -- where the elimination
      // kernel's pivot test (1e-9 x the largest diagonal entry) starts to flag as well
      if (rnorm > 0.5 * rprev && rnorm <= 1e-7) break;
      if (++it > SB4_MAXIT || !(rnorm == rnorm)) bad = true;
    }
    // A numerically singular P_hat (cond >~ 1e10: rank-deficient R Q R' behind a long series) is not for this kernel: its residual
    // test passes -- |I - P_hat X| stays small in the directions that matter least -- while the error of X along the near-null
    // directions makes the smoothed recursion itself unstable (a fuzz case with cond = 1.4e12, profiles/r04/fuzz_summary_b.txt: the
    // smoothed covariance grew fivefold per step to 1e30 where the reference's pinv stays at 1e0).  ||X||_inf ||P_hat||_1 (the norm of
    // the last cold start: P_hat moves slowly) at every cold start and every 32nd step (at every step that iterated it cost the
    // headline 5 %: the slow draws iterate at every step and set the launch's length; a flagged draw is redone as a whole, so when the
    // flag is raised does not matter); such a draw goes to the pseudo-inverse kernel behind this one.
#ifndef KF_NO_COND_GUARD
    if ((cold || (t & 31) == 0) && !bad) {
      double xs_ = 0.0;
#pragma unroll
      for (int I = 0; I < NB; ++I) xs_ = fmax(xs_, row_allsum(on_mm[I] ? fabs(X[I]) : 0.0));
      if (!(uniform_f64(wave_allmax_dpp(xs_)) * s_pad < 1e10)) bad = true;
    }
#endif
    have_x = true;
    if (bad) break;
    have_ref = !A.exact;
    }
    if (t > 0) prefetch();
    // ---- X_e = X U^ ;  E = D X_e ;  S_s = S_f + X_e' E ---------------------------------------------------------------------------
    double E[NB], Xe[NB];
    if (st) {
#pragma unroll
      for (int I = 0; I < NB; ++I) Xe[I] = TA[(4 * I + g) * TRS + fb];
    } else {
      double uh[NB];
#pragma unroll
      for (int K = 0; K < NB; ++K) {  // U^: the state block of U (back from TU), s at (JC, JC)
        const double uk = TU[(4 * K + g) * TRS + fb];
        uh[K] = on_mm[K] ? uk : ((K == NB - 1 && on_jc) ? s_pad : 0.0);
      }
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        const double* ls = TX + (4 * I + q) * TRS + 4 * g;
        const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
        double2_t d23 = {0.0, 0.0};
        if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
        const double xr[4] = {d01.x, d01.y, d23.x, d23.y};
        double acc = mfma4(xr[0], uh[0], 0.0);
#pragma unroll
        for (int K = 1; K < NB; ++K) acc = mfma4(xr[K], uh[K], acc);
        Xe[I] = acc;
      }
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const double* ls = TD + (4 * I + q) * TRS + 4 * g;
      const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
      double2_t d23 = {0.0, 0.0};
      if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
      const double dr[4] = {d01.x, d01.y, d23.x, d23.y};
      double acc = mfma4(dr[0], Xe[0], 0.0);
#pragma unroll
      for (int K = 1; K < NB; ++K) acc = mfma4(dr[K], Xe[K], acc);
      E[I] = acc;
    }
    wave_lds_fence();
    if (!st) to_tile(TA, Xe);
    // the next step (t - 1, its tile is on its way into PN): may it reuse this step's X_e?
    st = false;
    if (have_ref && t > 0) {
      bool off = false;
#pragma unroll
      for (int I = 0; I < NB; ++I) off |= on_mm[I] && !(PN[I] == Sf[I]);
      st = __ballot(off) == 0;
    }
#pragma unroll
    for (int K = 0; K < NB; ++K) {  // left factor (X_e')(I, K): lane (g = k, any b, q = i) holds X_e[4K+g][4I+q]
      const double* ls = TA + (4 * K + g) * TRS + 4 * q;
      const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
      double2_t d23 = {0.0, 0.0};
      if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
      const double LK[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
      for (int I = 0; I < NB; ++I) Ps[I] = mfma4(LK[I], E[K], K == 0 ? Sf[I] : Ps[I]);  // (S_f as the first C operand)
    }
    wave_lds_fence();
    store_moments(t, Ps);
  }
  if (lane == 0) flags[b] = bad ? 1 : 0;
}

}  // namespace

bool kf_smoother_b4_ok(int m, int r) { return m >= 5 && m + 1 <= 16 && r >= 1; }

int kf_launch_smoother_b4(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
  if (!kf_smoother_b4_ok(a.m, a.r) || a.flags || !workspace || workspace_bytes < kf_smoother_mfma_workspace_bytes(a.B)) {
    KF_SET_ERR(h, "smoother(b4): needs 5 <= m <= 15, static matrices and %zu bytes of workspace",
               kf_smoother_mfma_workspace_bytes(a.B));
    return KF_ERR_UNSUPPORTED;
  }
  int32_t* flags = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  // behind the filter's convergence point: the time-parallel kernels (kf_smoother_b4_steady.hip), when the caller's workspace holds them
  int32_t *start = nullptr, *list = nullptr;
  const size_t base = kf_smoother_mfma_workspace_bytes(a.B);
  if (kf_smoother_b4_split_ok(a) && workspace_bytes >= base + kf_smoother_b4_split_workspace_bytes(a.B, a.m)) {
    const int rc = kf_launch_smoother_b4_steady(h, a, reinterpret_cast<char*>(workspace) + base, &start, &list, stream);
    if (rc) return rc;
  }
  const int nb = (a.m + 1 + 3) / 4;
#define KF_SB4(NB_) \
  hipLaunchKernelGGL((kf_smooth_b4_kernel<NB_>), dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, flags, start, list)
  switch (nb) {
    case 2: KF_SB4(2); break;
    case 3: KF_SB4(3); break;
    default: KF_SB4(4); break;
  }
#undef KF_SB4
  KF_HIP_CHECK(h, hipGetLastError());
  // draws whose iteration did not settle or whose P_hat is numerically singular: first the elimination kernel (pivoted Gauss-Jordan on
  // register tiles: a nearly singular but invertible P_hat is its everyday case, and it is fast -- the pseudo-inverse kernel's Jacobi
  // sweeps took 8 ms for four flagged draws of 4096 at m = 13 x 300 steps, the whole launch 3.9); what IT flags (pivots below 1e-9:
  // singular, the reference's pinv case) goes on to the pseudo-inverse kernel.  Draws without a flag exit both at once.
  KfSmoothArgs fbk = a;
  fbk.only_flagged = flags;
  return kf_launch_smoother_mfma(h, fbk, workspace, workspace_bytes, stream);
}
