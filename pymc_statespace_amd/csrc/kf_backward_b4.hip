// kf_backward_b4.hip -- reverse-mode gradient of the Kalman log-likelihood as the adjoint of the AUGMENTED recursion
// kf_forward_b4.hip runs (one wavefront per draw, p == 1, k_states + 2 <= 16, static system matrices), on
// v_mfma_f64_4x4x4_4b_f64.  Companion of kf_forward_b4.hip (layout, MFMA conventions: its header and kf_b4.h).
//
// The reference has no hand-written gradient (PyTensor autodiff of the scan, core/statespace.py:174); this is the
// adjoint of the reference operation sequence (filters/kalman_filter.py:215-223, 255-351) written on the second-moment
// tile S of (alpha, Z alpha, "1") -- indices 0..m-1 states, JA = Z alpha, JC = carrier of the mean (entries 2 a):
//     u = S e_A + yo e_C        yo = -2 (y -+ d) when observed         F = H + u_A        rs = observed / sqrt(F)
//     S_f = S - (rs u)(rs u)'   S' = T_e S_f T_e' + C_e                ll = -1/2 (log 2 pi + log F + (rs u_C)^2 / 4)
// With G = d L / d S' (kept symmetric) one step backwards is
//     C_b += G      Y = T_e' G      G_f = T_e' Y'      T_b += 2 Y' S_f                     (three m^3 products)
//     g = -2 G_f (rs u) - (gl / 4)(rs u_C) e_C        u_b = rs g        rs_b = g . u
//     F_b = -gl / (2 F) - rs_b rs / (2 F)             H_b += F_b        d_b += -2 dsign u_b[C]
//     G  <- G_f + (u_b e_A' + e_A u_b') / 2 + F_b e_A e_A'
// (derivation and a numpy restatement checked against torch.autograd: scripts/r02/aug_adjoint_proto.py).  The update's
// gain, innovation and Joseph form never appear: the whole update is one rank-one term, as in the forward kernel.
// Per step: 40 block MFMAs (~410 cycles of the fp64 pipe) and ~70 VALU instructions, against 9 16x16x4 MFMAs (585
// cycles) and ~200 VALU in kf_backward_mfma.hip.
//
// T_e and C_e are functions of (T, Z, R, Q, c), S_0 of (a0, P0, Z): the chain rule back to the reference's inputs runs
// once per draw after the sweep.  The sweep stops after step 1 and de-augments G_1 into the adjoints (a_b, P_b) of
// (a_pred[1], P_pred[1]); step 0 is taken by kf_backward_mfma.hip in the reference's own variables (KfBwdArgs::init_ab /
// init_Pb / accumulate), because the ENTRYWISE gradient with respect to a non-symmetric P0 -- what autodiff of the
// reference graph returns -- is not a function of the symmetric tile.
#include <stdlib.h>

#include "kf_b4.h"

namespace {

constexpr int BB_OFF_UX = 16 * TRS;            // vector exchange by column index (16) | zeros (4)
constexpr int BB_OFF_YB = BB_OFF_UX + 20;      // y of a block of 64 steps
constexpr int BB_OFF_GL = BB_OFF_YB + 64;      // g_ll_obs of the block
constexpr int BB_OFF_CB = BB_OFF_GL + 64;      // C_e adjoint accumulators, NB x 64 (touched once per step: LDS, not VGPRs)
constexpr int BB_LOOP_DOUBLES = BB_OFF_CB + 4 * KF_WAVE;
static_assert(BB_LOOP_DOUBLES <= 1280, "four wavefronts per SIMD need <= 10 KiB of LDS per wavefront");

__host__ __device__ inline int bb_epilogue_doubles(int n4, int m, int r) {
  return 3 * n4 * n4 + 2 * m * m + m * r + 8 * m + 8;  // EB, EC, EG | RQR, Ms | Ms R | vectors
}

// TILE: the tape is kf_forward_b4.hip's own tile tape (KfBwdArgs::tile_tape, kf_filter_value_and_grad): slot t holds S_f of
// step t in the layout the registers want it, with u_hat in its (otherwise unused) row JA and 1 / sqrt(F) at [JA][JA] --
// no tape image to assemble, no matrix-vector product, no rsqrt, no rank-one MFMAs; slot T holds the predicted tile S_1
#ifndef KF_BWD_AHEAD
#define KF_BWD_AHEAD 2
#endif
template <int NB, bool TILE>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu((NB <= 3 && KF_BWD_AHEAD <= 2) ? 4 : 3))) void kf_bwd_b4_kernel(
    const KfBwdArgs A, double* __restrict__ scratch) {
  constexpr int N4 = 4 * NB, JA = N4 - 2, JC = N4 - 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15, q = lane & 3;
  const int fb = q * 4 + ((lane >> 2) & 3);  // position of column bq in a transpose-tile row / the exchange vector
  const long b = A.list ? (long)A.list[blockIdx.x] : (long)blockIdx.x;
  // per-draw step counts (the steady-state adjoint took the steps behind ts = steps[b], kf_backward_steady.hip): the sweep runs
  // steps ts-1 .. 1 from G = d L / d S_ts, the de-augmentation's inverse of (a_b, P_b) = (init_ab, 0); tapes / y stay strided by
  // tape_T / y_T and indexed by the absolute step
  const int n = A.steps ? (A.steps[b] < A.T ? A.steps[b] : A.T) : A.T;
  const int m = A.m, mm = m * m, r_ = A.r;

  double* TR = smem;
  double* UX = smem + BB_OFF_UX;
  double* YB = smem + BB_OFF_YB;
  double* GL = smem + BB_OFF_GL;
  double* CB = smem + BB_OFF_CB;

  const double* gT = A.Tm + b * (long)mm;
  const double* gZ = A.Z + b * (long)m;

  // ---- per-draw constants ------------------------------------------------------------------------------------------------
  // z T (row JA of T_e) through LDS
  if (lane < m) {
    double s = 0.0;
    for (int l = 0; l < m; ++l) s += gZ[l] * gT[l * m + lane];
    TR[lane] = s;
  }
  wave_lds_fence();
  auto Te = [&](int rr, int cc) -> double {  // T_e[rr][cc]
    if (cc < m) return rr < m ? gT[rr * m + cc] : (rr == JA ? TR[cc] : 0.0);
    return (rr == JC && cc == JC) ? 1.0 : 0.0;
  };
  double TtA[NB][NB];  // left-factor blocks of T_e': lane (g = k, any b, q = i) holds T_e'[4I+i][4K+k] = T_e[4K+g][4I+q]
  double zArow[NB];    // z as a one-row left factor: lane (g = k, any b, q = 0) holds z[4K+k]
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    const int k = 4 * K + g;
    zArow[K] = (q == 0 && k < m) ? gZ[k] : 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I) TtA[I][K] = Te(k, 4 * I + q);
  }
  wave_lds_fence();
  const double zc = (g == 0 && bq < m) ? gZ[bq] : 0.0;
  double H1 = A.H[b];
  asm volatile("" : "+v"(H1));
  const double d1 = A.d ? A.d[b] : 0.0;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double selA = (lane == JA) ? 1.0 : 0.0, selC = (lane == JC) ? 1.0 : 0.0;  // row-0 layout: entry JA / JC of a vector
  const double eA_row0 = selA;                                                      // e_A as a right factor (k = 0)
  const double eA_rep = (g == 0 && q == (JA & 3)) ? 1.0 : 0.0;                      // e_A as a left factor, block row NB-1
  const double mAA = (g == (JA & 3) && bq == JA) ? 1.0 : 0.0;                       // tile entry (JA, JA), register NB-1

  // tape image: lane (register I) reads P[rr][cc], or 2 a[rr] in column JC, or 2 a[cc] in row JC
  // (every other lane reads some finite tape value: those tile entries only ever meet exact zeros of T_e / z, see below)
  const long nt1 = (long)A.tape_T + 1;
  const double* gpa = TILE ? nullptr : A.pred_a + b * nt1 * m;
  const double* gpP = TILE ? nullptr : A.pred_P + b * nt1 * mm;
  const double* gtl = TILE ? A.tile_tape + b * nt1 * (N4 * N4) : nullptr;  // [T+1][N4][N4]
  const double* src[NB];
  int stp[NB], sh[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int rr = 4 * I + g;
    if (TILE) { src[I] = gtl + (bq < N4 ? rr * N4 + bq : 0); stp[I] = N4 * N4; sh[I] = 0; }
    else if (rr < m && bq < m) { src[I] = gpP + rr * m + bq; stp[I] = mm; sh[I] = 0; }
    else if (rr < m && bq == JC) { src[I] = gpa + rr; stp[I] = m; sh[I] = 1; }
    else if (rr == JC && bq < m) { src[I] = gpa + bq; stp[I] = m; sh[I] = 1; }
    else { src[I] = gpa; stp[I] = 0; sh[I] = 0; }
    src[I] += (long)(n - 1) * stp[I];
  }
  // the tape is read TWO steps ahead, into two register sets the (manually unrolled) loop alternates between: a tape line comes from
  // HBM (the forward kernel wrote n x 1.2 KB per draw, long evicted from L2), ~2 us under load, and with the load issued one step ahead
  // a wavefront running alone on its SIMD -- the tail of every launch, and all of a launch of <= 1024 draws -- waited for it at the
  // top of every step (0.72 us per step against the forward kernel's 0.48)
  // (the compact-tape instantiations have no registers for the second set: one step ahead, as before)
  constexpr int AHEAD = TILE ? KF_BWD_AHEAD : 1;
  double PR[AHEAD][NB];
  auto prefetch = [&](double (&PN)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      PN[I] = *src[I];
      src[I] -= stp[I];
    }
  };
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)A.y_T : 0L);
  const double* gll = A.g_ll_obs ? A.g_ll_obs + b * (long)A.y_T : nullptr;
  const double g_all = A.g_logp ? A.g_logp[b] : 0.0;
  auto stage_block = [&](int t) {  // y and g_ll_obs of the 64-step block that holds t
    const int t0 = t & ~63;
    const int cnt = (n - t0 < 64 ? n - t0 : 64);
    wave_lds_fence();
    if (lane < cnt) {
      YB[lane] = ybase[t0 + lane];
      GL[lane] = gll ? gll[t0 + lane] : 0.0;
    }
    wave_lds_fence();
  };
  if (lane < 20) UX[lane] = 0.0;
  for (int e = lane; e < 16 * TRS; e += KF_WAVE) TR[e] = 0.0;
  wave_lds_fence();

  // (the tile-tape sweep has registers to spare at NB <= 3: its C_e accumulators stay in VGPRs)
  constexpr bool CB_REG = TILE && NB <= 3;
  double G[NB], Tb[NB], Cbr[CB_REG ? NB : 1];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    G[I] = Tb[I] = 0.0;
    if (CB_REG) Cbr[CB_REG ? I : 0] = 0.0;
    else CB[I * KF_WAVE + lane] = 0.0;
  }
  if (A.init_ab) {
    // a enters the tile as S[i][JC] = S[JC][i] = 2 a_i: with G symmetric, a_b[i] = 4 G[i][JC] (the epilogue below); the derived
    // entries (row / column JA) carry no adjoint of their own -- any extension of the objective off the consistent tiles has the
    // same total derivative
    const double* ib = A.init_ab + b * (long)m;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g;
      if (rr < m && bq == JC) G[I] = 0.25 * ib[rr];
      else if (rr == JC && bq < m) G[I] = 0.25 * ib[bq];
    }
  }
  double Hb = 0.0, dacc = 0.0;

  // vector in row-0 layout (lanes g == 0: entry bq) -> left-factor layouts through the exchange buffer
  // (lanes outside the factor's support read the four zeros behind the vector)
  const double* ux_rep = (g == 0) ? UX + 4 * q : UX + 16;   // {v[q], v[4+q], v[8+q], v[12+q]}: block I of v replicated
  const double* ux_row = (q == 0) ? UX + 4 * g : UX + 16;   // {v[g], v[4+g], v[8+g], v[12+g]}: one-row left factor

  prefetch(PR[0]);
#pragma unroll
  for (int j = 1; j < AHEAD; ++j)
    if (n - 1 - j >= 1) prefetch(PR[j]);
  stage_block(n - 1);
  auto step = [&](const int t, double (&PN)[NB]) __attribute__((always_inline)) {
    double Sx[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) Sx[I] = TILE ? PN[I] : ldexp(PN[I], sh[I]);
    if ((t & 63) == 63 && t != n - 1) stage_block(t);
    const double yv = YB[t & 63];
    const double gl = g_all + GL[t & 63];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      if (CB_REG) Cbr[CB_REG ? I : 0] += G[I];
      else CB[I * KF_WAVE + lane] += G[I];
    }
    double u, ut, rs, invF;   // u: what goes through the exchange (u itself, or u_hat with TILE); ut = u_hat
    bool obe;
    if (TILE) {
      // row JA of the taped tile = (u_hat, rs at JA) in DPP row 2 of the last register (the lane swap must run in EVERY lane:
      // not inside a select).  Entry JA of the vector only ever meets exact zeros (row / column JA of G_f)
      const double hh = high_half_everywhere(Sx[NB - 1]);
      ut = (g == 0 && bq < N4) ? hh : 0.0;
      rs = readlane_f64(ut, JA);
      invF = rs * rs;
      obe = true;  // (a missing / skipped step has rs = 0 and u_hat = 0 on the tape: every term vanishes by itself)
      u = ut;
    } else {
      const bool ob = !(yv != yv);
      const double yo = ob ? -2.0 * (yv + dsign * d1) : 0.0;
      // ---- u = S e_A + yo e_C from the tape image: r = z S (row-0 layout), u_A = r . z -----------------------------------
      double r = 0.0;
#pragma unroll
      for (int K = 0; K < NB; ++K) r = mfma4(zArow[K], Sx[K], r);
      const double uA = readlane_f64(row_allsum(r * zc), 0);
      const double F = H1 + uA;
      obe = ob && !(univariate && F == 0.0);  // Univariate skips a zero F (:468-471)
      rs = obe ? fast_rsqrt(fmax(F, 1e-300)) : 0.0;
      invF = rs * rs;
      u = fma(selC, yo, (lane == JA) ? uA : r);
      ut = u * rs;
    }
    if (g == 0) UX[fb] = u;
    // ---- Y = T_e' G ------------------------------------------------------------------------------------------------------
    double Y[NB];  // (the first product of each chain takes the literal 0 as its C operand: an inline constant, no v_mov)
#pragma unroll
    for (int I = 0; I < NB; ++I) Y[I] = mfma4(TtA[I][0], G[0], 0.0);
#pragma unroll
    for (int K = 1; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I) Y[I] = mfma4(TtA[I][K], G[K], Y[I]);
#pragma unroll
    for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = Y[I];
    wave_lds_fence();
    // one LDS round trip: u as both left factors, Y' as row-block registers and, block column by block column, as a left factor
    double urep[NB], urow[NB], Yt[NB];
    {
      const double2_t a01 = *reinterpret_cast<const double2_t*>(ux_rep);
      const double2_t b01 = *reinterpret_cast<const double2_t*>(ux_row);
      const double* ys = TR + bq * TRS + 4 * g;
      const double2_t c01 = *reinterpret_cast<const double2_t*>(ys);
      double2_t a23 = {0.0, 0.0}, b23 = {0.0, 0.0}, c23 = {0.0, 0.0};
      if (NB > 2) {
        a23 = *reinterpret_cast<const double2_t*>(ux_rep + 2);
        b23 = *reinterpret_cast<const double2_t*>(ux_row + 2);
        c23 = *reinterpret_cast<const double2_t*>(ys + 2);
      }
      const double ua[4] = {a01.x, a01.y, a23.x, a23.y}, ub4[4] = {b01.x, b01.y, b23.x, b23.y},
                   yc[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        urep[I] = ua[I];
        urow[I] = ub4[I];
        Yt[I] = yc[I];
      }
    }
    // ---- S_f = S - ut ut' (rows < m and JC, columns < m are what T_b needs) --------------------------------------------------
    double Sf[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) Sf[I] = TILE ? Sx[I] : mfma4_nega(urep[I] * rs, ut, Sx[I]);
    // ---- G_f = T_e' Y' ;  T_b += Y' S_f   (left factor (Y')(I, K): lane (g = k, any b, q = i) holds Y[4K+g][4I+q]) -------------------
    double Gf[NB];
#pragma unroll
    for (int K = 0; K < NB; ++K) {
      const double* ls = TR + (4 * K + g) * TRS + 4 * q;
      const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
      double2_t d23 = {0.0, 0.0};
      if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
      const double LK[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        Gf[I] = mfma4(TtA[I][K], Yt[K], K == 0 ? 0.0 : Gf[I]);
        Tb[I] = mfma4(LK[I], Sf[K], Tb[I]);
      }
    }
    wave_lds_fence();
    if (t > AHEAD) prefetch(PN);   // step t - AHEAD
    // ---- g = -2 G_f ut - (gl / 4) ut_C e_C (row-0 layout) -------------------------------------------------------------------------
    const double m2rs = TILE ? -2.0 : -2.0 * rs;  // (TILE: u_hat itself went through the exchange)
    double gv = mfma4(urow[0] * m2rs, Gf[0], 0.0);
#pragma unroll
    for (int K = 1; K < NB; ++K) gv = mfma4(urow[K] * m2rs, Gf[K], gv);
    const double utC = readlane_f64(ut, JC);
    const double gut = fma(selC, obe ? -0.25 * gl * utC : 0.0, gv);
    // rs_b rs = (g . u) rs = g . u_hat
    const double rsbrs = TILE ? readlane_f64(row_allsum(gut * ut), 0) : readlane_f64(row_allsum(gut * u), 0) * rs;
    const double Fb = obe ? -0.5 * invF * (gl + rsbrs) : 0.0;
    Hb += Fb;
    const double ub = rs * gut;
    dacc += ub;
    // ---- G <- G_f + (u_b e_A' + e_A u_b') / 2 + F_b e_A e_A' ------------------------------------------------------------------------
    if (g == 0) UX[fb] = ub;
    wave_lds_fence();
    double brep[NB];
    {
      const double2_t a01 = *reinterpret_cast<const double2_t*>(ux_rep);
      double2_t a23 = {0.0, 0.0};
      if (NB > 2) a23 = *reinterpret_cast<const double2_t*>(ux_rep + 2);
      const double ba[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
      for (int I = 0; I < NB; ++I) brep[I] = ba[I];
    }
    wave_lds_fence();
#pragma unroll
    for (int I = 0; I < NB; ++I) G[I] = mfma4(0.5 * brep[I], eA_row0, Gf[I]);
    G[NB - 1] = mfma4(eA_rep, 0.5 * ub, G[NB - 1]);
    G[NB - 1] = fma(mAA, Fb, G[NB - 1]);
  };
  {
    int t = n - 1;
    for (; t >= AHEAD; t -= AHEAD) {
#pragma unroll
      for (int j = 0; j < AHEAD; ++j) step(t - j, PR[j]);
    }
    if constexpr (AHEAD == 2) {
      if (t == 1) step(1, PR[0]);
    } else {
#pragma unroll
      for (int j = 0; j < AHEAD - 1; ++j)
        if (t - j >= 1) step(t - j, PR[j]);
    }
  }

  // ---- after the sweep: G = d L / d S_1.  Chain rule to (T, Z, R, Q, H, c, d), de-augmentation to (a_b, P_b) ------------------
  const bool acc = A.accumulate != 0;  // (the steady-state adjoint's shares are in the outputs already)
  double cb_fin[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) cb_fin[I] = CB_REG ? Cbr[CB_REG ? I : 0] : CB[I * KF_WAVE + lane];
  wave_lds_fence();
  double* EB = smem;             // T_e adjoint (2 T_b), N4 x N4 row-major
  double* EC = EB + N4 * N4;     // C_e adjoint
  double* EG = EC + N4 * N4;     // G_1
  double* RQR = EG + N4 * N4;    // m x m
  double* Ms = RQR + mm;         // m x m   symmetrised adjoint of R Q R'
  double* MR = Ms + mm;          // m x r   Ms R
  double* vz = MR + m * r_;      // z
  double* vA = vz + m;           // EC[:, JA]
  double* v1 = vA + m;           // EG[:, JA]
  double* vq = v1 + m;           // RQR z
  double* vp = vq + m;           // P_1 z
  double* vt = vp + m;           // EB[JA, :]
  if (bq < N4) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      EB[(4 * I + g) * N4 + bq] = 2.0 * Tb[I];
      EC[(4 * I + g) * N4 + bq] = cb_fin[I];
      EG[(4 * I + g) * N4 + bq] = G[I];
    }
  }
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  // predicted moments of step 1: compact tape rows, or the tile tape's last slot (S_1[i][j] = P_1, S_1[i][JC] = 2 a_1)
  const double* P1 = TILE ? gtl + (long)A.tape_T * (N4 * N4) : gpP + mm;
  const double* a1 = TILE ? gtl + (long)A.tape_T * (N4 * N4) + JC : gpa + m;
  const int p1s = TILE ? N4 : m, a1s = TILE ? N4 : 1;
  const double a1f = TILE ? 0.5 : 1.0;
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    double s = 0.0;
    for (int l = 0; l < r_; ++l) {
      double rq = 0.0;
      for (int k = 0; k < r_; ++k) rq += gR[i * r_ + k] * gQ[k * r_ + l];
      s += rq * gR[j * r_ + l];
    }
    RQR[e] = s;
  }
  if (lane < m) vz[lane] = gZ[lane];
  wave_lds_fence();
  if (lane < m) {
    double sq = 0.0, sp = 0.0;
    for (int j = 0; j < m; ++j) {
      sq += RQR[lane * m + j] * vz[j];
      sp += 0.5 * (P1[lane * p1s + j] + P1[j * p1s + lane]) * vz[j];
    }
    vq[lane] = sq;
    vp[lane] = sp;
    vA[lane] = EC[lane * N4 + JA];
    v1[lane] = EG[lane * N4 + JA];
    vt[lane] = EB[JA * N4 + lane];
  }
  wave_lds_fence();
  const double CAA = EC[JA * N4 + JA], CAC = EC[JA * N4 + JC], GAA = EG[JA * N4 + JA], GAC = EG[JA * N4 + JC];
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    if (A.g_T) A.g_T[b * (long)mm + e] = (acc ? A.g_T[b * (long)mm + e] : 0.0) + EB[i * N4 + j] + vz[i] * vt[j];
    // symmetrised adjoint of R Q R' (the recursion sees sym(R Q R') only, kalman_filter.py:221)
    Ms[e] = 0.5 * (EC[i * N4 + j] + EC[j * N4 + i]) + vA[i] * vz[j] + vz[i] * vA[j] + CAA * vz[i] * vz[j];
    // adjoint of P_pred[1] (symmetric)
    scratch[A.B * (long)m + b * (long)mm + e] =
        0.5 * (EG[i * N4 + j] + EG[j * N4 + i]) + v1[i] * vz[j] + vz[i] * v1[j] + GAA * vz[i] * vz[j];
  }
  if (lane < m) {
    const int i = lane;
    const double ci = A.c ? A.c[b * (long)m + i] : 0.0;
    double s = 0.0;
    for (int j = 0; j < m; ++j) {
      s += gT[i * m + j] * vt[j];                                                   // T_e row JA = z T
      s += 2.0 * RQR[i * m + j] * vA[j];                                            // C_e column JA = R Q R' z'
      s += 2.0 * 0.5 * (P1[i * p1s + j] + P1[j * p1s + i]) * v1[j];                 // S_1 column JA = P_1 z'
    }
    s += 2.0 * CAA * vq[i] + 4.0 * CAC * ci + 2.0 * GAA * vp[i] + 4.0 * GAC * a1f * a1[i * a1s];
    if (A.g_Z) A.g_Z[b * (long)m + i] = (acc ? A.g_Z[b * (long)m + i] : 0.0) + s;
    if (A.g_c) A.g_c[b * (long)m + i] = (acc ? A.g_c[b * (long)m + i] : 0.0) + 4.0 * EC[i * N4 + JC] + 4.0 * CAC * vz[i];
    scratch[b * (long)m + i] = 4.0 * EG[i * N4 + JC] + 4.0 * GAC * vz[i];   // adjoint of a_pred[1]
  }
  if (lane == 0) {
    if (A.g_H) A.g_H[b] = (acc ? A.g_H[b] : 0.0) + Hb;
  }
  if (lane == JC && A.g_d) A.g_d[b] = (acc ? A.g_d[b] : 0.0) + -2.0 * dsign * dacc;
  wave_lds_fence();
  // R_b = Ms R (Q + Q') ;  Q_b = R' Ms R
  for (int e = lane; e < m * r_; e += KF_WAVE) {
    const int i = e / r_, l = e - i * r_;
    double s = 0.0;
    for (int j = 0; j < m; ++j) s += Ms[i * m + j] * gR[j * r_ + l];
    MR[e] = s;
  }
  wave_lds_fence();
  if (A.g_R)
    for (int e = lane; e < m * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double s = 0.0;
      for (int k = 0; k < r_; ++k) s += MR[i * r_ + k] * (gQ[k * r_ + l] + gQ[l * r_ + k]);
      A.g_R[b * (long)m * r_ + e] = (acc ? A.g_R[b * (long)m * r_ + e] : 0.0) + s;
    }
  if (A.g_Q)
    for (int e = lane; e < r_ * r_; e += KF_WAVE) {
      const int k = e / r_, l = e - k * r_;
      double s = 0.0;
      for (int i = 0; i < m; ++i) s += gR[i * r_ + k] * MR[i * r_ + l];
      A.g_Q[b * (long)r_ * r_ + e] = (acc ? A.g_Q[b * (long)r_ * r_ + e] : 0.0) + s;
    }
}

}  // namespace

bool kf_backward_b4_ok(int m, int p, int r) { return p == 1 && m >= 1 && m + 2 <= 16 && r >= 1 && r <= 64; }
size_t kf_backward_b4_workspace_bytes(long B, int m) { return sizeof(double) * (size_t)B * (size_t)(m + m * m) + 64; }

int kf_launch_backward_b4(kf_handle* h, const KfBwdArgs& a, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
  if (!kf_backward_b4_ok(a.m, a.p, a.r) || (a.flags & KF_FLAG_TV_ANY) || !workspace ||
      workspace_bytes < kf_backward_b4_workspace_bytes(a.B, a.m)) {
    KF_SET_ERR(h, "backward(b4): needs p == 1, m <= 14, static matrices and %zu bytes of workspace (got p=%d m=%d, %zu)",
               kf_backward_b4_workspace_bytes(a.B, a.m), a.p, a.m, workspace_bytes);
    return KF_ERR_UNSUPPORTED;
  }
  if (a.T < 2) {
    if (a.tile_tape) {
      KfBwdArgs one = a;
      one.tile_tape = nullptr; one.pred_a = a.a0; one.pred_P = a.P0; one.tape_T = 0;
      return kf_launch_backward_mfma(h, one, stream);
    }
    return kf_launch_backward_mfma(h, a, stream);
  }
  double* scratch = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  const int nb = (a.m + 2 + 3) / 4;
  const int epi = bb_epilogue_doubles(4 * nb, a.m, a.r);
  const size_t lds = sizeof(double) * (size_t)(epi > BB_LOOP_DOUBLES ? epi : BB_LOOP_DOUBLES);
  if (lds > 64 * 1024) {
    KF_SET_ERR(h, "backward(b4): r=%d needs %zu B of LDS", a.r, lds);
    return KF_ERR_UNSUPPORTED;
  }
#define KF_BB_LAUNCH(NB_)                                                                                              \
  if (a.tile_tape)                                                                                                     \
    hipLaunchKernelGGL((kf_bwd_b4_kernel<NB_, true>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a, scratch);   \
  else                                                                                                                 \
    hipLaunchKernelGGL((kf_bwd_b4_kernel<NB_, false>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a, scratch)
  switch (nb) {
    case 1: KF_BB_LAUNCH(1); break;
    case 2: KF_BB_LAUNCH(2); break;
    case 3: KF_BB_LAUNCH(3); break;
    default: KF_BB_LAUNCH(4); break;
  }
#undef KF_BB_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  // step 0 in the reference's own variables (exact entrywise gradient of P0), added to what the sweep wrote
  KfBwdArgs tail = a;
  tail.T = 1;
  tail.init_ab = scratch;               // [B][m]
  tail.init_Pb = scratch + a.B * a.m;   // [B][m][m]
  tail.accumulate = 1;
  tail.steps = nullptr;  // (the tail is step 0 of every draw, whatever part of the sweep the steady-state adjoint took)
  tail.list = nullptr;
  if (a.tile_tape) {  // the tail's one tape row is the caller's (a0, P0) itself
    tail.tile_tape = nullptr;
    tail.pred_a = a.a0;
    tail.pred_P = a.P0;
    tail.tape_T = 0;
  }
  return kf_launch_backward_mfma(h, tail, stream);
}
