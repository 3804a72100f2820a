// kf_forward_b4q.hip -- forward Kalman filter on v_mfma_f64_4x4x4_4b_f64 for SEVERAL scalar observations per step
// (2 <= p <= 8 processed one at a time: the Univariate filter, or any filter with KF_FLAG_H_DIAGONAL),
// k_states + 1 <= 16, static system matrices, one wavefront per draw.  The p > 1 sibling of kf_forward_b4.hip
// (layout, instruction semantics and the predict step are described in that file's header).
//
// Tile: the second-moment matrix of (alpha, "1") -- indices 0..m-1 the states, JC = N4-1 the carrier of the mean
// (row / column JC holds 2 a).  There is no Z alpha row here (p of them would not fit next to m = 15); instead each
// scalar observation q costs one extra vector-matrix product on the matrix cores:
//     r = z_q S                   : NB MFMAs, left factor = z_q in ONE row of each 4 x 4 block (rows 1..3 zero), so the
//                                   result sits in DPP row 0 and is zero elsewhere; r_JC = 2 z_q a
//     F = H_qq + r . z_q          : one DPP row sum
//     u = r,  u_JC += -2 (y_q -+ d_q)      (= -2 v)
//     tile -= (u / sqrt F)(u / sqrt F)'    : NB MFMAs (A = -u_hat replicated over the block slots, k = 0 only)
// which is kalman_filter.py:468-500's scalar update (P_f = P - M M'/F, a_f = a + M v/F) on covariance and mean at once.
// Likelihood bookkeeping (per-filter log 2 pi counts, skipped / missing observations, non-positive F) follows
// kf_forward_mfma.hip, but is evaluated LB steps at a time, one step per lane, from a ring of (F, u_hat_JC) pairs.
#include <stdlib.h>

#include "kf_b4.h"
#include "kf_b4_split.h"

namespace {

#ifndef KF_B4Q_GRP_WAVES
#define KF_B4Q_GRP_WAVES 4
#endif
constexpr int QLB = 8;   // steps per y block / likelihood ring
constexpr int QPMAX = 8;
// loop-time LDS (doubles): 16 wavefronts per CU need <= 1280 each
constexpr int Q_OFF_UX = 16 * TRS;                      // u by column index (16) | zeros (4)
constexpr int Q_OFF_ZQ = Q_OFF_UX + 20;                 // Z rows, element j of row q at q * 16 + 4 (j & 3) + (j >> 2)
constexpr int Q_OFF_HD = Q_OFF_ZQ + 16 * QPMAX;         // diag(H)
constexpr int Q_OFF_RG = Q_OFF_HD + QPMAX;              // likelihood ring: QLB x p x {F, u_hat_JC}
constexpr int Q_OFF_YB = Q_OFF_RG + 2 * QLB * QPMAX;    // y block: QLB x p x {-2 (y -+ d) or 0, observed ? 1 : 0}
constexpr int Q_OFF_STG = Q_OFF_YB + 2 * QLB * QPMAX;   // covariance tape image: m x m compact + 64 dump slots at 256
constexpr int Q_OFF_PO = Q_OFF_STG + 256 + 64;          // steady-state test: the predicted covariance a checked step started from (m x m compact)
constexpr int Q_LOOP_DOUBLES = Q_OFF_PO + 256;
static_assert(Q_LOOP_DOUBLES <= 1280, "four wavefronts per SIMD need <= 10 KiB of LDS per wavefront");
inline int b4q_setup_doubles(int m, int r) { return 2 * m * r + r * r; }

// TM: which tapes are written -- bit 0 filtered states, 1 predicted states, 2 filtered covariances, 3 predicted
// covariances; TM < 0: decided at run time
__device__ __forceinline__ void st_off(double* base, unsigned byte_off, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// GRP: observations are taken FOUR at a time while at least three are left (see group_update below)
// SS: steady-state phase (3 <= p <= 5 with GRP: one group and at most one single observation) -- see the end of the kernel
template <int NB, int TM, bool GRP, bool SS>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(GRP ? KF_B4Q_GRP_WAVES : 4))) void kf_fwd_b4q_kernel(const KfFwdArgs A) {
  constexpr int N4 = 4 * NB, JC = N4 - 1;
  constexpr bool HALF = NB <= 3;  // T / 2 kept as registers; else the 1/2 is applied at the symmetrisation
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15;
  const int fb = (lane & 3) * 4 + ((lane >> 2) & 3);
  const long b = blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m, p = A.p, r_ = A.r;

  double* TR = smem;
  double* UX = smem + Q_OFF_UX;
  double* ZQ = smem + Q_OFF_ZQ;
  double* HD = smem + Q_OFF_HD;
  double* RG = smem + Q_OFF_RG;
  double* YB = smem + Q_OFF_YB;
  double* STG = smem + Q_OFF_STG;
  double* PO = smem + Q_OFF_PO;

  // ---- per-draw constants ---------------------------------------------------------------------------------------------------
  double Ta[NB][NB], Tha[HALF ? NB : 1][HALF ? NB : 1], Ch[NB], P[NB];
  const double* gP = A.P0 + b * (long)mm;
  {
    const double* gT = A.Tm + b * (long)mm;
    const double* gR = A.R + b * (long)m * r_;
    const double* gQ = A.Q + b * (long)r_ * r_;
    const double* ga = A.a0 + b * (long)m;
    const double* gc = A.c ? A.c + b * (long)m : nullptr;
    double* sR = smem;               // R     m x r
    double* sQ = sR + m * r_;        // Q     r x r
    double* sRQ = sQ + r_ * r_;      // R Q   m x r
    for (int e = lane; e < m * r_; e += KF_WAVE) sR[e] = gR[e];
    for (int e = lane; e < r_ * r_; e += KF_WAVE) sQ[e] = gQ[e];
    wave_lds_fence();
    for (int e = lane; e < m * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      for (int k = 0; k < r_; ++k) v += sR[i * r_ + k] * sQ[k * r_ + l];
      sRQ[e] = v;
    }
    wave_lds_fence();
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g, cc = bq;
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const int ti = 4 * I + (lane & 3), tk = 4 * K + g;
        double tv = 0.0;
        if (ti < m && tk < m) tv = gT[ti * m + tk];
        else if (ti == JC && tk == JC) tv = 1.0;
        Ta[I][K] = tv;
        if (HALF) {
          double th = 0.5 * tv;
          asm volatile("" : "+v"(th));
          Tha[HALF ? I : 0][HALF ? K : 0] = th;
        }
      }
      double pv = 0.0, cv = 0.0;
      if (rr < m && cc < m) {
        pv = (cc >= rr) ? gP[rr * m + cc] : gP[cc * m + rr];  // the upper triangle, mirrored (kf_forward_b4.hip does the same)
        for (int l = 0; l < r_; ++l) cv += sRQ[rr * r_ + l] * sR[cc * r_ + l];
      } else if (rr < m && cc == JC) {
        pv = 2.0 * ga[rr];
        cv = gc ? 2.0 * gc[rr] : 0.0;
      } else if (rr == JC && cc < m) {
        pv = 2.0 * ga[cc];
        cv = gc ? 2.0 * gc[cc] : 0.0;
      }
      P[I] = pv;
      Ch[I] = HALF ? 0.5 * cv : cv;
    }
    wave_lds_fence();  // the scratch is dead
  }
  // C_ext bitwise symmetric: transpose once through the (zeroed) tile and average
  for (int e = lane; e < 16 * TRS; e += KF_WAVE) TR[e] = 0.0;
  wave_lds_fence();
#pragma unroll
  for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = Ch[I];
  wave_lds_fence();
#pragma unroll
  for (int I = 0; I < NB; ++I) Ch[I] = 0.5 * (Ch[I] + TR[bq * TRS + 4 * g + I]);
  wave_lds_fence();
  if (lane < 20) UX[lane] = 0.0;
  {
    const double* gZ = A.Z + b * (long)p * m;
    for (int e = lane; e < 16 * QPMAX; e += KF_WAVE) {
      const int q = e >> 4, k = e & 15;
      ZQ[q * 16 + (k & 3) * 4 + (k >> 2)] = (q < p && k < m) ? gZ[q * m + k] : 0.0;
    }
    if (lane < QPMAX) HD[lane] = lane < p ? A.H[b * (long)p * p + lane * p + lane] : 0.0;
  }
  wave_lds_fence();

  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double joint_const = (A.filter == KF_FILTER_CHOLESKY) ? (double)p : 1.0;

  // z_q as a one-row left factor: lane (g = k, any b, q' = 0) holds z_q[4 K + g] for K = 0..3 (16 bytes x 2); the other lanes read zeros
  const bool zlane = (lane & 3) == 0;
  const double* zq_r = zlane ? ZQ + 4 * g : UX + 16;
  const int zq_s = zlane ? 16 : 0;
  const double* zc_r = ZQ + fb;  // z_q[column of the lane]
  double* ux_w = (g == 0) ? UX + fb : STG + 256 + lane;
  char* rg_w = reinterpret_cast<char*>((lane == JC) ? RG : STG + 256 + 2 * (lane & 31));
  const int rg_s = (lane == JC) ? 16 : 0;
  const double* ux_r = (g == 0) ? UX + 4 * (lane & 3) : UX + 16;
  const double adjm = (lane == JC) ? 1.0 : 0.0;
  const bool vec_on = g == 3 && bq < m;  // row JC of the tile = 2 a'
  double* t_fa = A.filt_a ? A.filt_a + b * (long)n * m : nullptr;
  double* t_pa = A.pred_a ? A.pred_a + b * (long)(n + 1) * m : nullptr;
  double* t_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* t_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;
  const bool w_fa = TM < 0 ? t_fa != nullptr : (TM & 1) != 0, w_pa = TM < 0 ? t_pa != nullptr : (TM & 2) != 0,
             w_fP = TM < 0 ? t_fP != nullptr : (TM & 4) != 0, w_pP = TM < 0 ? t_pP != nullptr : (TM & 8) != 0;
  const unsigned voff = (unsigned)(bq * 8);
  unsigned stg_off[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int rr = 4 * I + g;
    stg_off[I] = (unsigned)(((rr < m && bq < m) ? rr * m + bq : 256 + lane) * 8);
  }
  const int nb128 = mm >> 1;
  auto stage_cov = [&](const double (&Preg)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) *reinterpret_cast<double*>(reinterpret_cast<char*>(STG) + stg_off[I]) = Preg[I];
  };
  const bool st1 = lane < nb128, st1b = (mm & 1) && lane == nb128;
  const bool st2 = lane + KF_WAVE < nb128, st2b = (mm & 1) && lane + KF_WAVE == nb128;
  const bool two_rounds = mm > 2 * KF_WAVE;  // m > 11
  const bool odd_tail = (mm & 1) != 0;
  auto flush_cov = [&](double* dst) {
    wave_lds_fence();
    const double2_t v2 = *reinterpret_cast<const double2_t*>(STG + 2 * lane);
    if (st1) st_stream(reinterpret_cast<double2_t*>(dst) + lane, v2);
    if (odd_tail && st1b) st_stream(dst + mm - 1, v2.x);
    if (two_rounds) {
      const double2_t w2 = *reinterpret_cast<const double2_t*>(STG + 2 * (lane + KF_WAVE));
      if (st2) st_stream(reinterpret_cast<double2_t*>(dst) + lane + KF_WAVE, w2);
      if (odd_tail && st2b) st_stream(dst + mm - 1, w2.x);
    }
    wave_lds_fence();
  };
  bool pred_pending = false;

  // predicted tape slot 0 = (a0, P0 as given)
  if (w_pa) {
    if (vec_on) st_off(t_pa, voff, 0.5 * P[NB - 1]);
    t_pa += m;
  }
  if (w_pP) {
    for (int e = lane; e < mm; e += KF_WAVE) t_pP[e] = gP[e];
    t_pP += mm;
  }

  auto transpose_tile = [&](const double (&X)[NB], double (&Xt)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = X[I];
    wave_lds_fence();
    const double* src = TR + bq * TRS + 4 * g;
    const double2_t t01 = *reinterpret_cast<const double2_t*>(src);
    double2_t t23 = {0.0, 0.0};
    if (NB > 2) t23 = *reinterpret_cast<const double2_t*>(src + 2);
    wave_lds_fence();
    Xt[0] = t01.x;
    if (NB > 1) Xt[NB > 1 ? 1 : 0] = t01.y;
    if (NB > 2) Xt[NB > 2 ? 2 : 0] = t23.x;
    if (NB > 3) Xt[NB > 3 ? 3 : 0] = t23.y;
  };


  // ---- group update: up to four scalar observations at once (GRP; NB >= 3: the exchange buffers alias rows 0..9 of the
  // transpose tile, which the predict step rewrites in full) --------------------------------------------------------------
  //   R  = Z_g S (+ yo in column JC), rows of missing observations zeroed            NB MFMAs, row i in DPP row i
  //   F  = H_g + Z_g R'   (4 x 4; missing observations: identity row / column)        NB MFMAs (R' through LDS)
  //   F  = L L' by scalar elimination, the same arithmetic in every lane; W = L^-1    (the p = 4 sequential filter IS this LDL')
  //   Uh = W R                                                                        1 MFMA
  //   tile -= Uh' Uh                                                                  NB MFMAs (Uh' replicated through LDS)
  // d_i = L_ii^2 is observation i's F and Uh[i][JC] its -2 v / sqrt F, exactly the quantities of the one-at-a-time loop.
  double* RX = TR;         // R rows: observation i at 16 i, column c at position 4 (c & 3) + (c >> 2): 64
  double* UB = TR + 64;    // Uh rows, same layout: 64
  double* FX = TR + 128;   // F row-major: 16
  double* WX = TR + 144;   // W lower triangle, packed: 10
  const int qq = lane & 3;
  double* rw = RX + g * 16 + fb;                  // this lane's element of a 4 x 16 matrix in D layout (UB: + 64)
  const double* rx_r = RX + qq * 16 + 4 * g;      // lane (g = k, q = j): R[j][4 K + k], K = 0..3
  const double* ub_r = UB + g * 16 + 4 * qq;      // lane (g = k, q = i): Uh[k][4 I + i], I = 0..3
  const bool fx_on = bq < 4;                      // one copy of F (the MFMA result is the same in every block slot)
  const double* wx_r = (g <= qq) ? WX + (qq * (qq + 1)) / 2 + g : UX + 16;  // lane (g = k, q = i): W[i][k] (0 above the diagonal)
  const bool isJC = bq == JC, isdg = qq == g;
  // 1 / sqrt(d) for a pivot, 0 for a non-positive one: v_rsq_f64 + ONE third-order correction (e = 1 - d r^2;
  // r <- r + r e (1/2 + 3/8 e): relative error e^3 -- five instructions where two Newton steps take seven)
  auto rs_of = [&](double dv) {
    const double r0 = __builtin_amdgcn_rsq(fmax(dv, 1e-300));
    const double e = fma(-dv * r0, r0, 1.0);
    const double r1 = fma(r0 * e, fma(0.375, e, 0.5), r0);
    return (dv > 0.0) ? r1 : 0.0;
  };
  auto group_update = [&](int q0, int e0, double* d_keep) {
    const int ob = q0 + g;
    const bool valid = ob < p;
    const double2_t yo = *reinterpret_cast<const double2_t*>(valid ? YB + 2 * (e0 + ob) : UX + 16);
    const double* za = ZQ + (q0 + qq) * 16 + 4 * g;
    const double2_t z01 = *reinterpret_cast<const double2_t*>(za);
    const double2_t z23 = *reinterpret_cast<const double2_t*>(za + 2);
    const double hq = HD[ob & (QPMAX - 1)];
    const double zA[4] = {z01.x, z01.y, z23.x, z23.y};
    double R = mfma4(zA[0], P[0], 0.0);
    {
      double R2 = mfma4(zA[1], P[NB > 1 ? 1 : 0], 0.0);
      if (NB > 2) R = mfma4(zA[2], P[NB > 2 ? 2 : 0], R);
      if (NB > 3) R2 = mfma4(zA[3], P[NB > 3 ? 3 : 0], R2);
      R += R2;
    }
    const double flag = yo.y;
    const double Rm = (R + (isJC ? yo.x : 0.0)) * flag;
    *rw = Rm;
    wave_lds_fence();
    const double2_t t01 = *reinterpret_cast<const double2_t*>(rx_r);
    const double2_t t23 = *reinterpret_cast<const double2_t*>(rx_r + 2);
    double G = mfma4(zA[0], t01.x, 0.0);
    {
      double G2 = mfma4(zA[1], t01.y, 0.0);
      if (NB > 2) G = mfma4(zA[2], t23.x, G);
      if (NB > 3) G2 = mfma4(zA[3], t23.y, G2);
      G += G2;
    }
    // F = flag (G + [i == j] H) + (1 - flag) [i == j]
    {
      const double Fv = isdg ? fma(flag, (G + hq) - 1.0, 1.0) : flag * G;
      if (fx_on) FX[g * 4 + qq] = Fv;
    }
    wave_lds_fence();
    const double2_t f0 = *reinterpret_cast<const double2_t*>(FX);
    const double2_t f1 = *reinterpret_cast<const double2_t*>(FX + 4);
    const double2_t f2a = *reinterpret_cast<const double2_t*>(FX + 8);
    const double2_t f2b = *reinterpret_cast<const double2_t*>(FX + 10);
    const double2_t f3a = *reinterpret_cast<const double2_t*>(FX + 12);
    const double2_t f3b = *reinterpret_cast<const double2_t*>(FX + 14);
    // L L' = F, skipping a non-positive pivot (rs = 0: that observation updates nothing, as in the one-at-a-time loop)
    const double d0 = f0.x, rs0 = rs_of(d0);
    const double l10 = f1.x * rs0, l20 = f2a.x * rs0, l30 = f3a.x * rs0;
    const double d1 = fma(-l10, l10, f1.y), rs1 = rs_of(d1);
    const double l21 = fma(-l20, l10, f2a.y) * rs1, l31 = fma(-l30, l10, f3a.y) * rs1;
    const double d2 = fma(-l21, l21, fma(-l20, l20, f2b.x)), rs2 = rs_of(d2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, f3b.x)) * rs2;
    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, f3b.y))), rs3 = rs_of(d3);
    // W = L^-1 (lower triangle)
    const double w10 = -(l10 * rs0) * rs1;
    const double w21 = -(l21 * rs1) * rs2;
    const double w32 = -(l32 * rs2) * rs3;
    const double w20 = -fma(l21, w10, l20 * rs0) * rs2;
    const double w31 = -fma(l32, w21, l31 * rs1) * rs3;
    const double w30 = -fma(l32, w20, fma(l31, w10, l30 * rs0)) * rs3;
    if (lane == 0) {
      double2_t a0 = {rs0, w10}, a1 = {rs1, w20}, a2 = {w21, rs2}, a3 = {w30, w31}, a4 = {w32, rs3};
      *reinterpret_cast<double2_t*>(WX) = a0;
      *reinterpret_cast<double2_t*>(WX + 2) = a1;
      *reinterpret_cast<double2_t*>(WX + 4) = a2;
      *reinterpret_cast<double2_t*>(WX + 6) = a3;
      *reinterpret_cast<double2_t*>(WX + 8) = a4;
    }
    wave_lds_fence();
    const double wv = *wx_r;
    const double Uh = mfma4(wv, Rm, 0.0);
    rw[64] = Uh;
    wave_lds_fence();
    const double2_t u01 = *reinterpret_cast<const double2_t*>(ub_r);
    const double2_t u23 = *reinterpret_cast<const double2_t*>(ub_r + 2);
    {
      const double dsel = (g == 0) ? d0 : (g == 1) ? d1 : (g == 2) ? d2 : d3;
      if (d_keep) *d_keep = dsel;
      double2_t fq = {dsel, Uh};
      if (valid && isJC) *reinterpret_cast<double2_t*>(RG + 2 * (e0 + ob)) = fq;
    }
    const double uT[4] = {u01.x, u01.y, u23.x, u23.y};
#pragma unroll
    for (int I = 0; I < NB; ++I) P[I] = mfma4_nega(uT[I], Uh, P[I]);
  };

  // one observation at a time (all of them when !GRP; the one or two left over behind the groups otherwise)
  auto single_update = [&](int q, int e0, const double* zq, const double* zc, char* rgw, double* f_keep, double* rs_keep) {
      const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (e0 + q));  // {-2 (y -+ d), observed}
      const double2_t z01 = *reinterpret_cast<const double2_t*>(zq);
      double2_t z23 = {0.0, 0.0};
      if (NB > 2) z23 = *reinterpret_cast<const double2_t*>(zq + 2);
      const double zcol = *zc;
      const double Hq = HD[q];
      const double zA[4] = {z01.x, z01.y, z23.x, z23.y};
      // two accumulation chains: a dependent 4x4x4 MFMA waits for its predecessor's four passes
      double r = mfma4(zA[0], P[0], 0.0);
      if (NB > 1) {
        double r2 = mfma4(zA[1], P[NB > 1 ? 1 : 0], 0.0);
        if (NB > 2) r = mfma4(zA[2], P[NB > 2 ? 2 : 0], r);
        if (NB > 3) r2 = mfma4(zA[3], P[NB > 3 ? 3 : 0], r2);
        r += r2;
      }
      double u;
      double2_t ur01, ur23 = {0.0, 0.0};
      if constexpr (GRP && NB >= 3) {
        // (an exchange slot of the group path's shape at TR + 160: r is zero outside DPP row 0, so rows 1..3 receive zeros)
        u = r + ((lane == JC) ? yo.x : 0.0);
        rw[160] = u;
        wave_lds_fence();
        ur01 = *reinterpret_cast<const double2_t*>(ub_r + 96);
        ur23 = *reinterpret_cast<const double2_t*>(ub_r + 98);
      } else {
        u = fma(adjm, yo.x, r);  // u_JC = 2 z a - 2 (y -+ d) = -2 v
        *ux_w = u;  // (rows 1..3 write zeros into a dump slot: no exec-mask switch)
        wave_lds_fence();
        ur01 = *reinterpret_cast<const double2_t*>(ux_r);
        if (NB > 2) ur23 = *reinterpret_cast<const double2_t*>(ux_r + 2);
      }
      wave_lds_fence();
      const double urep[4] = {ur01.x, ur01.y, ur23.x, ur23.y};
      const double F = Hq + readlane_f64(row_allsum(r * zcol), 0);
      // missing -> no update; F == 0 (Univariate: skipped, :468-471) and F < 0 / NaN (flagged when the ring is evaluated): no update
      // either.  (A select, not a branch: F is wave-uniform, but a taken branch costs more than the nine instructions.)
      const double rs = fast_rsqrt(fmax(F, 1e-300)) * ((F > 0.0) ? yo.y : 0.0);
      const double uh = u * rs;
      if (f_keep) {
        *f_keep = F;
        *rs_keep = rs;
      }
      {
        double2_t fq = {F, uh};
        if constexpr (GRP && NB >= 3) {
          if (lane == JC) *reinterpret_cast<double2_t*>(RG + 2 * (e0 + q)) = fq;
        } else {
          *reinterpret_cast<double2_t*>(rgw) = fq;  // lane JC: the ring; the others: their dump slot
        }
      }
#pragma unroll
      for (int I = 0; I < NB; ++I) P[I] = mfma4_nega(urep[I] * rs, uh, P[I]);
  };

  double logp = 0.0;
  int info = 0;
  // ll of the steps [tb, tb + cnt) held in the ring, one step per lane; the observed flags of these steps are still in
  // the y block.  Mirrors the per-observation bookkeeping of kf_forward_mfma.hip (kalman_filter.py:468-500 / :255-351).
  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
    const bool act = lane < cnt;
    const int ls = act ? lane : 0;
    double mant = 1.0, qs = 0.0;
    int ex = 0, cnt_ll = 0, nob = 0, first_bad = 0;
    for (int q = 0; q < p; ++q) {
      const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * (ls * p + q));
      const bool obs = YB[2 * (ls * p + q) + 1] != 0.0;
      const double Fv = fq.x;
      const bool degenerate = !obs || (univariate && Fv == 0.0);  // :468-471
      const bool notpd = !degenerate && !(Fv > 0.0);
      if (notpd && first_bad == 0) first_bad = q + 1;
      nob += obs ? 1 : 0;
      if (!degenerate) {
        int e2;
        mant *= frexp(Fv, &e2);
        ex += e2;
        const double quad = 0.25 * fq.y * fq.y;  // v^2 / F
        qs += quad;
        cnt_ll += (Fv == 1.0 && quad == 0.0) ? 0 : 1;  // :497-500 counts non-zero scalar terms
      }
    }
    const unsigned long long badm = __ballot(act && first_bad != 0);
    if (badm != 0ull && info == 0) {
      const int fl = __ffsll((long long)badm) - 1;
      info = ((tb + fl + 1) << 8) | __shfl(first_bad, fl);
    }
    double nconst, xv;
    if (univariate) {
      nconst = exact_const ? (double)nob : (double)cnt_ll;
      xv = mant;
    } else {
      nconst = exact_const ? (double)nob : joint_const;
      xv = (nob == 0) ? -1.0 : mant;  // all-missing step: ll = 0
    }
    double llv = (xv == -1.0 || !act) ? 0.0 : -0.5 * (log(xv) + fma(0.69314718055994530942, (double)ex, qs) + nconst * KF_LOG_2PI);
    if (act && first_bad != 0) llv = __builtin_nan("");
    if (act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
    __syncthreads();
  };

  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L);
  const double* gd = A.d ? A.d + b * (long)p : nullptr;

  auto load_y_block = [&](int t) {
    const int cnt = ((n - t) < QLB ? (n - t) : QLB) * p;
    for (int e = lane; e < cnt; e += KF_WAVE) {
      const double yv = ybase[(long)t * p + e];
      const bool ob = !(yv != yv);
      const double dq = gd ? gd[e % p] : 0.0;
      double2_t ye = {ob ? -2.0 * (yv + dsign * dq) : 0.0, ob ? 1.0 : 0.0};
      *reinterpret_cast<double2_t*>(YB + 2 * e) = ye;
    }
    wave_lds_fence();
  };

  // ---- steady state (SS; see kf_forward_b4.hip): every 8 steps, behind the last missing observation, the predicted covariance is
  // compared with the one the step started from (parked in LDS); once it has stopped moving the loop at the end takes over
  constexpr bool SSK = SS && GRP && NB >= 3;
  const bool steady_en = SSK && p >= 3 && p <= 5 && n >= 64;
  int t_free = 0;
  if (steady_en) {
    int tn = -1;
    for (int e = lane; e < n * p; e += KF_WAVE) {
      const double yv = ybase[e];
      if (yv != yv) tn = e;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) tn = max(tn, __shfl_xor(tn, off));
    t_free = tn < 0 ? 0 : tn / p + 1;
  }

  int t = 0;
  int npass = 0;  // consecutive convergence checks passed
  for (; t < n; ++t) {
    if ((t & (QLB - 1)) == 0) load_y_block(t);
    const bool chk = steady_en && (t & 7) == 7 && t >= t_free && t + 16 < n;  // wave-uniform
    if (chk) {
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (stg_off[I] < 256u * 8u) *reinterpret_cast<double*>(reinterpret_cast<char*>(PO) + stg_off[I]) = P[I];
    }
    if (pred_pending) {  // predicted covariance of the previous step: tile image -> HBM
      flush_cov(t_pP);
      t_pP += mm;
      pred_pending = false;
    }
    // ---- update: the p scalar observations in order ---------------------------------------------------------------------
    const int e0 = (t & (QLB - 1)) * p;
    int qs0 = 0;
    if (GRP && NB >= 3) {
#pragma unroll 1
      for (; p - qs0 >= 3; qs0 += 4) group_update(qs0, e0, nullptr);
      if (qs0 > p) qs0 = p;
    }
    const double* zq = zq_r + qs0 * zq_s;
    const double* zc = zc_r + qs0 * 16;
    char* rgw = rg_w + (e0 + qs0) * rg_s;
#pragma unroll 1
    for (int q = qs0; q < p; ++q) {
      single_update(q, e0, zq, zc, rgw, nullptr, nullptr);
      zq += zq_s;
      zc += 16;
      rgw += rg_s;
    }
    // ---- filtered tapes -------------------------------------------------------------------------------------------------------
    if (w_fa) {
      if (vec_on) st_off(t_fa, voff, 0.5 * P[NB - 1]);
      t_fa += m;
    }
    if (w_fP) stage_cov(P);  // flushed below, behind the MFMAs
    // ---- predict: tile <- sym(T_ext tile T_ext' + C_ext) -------------------------------------------------------------------
    double U[NB], Y[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][0], P[0], 0.0);
#pragma unroll
    for (int K = 1; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][K], P[K], U[I]);
    if (w_fP) {
      flush_cov(t_fP);
      t_fP += mm;
    }
    transpose_tile(U, U);
#pragma unroll
    for (int I = 0; I < NB; ++I) Y[I] = Ch[I];
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I) Y[I] = mfma4(HALF ? Tha[HALF ? I : 0][HALF ? K : 0] : Ta[I][K], U[K], Y[I]);
    {
      double Yt[NB];
      transpose_tile(Y, Yt);
#pragma unroll
      for (int I = 0; I < NB; ++I) P[I] = HALF ? Y[I] + Yt[I] : 0.5 * (Y[I] + Yt[I]);
    }
    bool conv = false;
    if (chk) {
      // converged (as in kf_forward_b4.hip): on two consecutive checks the one-step change is within 1e-15 of the largest entry AND
      // every entry's change within 1e-12 of its own scale sqrt(P_ii P_jj) -- a diffuse block cannot hide a small one that still moves
      unsigned pm = 0;
      bool bad = false;
      double* DG = PO + 240;  // (m * m <= 225: the last 16 doubles of the parked image are free)
      wave_lds_fence();
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (stg_off[I] < 256u * 8u && 4 * I + g == bq) DG[bq] = P[I];
      wave_lds_fence();
      const double dcol = DG[bq < m ? bq : 0];
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (stg_off[I] < 256u * 8u) pm = max(pm, abs_hi(P[I]));
      const double lim = 1e-15 * wave_absmax_hi(pm);
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        const bool on = stg_off[I] < 256u * 8u;
        const double old = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(PO) + (on ? stg_off[I] : 0u));
        const double drow = DG[on ? 4 * I + g : 0];
        const double df = P[I] - old;
        bad = bad || (on && !(fabs(df) <= lim && df * df <= A.ss_tol2 * fabs(dcol * drow)));
      }
      wave_lds_fence();
      const bool ok = __ballot(bad) == 0ull;
      npass = ok ? npass + 1 : 0;
      conv = npass >= 2;
    }
    if (w_pa) {
      if (vec_on) st_off(t_pa, voff, 0.5 * P[NB - 1]);
      t_pa += m;
    }
    if (w_pP) {
      stage_cov(P);
      pred_pending = true;
    }
    if ((t & (QLB - 1)) == QLB - 1) flush_ring(t - (QLB - 1), QLB);
    if (conv) {
      ++t;
      break;
    }
  }
  if (A.hs_ts && lane == 0) A.hs_ts[b] = t;  // (where the steady-state phase starts; n if it never does)
  if constexpr (SSK) if (t < n && A.hs_tile) {
    // split steady-state phase (kf_forward_b4q_steady.hip): leave the converged tile, the log-likelihood so far and the flag; the
    // constant covariance tapes, the mean recursion and the rest of the log-likelihood come from the kernels launched behind this one
    if (pred_pending) flush_cov(t_pP);
    if (t & (QLB - 1)) flush_ring(t & ~(QLB - 1), t & (QLB - 1));
    double* ht = A.hs_tile + b * (long)B4_HS_STRIDE + lane;
#pragma unroll
    for (int I = 0; I < NB; ++I) ht[I * KF_WAVE] = P[I];
    if (lane == 0) {
      A.hs_logp[b] = logp;
      A.hs_info[b] = info;
    }
    return;
  }
  if constexpr (SSK) if (t < n) {
    // ---- steady-state steps t .. n-1: P is the converged predicted tile of step t.  The first of them runs the ordinary
    // update once more and leaves its constants where the update put them -- W and the rows of Uh (the group), u (the single
    // observation) in their LDS exchange slots, which nothing overwrites from here on (no more transposes) -- then only a MEAN
    // tile Sm (column JC = 2 a, zero elsewhere) goes through the same MFMAs: R = Z_g Sm, Uh = W R, Sm -= Uh_const' Uh,
    // and Sm <- T_e Sm + (column JC of C_e) for the predict.
    double Sm[NB], Cc[NB];
    double d_c = 1.0, f_c = 1.0, rs_c = 0.0;
    constexpr int qs1 = 4;  // observations 0..3 are the group; at most one more behind it (3 <= p <= 5)
    // The covariance tapes of ALL remaining steps are two constants: written here in one go, as whole aligned KiB straight
    // from a compact image in LDS (element i of the tape = image[i mod m^2]) -- 16 bytes per lane, every lane busy.  (Per-step
    // image flushes, 14 partly filled vector-memory instructions per wavefront and step, held the steady C4 steps at the
    // address-processing rate of the CU: 16.3 ms with full tapes against 6.3 ms without.)
    auto bulk_fill = [&](double* tp, int cnt) {
      const int a0 = (int)((reinterpret_cast<unsigned long long>(tp) >> 3) & 127ull);
      const int Aend = a0 + cnt, r128 = 128 % mm;
      int idx = (2 * lane - a0) % mm;
      if (idx < 0) idx += mm;
      for (int Aw = 0; Aw < Aend; Aw += 128) {
        const int i = Aw + 2 * lane - a0;
        const int idx1 = (idx + 1 == mm) ? 0 : idx + 1;
        const double2_t v = {STG[idx], STG[idx1]};
        double* dst = tp + i;
        if (Aw >= a0 && Aw + 128 <= Aend) {  // wave-uniform: a whole window
          *reinterpret_cast<double2_t*>(dst) = v;
        } else {
          const bool k0 = i >= 0 && i < cnt, k1 = i + 1 >= 0 && i + 1 < cnt;
          if (k0 && k1) *reinterpret_cast<double2_t*>(dst) = v;
          else if (k0) dst[0] = v.x;
          else if (k1) dst[1] = v.y;
        }
        idx += r128;
        if (idx >= mm) idx -= mm;
      }
    };
    // (tried: odd draws fill behind the mean loop instead of before it, so that the wavefronts of a SIMD stream and compute at the
    // same time -- 14.9 ms against 13.0 at 16384 draws, 2.29 against 2.39 at 2048; not kept)
    constexpr bool fill_first = true;
    if (w_pP && pred_pending) {  // the slot of step t is still pending
      flush_cov(t_pP);
      t_pP += mm;
      pred_pending = false;
    }
    double Pp[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) Pp[I] = P[I];
    const int n_fill = (n - t) * mm;
    const bool do_fill = fill_first && !A.skip_steady_cov;  // (the steady-state adjoint reads slot t of the predicted tape only)
    if (w_pP && do_fill) {  // predicted covariances of steps t+1 .. n: the converged tile
      stage_cov(Pp);
      wave_lds_fence();
      bulk_fill(t_pP, n_fill);
      wave_lds_fence();
    }
    // step t: the ordinary update once more; it leaves W / Uh / u in their exchange slots
    if ((t & (QLB - 1)) == 0) load_y_block(t);
    {
      const int e0 = (t & (QLB - 1)) * p;
      group_update(0, e0, &d_c);
      if (qs1 < p) single_update(qs1, e0, zq_r + qs1 * zq_s, zc_r + qs1 * 16, nullptr, &f_c, &rs_c);
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      Sm[I] = isJC ? P[I] : 0.0;
      Cc[I] = isJC ? (HALF ? 2.0 * Ch[I] : Ch[I]) : 0.0;
    }
    if (isJC) {  // the constants' own mean entries (column JC of Uh / u) take no part in the mean tile's updates
      rw[64] = 0.0;
      rw[160] = 0.0;
    }
    wave_lds_fence();
    if (w_fP && do_fill) {  // filtered covariances of steps t .. n-1
      stage_cov(P);
      wave_lds_fence();
      bulk_fill(t_fP, n_fill);
      wave_lds_fence();
    }
    // state tapes: column JC of the mean tile (lane (g, JC) of register I: 2 a[4 I + g]) through two LDS rings, out as aligned windows
    TapeStream<128, 64> s_fa, s_pa;
    s_fa.init(t_fa, PO);
    s_pa.init(t_pa, PO + 128);
    unsigned o_fa[NB], o_pa[NB];
    bool a_on[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      a_on[I] = isJC && 4 * I + g < m;
      o_fa[I] = s_fa.slot(a_on[I] ? 4 * I + g : 0);
      o_pa[I] = s_pa.slot(a_on[I] ? 4 * I + g : 0);
    }
    auto put_state = [&](TapeStream<128, 64>& st, unsigned (&off)[NB]) {
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        if (a_on[I]) *reinterpret_cast<double*>(reinterpret_cast<char*>(st.ring) + off[I]) = Sm[I];
        off[I] = TapeStream<128, 64>::next(off[I], m);
      }
      st.appended(m, lane, 0.5);
    };
    bool first = true;
    for (; t < n; ++t) {
      const int e0 = (t & (QLB - 1)) * p;
      if (!first) {
        if ((t & (QLB - 1)) == 0) load_y_block(t);
        {
          const int ob = g;
          const bool valid = ob < p;
          const double2_t yo = *reinterpret_cast<const double2_t*>(valid ? YB + 2 * (e0 + ob) : UX + 16);
          const double* za = ZQ + qq * 16 + 4 * g;
          const double2_t z01 = *reinterpret_cast<const double2_t*>(za);
          const double2_t z23 = *reinterpret_cast<const double2_t*>(za + 2);
          const double wv = *wx_r;
          const double2_t u01 = *reinterpret_cast<const double2_t*>(ub_r);
          const double2_t u23 = *reinterpret_cast<const double2_t*>(ub_r + 2);
          double R = mfma4(z01.x, Sm[0], 0.0);
          {
            double R2 = mfma4(z01.y, Sm[1], 0.0);
            R = mfma4(z23.x, Sm[2], R);
            if (NB > 3) R2 = mfma4(z23.y, Sm[NB > 3 ? 3 : 0], R2);
            R += R2;
          }
          const double Rm = R + (isJC ? yo.x : 0.0);
          const double Uh = mfma4(wv, Rm, 0.0);
          {
            double2_t fq = {d_c, Uh};
            if (valid && isJC) *reinterpret_cast<double2_t*>(RG + 2 * (e0 + ob)) = fq;
          }
          const double uT[4] = {u01.x, u01.y, u23.x, u23.y};
#pragma unroll
          for (int I = 0; I < NB; ++I) Sm[I] = mfma4_nega(uT[I], Uh, Sm[I]);
        }
        if (qs1 < p) {
          const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (e0 + qs1));
          const double* zq = zq_r + qs1 * zq_s;
          const double2_t z01 = *reinterpret_cast<const double2_t*>(zq);
          const double2_t z23 = *reinterpret_cast<const double2_t*>(zq + 2);
          const double2_t ur01 = *reinterpret_cast<const double2_t*>(ub_r + 96);
          const double2_t ur23 = *reinterpret_cast<const double2_t*>(ub_r + 98);
          double r = mfma4(z01.x, Sm[0], 0.0);
          {
            double r2 = mfma4(z01.y, Sm[1], 0.0);
            r = mfma4(z23.x, Sm[2], r);
            if (NB > 3) r2 = mfma4(z23.y, Sm[NB > 3 ? 3 : 0], r2);
            r += r2;
          }
          const double uh = (r + ((lane == JC) ? yo.x : 0.0)) * rs_c;
          {
            double2_t fq = {f_c, uh};
            if (lane == JC) *reinterpret_cast<double2_t*>(RG + 2 * (e0 + qs1)) = fq;
          }
          const double urep[4] = {ur01.x, ur01.y, ur23.x, ur23.y};
#pragma unroll
          for (int I = 0; I < NB; ++I) Sm[I] = mfma4_nega(urep[I] * rs_c, uh, Sm[I]);
        }
      }
      first = false;
      if (w_fa) put_state(s_fa, o_fa);
      // predict: column JC of T_e Sm + C_e
      double Sn[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) Sn[I] = mfma4(Ta[I][0], Sm[0], Cc[I]);
#pragma unroll
      for (int K = 1; K < NB; ++K)
#pragma unroll
        for (int I = 0; I < NB; ++I) Sn[I] = mfma4(Ta[I][K], Sm[K], Sn[I]);
#pragma unroll
      for (int I = 0; I < NB; ++I) Sm[I] = Sn[I];
      if (w_pa) put_state(s_pa, o_pa);
      if ((t & (QLB - 1)) == QLB - 1) flush_ring(t - (QLB - 1), QLB);
    }
    s_fa.finish(lane, 0.5);
    s_pa.finish(lane, 0.5);
    if (!fill_first) {
      if (w_pP) {
        stage_cov(Pp);
        wave_lds_fence();
        bulk_fill(t_pP, n_fill);
        wave_lds_fence();
      }
      if (w_fP) {
        stage_cov(P);
        wave_lds_fence();
        bulk_fill(t_fP, n_fill);
      }
    }
  }
  if (pred_pending) flush_cov(t_pP);
  if (n & (QLB - 1)) flush_ring(n & ~(QLB - 1), n & (QLB - 1));
  if (lane == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

bool kf_forward_b4q_ok(int m, int p, int r) {
  return p >= 2 && p <= QPMAX && m >= 1 && m + 1 <= 16 && r >= 1 && r <= 64;
}

size_t kf_b4q_split_workspace_bytes(long B);
void kf_b4q_split_carve(void* base, long B, KfFwdArgs* a, double** sp);
int kf_b4q_split_steady(kf_handle* h, const KfFwdArgs& a, double* sp, hipStream_t s_prep, hipStream_t s_fill, hipStream_t s_scan,
                        hipEvent_t ev_rec, hipEvent_t ev_prep, hipEvent_t ev_fill, hipEvent_t ev_scan);
int kf_b4q_sp_stride();

// the arguments of the draws [lo, lo + cnt): every per-draw pointer moved, B = cnt
static KfFwdArgs b4q_draw_range(const KfFwdArgs& a, long lo, long cnt) {
  KfFwdArgs s = a;
  const long n = a.T, m = a.m, p = a.p, r = a.r, mm = m * m;
  s.B = cnt;
  if (a.flags & KF_FLAG_Y_PER_DRAW) s.y = a.y + lo * n * p;
  s.a0 = a.a0 + lo * m; s.P0 = a.P0 + lo * mm; s.Tm = a.Tm + lo * mm; s.Z = a.Z + lo * p * m; s.R = a.R + lo * m * r;
  s.H = a.H + lo * p * p; s.Q = a.Q + lo * r * r;
  if (a.c) s.c = a.c + lo * m;
  if (a.d) s.d = a.d + lo * p;
  s.logp = a.logp + lo;
  if (a.ll_obs) s.ll_obs = a.ll_obs + lo * n;
  if (a.filt_a) s.filt_a = a.filt_a + lo * n * m;
  if (a.pred_a) s.pred_a = a.pred_a + lo * (n + 1) * m;
  if (a.filt_P) s.filt_P = a.filt_P + lo * n * mm;
  if (a.pred_P) s.pred_P = a.pred_P + lo * (n + 1) * mm;
  if (a.info) s.info = a.info + lo;
  s.hs_tile = a.hs_tile + lo * B4_HS_STRIDE; s.hs_logp = a.hs_logp + lo; s.hs_info = a.hs_info + lo; s.hs_ts = a.hs_ts + lo;
  return s;
}

int kf_launch_forward_b4q(kf_handle* h, const KfFwdArgs& a_in, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  KfFwdArgs a = a_in;
  if (!kf_forward_b4q_ok(a.m, a.p, a.r) || (a.flags & KF_FLAG_TV_ANY)) {
    KF_SET_ERR(h, "forward(b4q): needs 2 <= p <= %d scalar observations, m <= 15, static matrices (got p=%d m=%d r=%d)",
               QPMAX, a.p, a.m, a.r);
    return KF_ERR_UNSUPPORTED;
  }
  const int lds_setup = b4q_setup_doubles(a.m, a.r);
  const size_t lds = sizeof(double) * (size_t)(Q_LOOP_DOUBLES > lds_setup ? Q_LOOP_DOUBLES : lds_setup);
  const int nb = (a.m + 1 + 3) / 4;
  const int mask = (a.filt_a ? 1 : 0) | (a.pred_a ? 2 : 0) | (a.filt_P ? 4 : 0) | (a.pred_P ? 8 : 0);
  // KF_B4Q_GROUP=0 in the environment: every observation one at a time (A/B measurements)
  const char* ge = kf_env_switch("KF_B4Q_GROUP");
  const bool grp = nb >= 3 && a.p >= 3 && !(ge && *ge && atoi(ge) == 0);
  // steady-state phase: its own instantiation (KF_FLAG_EXACT_RECURSION / KF_STEADY_STATE=0 run the kernel without it)
  const char* sse = kf_env_switch("KF_STEADY_STATE");
  const bool ss = grp && a.p <= 5 && a.T >= 64 && !(a.flags & KF_FLAG_EXACT_RECURSION) && !(sse && *sse && atoi(sse) == 0);
  // split steady-state phase (kf_forward_b4q_steady.hip) when the caller passed the workspace; KF_B4Q_SPLIT=0: the phase inside
  // the recursion kernel (round 2)
  double* sp = nullptr;
  bool split = false;
  {
    const char* e = kf_env_switch("KF_B4Q_SPLIT");
    split = ss && nb >= 3 && a.m <= 15 && workspace && workspace_bytes >= kf_b4q_split_workspace_bytes(a.B) && !(e && e[0] == '0');
    if (split) {
      const int rc0 = kf_handle_aux(h);
      if (rc0) return rc0;
      kf_b4q_split_carve(workspace, a.B, &a, &sp);
    }
  }
#define KF_B4Q_LAUNCH2(NB_, TM_)                                                                                              \
  do {                                                                                                                        \
    if (ss && NB_ >= 3)                                                                                                       \
      hipLaunchKernelGGL((kf_fwd_b4q_kernel<NB_, TM_, (NB_ >= 3), (NB_ >= 3)>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a); \
    else if (grp && NB_ >= 3)                                                                                                 \
      hipLaunchKernelGGL((kf_fwd_b4q_kernel<NB_, TM_, (NB_ >= 3), false>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a); \
    else                                                                                                                      \
      hipLaunchKernelGGL((kf_fwd_b4q_kernel<NB_, TM_, false, false>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a);     \
  } while (0)
#define KF_B4Q_LAUNCH(NB_)                           \
  do {                                               \
    if (mask == 15) KF_B4Q_LAUNCH2(NB_, 15);         \
    else if (mask == 0) KF_B4Q_LAUNCH2(NB_, 0);      \
    else if (mask == 10) KF_B4Q_LAUNCH2(NB_, 10);    \
    else KF_B4Q_LAUNCH2(NB_, -1);                    \
  } while (0)
  // split phase: draws in chunks of 4096 (one round of the chip's wavefront slots), so that the constants, the fill and the scan of a
  // chunk run beside the recursion of the next one
  long chunk = split ? 4096 : a.B;
  {
    const char* e = kf_env_experiment("KF_B4Q_CHUNK");  // experiments
    if (split && e && *e && atol(e) >= 256) chunk = atol(e);
  }
  const char* pse = kf_env_experiment("KF_B4Q_PREP_STREAM");
  hipStream_t s_prep = (pse && pse[0] == '1') ? h->aux_stream : h->aux_stream3;
  const KfFwdArgs a_all = a;
  KfAuxJoinGuard join_guard(split ? h : nullptr, stream);
  int ci = 0;
  for (long lo = 0; lo < a_all.B; lo += chunk, ++ci) {
    const long cnt = (a_all.B - lo) < chunk ? (a_all.B - lo) : chunk;
    if (split) a = b4q_draw_range(a_all, lo, cnt);
    switch (nb) {
      case 1: KF_B4Q_LAUNCH(1); break;
      case 2: KF_B4Q_LAUNCH(2); break;
      case 3: KF_B4Q_LAUNCH(3); break;
      default: KF_B4Q_LAUNCH(4); break;
    }
    KF_HIP_CHECK(h, hipGetLastError());
    if (split) {
      hipEvent_t* ev = h->aux_ev + 4 * (ci % 8);
      KF_HIP_CHECK(h, hipEventRecord(ev[0], stream));
      const int rc = kf_b4q_split_steady(h, a, sp + lo * kf_b4q_sp_stride(), s_prep, h->aux_stream, h->aux_stream2, ev[0], ev[1], ev[2], ev[3]);
      if (rc) return rc;
      if (lo + cnt >= a_all.B) {  // the last chunk: the caller's stream continues behind both steady streams
        KF_HIP_CHECK(h, hipStreamWaitEvent(stream, ev[2], 0));
        KF_HIP_CHECK(h, hipStreamWaitEvent(stream, ev[3], 0));
      }
    }
  }
#undef KF_B4Q_LAUNCH2
#undef KF_B4Q_LAUNCH
  join_guard.dismiss();
  return KF_OK;
}
