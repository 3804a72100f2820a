// kf_forward_mfma.hip -- forward Kalman filter with the covariance in MFMA accumulator layout
// (one wavefront per draw, k_states <= 16, static system matrices, p <= 8 observations processed
// one at a time).
//
// Why: the row-owner kernels (kf_forward_row*.hip) feed their two m x m x m products per step
// from LDS broadcast reads -- ~70 KB of LDS return traffic per wavefront-step at m = 10, and the
// 128 B/clk/CU LDS pipe, not latency, is what bounds them at the north-star shape
// (4096 draws x T = 1000 x m = 10: 8 waves/CU x 70 KB / 128 B/clk ~ 4400 clk per step).
// Here the 16 x 16 (zero-padded) covariance lives in the register layout of
// v_mfma_f64_16x16x4_f64's accumulator: lane l = (g = l >> 4, c = l & 15), register r holds
// P[g + 4r][c].  Because P is kept BITWISE symmetric, the same four registers are also a valid
// A operand (A[i = c][k = 4s + g] = P[4s + g][c]), so
//     W2 = P_f T'      : 4 MFMAs, A = P_f registers,        B = T (static registers)
//     S  = T W2 + RQR' : 4 MFMAs, A = T/2 (static),         B = W2 accumulators, C = RQR'/2
// need no operand traffic at all.  P_hat = (S + S')/2 (:221) takes one 2 KB LDS transpose.
//
// Update (reference kalman_filter.py:255-351 for p == 1; sequential processing :444-505, which
// with diagonal H is algebraically the joint update -- see kf_forward_rowsplit_mp.hip), per
// observation q, all cross-lane traffic by DPP / v_permlane*_swap except one 256-byte exchange:
//     M_c  = sum_g sum_r P[g+4r][c] Z[g+4r]            column sums over lane groups: permlane swaps
//     F    = H_qq + sum_c Z_c M_c,  Z a = sum_c Z_c a_c  all-reduce inside a DPP row (row_ror)
//     M_row: M_c re-indexed by row through LDS (16 doubles), P[g+4r][c] -= M_{g+4r} M_c / F
// Tapes are stored straight from the accumulator registers: for each r the active lanes cover
// 4 consecutive rows = 4m contiguous doubles.
//
// AUG variant (m + p <= 16, the north-star shape): the zero padding of the tile carries the
// AUGMENTED state (alpha, Z alpha): T_ext = [T 0; Z T 0], R_ext = [R; Z R], so the predicted
// tile already holds M = P Z_q' (column m+q) and Z_q P Z_q' (diagonal entry m+q), and the
// predicted augmented state holds Z_q a.  Observation q then is "component m+q plus noise H_qq":
// the update reads one column of the tile -- no matrix-vector product, no reductions -- and the
// rank-one update of the augmented tile keeps later columns (q' > q) consistent.  The MFMAs do
// this extra work for free inside the padding.
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int TILE = 16;
constexpr int PMAXQ = KF_MP_P_MAX;
// dynamic LDS (doubles): transpose tile | tape staging tile | M exchange | a exchange | ring | diag H | d | y block (64 p)
// | Z, Z permuted (16 p each, non-augmented variant only).  3.9 KB at p = 1: never the occupancy limit.
constexpr int OFF_STG = TILE * TILE;              // tape staging tile (separate from the transpose tile)
constexpr int OFF_EXM = OFF_STG + TILE * TILE;
constexpr int OFF_EXA = OFF_EXM + TILE;
constexpr int OFF_RG = OFF_EXA + TILE;
constexpr int OFF_HD = OFF_RG + 128;
constexpr int OFF_DL = OFF_HD + PMAXQ;
constexpr int OFF_YB = OFF_DL + PMAXQ;
__host__ __device__ inline int mfma_lds_doubles(int p, bool aug) { return OFF_YB + 64 * p + (aug ? 0 : 2 * TILE * p); }
// setup scratch (aliases the loop buffers): T, P0, R, Q, Z, a0, c, R_ext, R_ext Q, Z T, P0 Z', Z P0 Z'
inline int mfma_setup_doubles(int m, int p, int r) {
  return 2 * m * m + m * r + r * r + 3 * p * m + 2 * m + 2 * TILE * r + p * p;
}

__device__ __forceinline__ void st_off(double* base, unsigned byte_off, double v) {
  // uniform base + 32-bit unsigned lane offset: global_store ... saddr form, no 64-bit VALU address math
#ifdef KF_MFMA_NT_STORES
  __builtin_nontemporal_store(v, reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off));
#else
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off) = v;
#endif
}

// AUG: augmented-state variant (m + p <= 4 KS); KS: k-steps / accumulator registers in use;
// P1: p == 1 (requires AUG) -- compile-time column of the tile, lean likelihood bookkeeping.
template <bool AUG, int KS, bool P1>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(4))) void kf_fwd_mfma_kernel(
    const KfFwdArgs A) {
  static_assert(!P1 || AUG, "the p == 1 fast path uses the augmented tile");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const int permc = (c & 3) * 4 + (c >> 2);  // position of index c in the "g + 4r" ordering
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, p = P1 ? 1 : A.p, mm = m * m;
  // augmented components Z_q alpha sit in the LAST p rows/columns of the 4 KS in use
  const int ja = AUG ? 4 * KS - p : TILE;
  const int je = AUG ? 4 * KS : TILE;  // tile indices >= je are padding too (exact zeros)

  double* TR = smem;
  double* STG = smem + OFF_STG;
  double* EXM = smem + OFF_EXM;
  double* EXA = smem + OFF_EXA;
  double* RG = smem + OFF_RG;
  double* HD = smem + OFF_HD;
  double* DL = smem + OFF_DL;
  double* YB = smem + OFF_YB;
  double* ZL = YB + 64 * p;
  double* ZP = ZL + TILE * p;

  // ---- per-draw constants, built through an LDS scratch that aliases the loop buffers ----------------
  // (dependent global loads inside per-lane loops cost ~0.13 ms per launch before this)
  double TA[KS], TAh[KS], RQRh[KS], P[KS];
  bool on[KS];
  unsigned offb[KS];
  const bool c_ok = c < m;
  const double* gP = A.P0 + b * (long)mm;
  const double* gZ = A.Z + b * (long)p * m;
  double cc = 0.0, ac = 0.0;
  {
    double* sT = smem;                 // T            m x m
    double* sP = sT + mm;              // P0           m x m (as given)
    double* sR = sP + mm;              // R            m x r
    double* sQ = sR + m * r_;          // Q            r x r
    double* sZ = sQ + r_ * r_;         // Z            p x m
    double* sa = sZ + p * m;           // a0
    double* sc = sa + m;               // c
    double* sRe = sc + m;              // R_ext        16 x r   rows: R | 0 | Z R
    double* sRQ = sRe + TILE * r_;     // R_ext Q      16 x r
    double* sTZ = sRQ + TILE * r_;     // Z T          p x m
    double* sPZ = sTZ + p * m;         // P0 Z'        m x p
    double* sZPZ = sPZ + m * p;        // Z P0 Z'      p x p
    {
      const double* gT = A.Tm + b * (long)mm;
      const double* gR = A.R + b * (long)m * r_;
      const double* gQ = A.Q + b * (long)r_ * r_;
      for (int e = lane; e < mm; e += KF_WAVE) {
        sT[e] = gT[e];
        sP[e] = gP[e];
      }
      for (int e = lane; e < m * r_; e += KF_WAVE) sR[e] = gR[e];
      for (int e = lane; e < r_ * r_; e += KF_WAVE) sQ[e] = gQ[e];
      for (int e = lane; e < p * m; e += KF_WAVE) sZ[e] = gZ[e];
      if (lane < m) {
        sa[lane] = A.a0[b * (long)m + lane];
        sc[lane] = A.c ? A.c[b * (long)m + lane] : 0.0;
      }
    }
    __syncthreads();
    // tile index i: state component (i < m), padding (exact zeros), or augmented component q = i - ja
    auto P0s = [&](int i, int j) { return (j >= i) ? sP[i * m + j] : sP[j * m + i]; };  // upper triangle
    for (int e = lane; e < TILE * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      if (i < m) {
        v = sR[e];
      } else if (i >= ja && i < je) {
        for (int j = 0; j < m; ++j) v += sZ[(i - ja) * m + j] * sR[j * r_ + l];
      }
      sRe[e] = v;
    }
    if (AUG) {
      for (int e = lane; e < p * m; e += KF_WAVE) {
        const int q = e / m, k = e - q * m;  // (Z T)[q][k] and (P0 Z')[k][q]
        double tz = 0.0, pz = 0.0;
        for (int l = 0; l < m; ++l) {
          tz += sZ[q * m + l] * sT[l * m + k];
          pz += P0s(k, l) * sZ[q * m + l];
        }
        sTZ[e] = tz;
        sPZ[k * p + q] = pz;
      }
    }
    __syncthreads();
    for (int e = lane; e < TILE * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      for (int kk = 0; kk < r_; ++kk) v += sRe[i * r_ + kk] * sQ[kk * r_ + l];
      sRQ[e] = v;
    }
    if (AUG) {
      for (int e = lane; e < p * p; e += KF_WAVE) {
        const int q1 = e / p, q2 = e - q1 * p;
        if (q1 <= q2) {  // one value for both mirrored entries: bitwise symmetric
          double v = 0.0;
          for (int l = 0; l < m; ++l) v += sZ[q1 * m + l] * sPZ[l * p + q2];
          sZPZ[q1 * p + q2] = v;
          sZPZ[q2 * p + q1] = v;
        }
      }
    }
    __syncthreads();
    const bool c_aug = c >= ja && c < je;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + g;  // row index of accumulator register s / k index of MFMA step s
      const bool k_aug = k >= ja && k < je;
      on[s] = c_ok && k < m;    // tapes carry the m x m block only
      offb[s] = (unsigned)((k * m + c) * 8);
      double tv = 0.0, pv = 0.0;
      if (k < m) tv = c_ok ? sT[c * m + k] : (c_aug ? sTZ[(c - ja) * m + k] : 0.0);
      if (k < m && c_ok) pv = P0s(k, c);
      else if (k < m && c_aug) pv = sPZ[k * p + (c - ja)];
      else if (k_aug && c_ok) pv = sPZ[c * p + (k - ja)];
      else if (k_aug && c_aug) pv = sZPZ[(k - ja) * p + (c - ja)];
      TA[s] = tv;
      TAh[s] = 0.5 * tv;
      P[s] = pv;
      double sacc = 0.0;
      for (int l = 0; l < r_; ++l) sacc += sRQ[k * r_ + l] * sRe[c * r_ + l];
      RQRh[s] = 0.5 * sacc;
    }
    if (c_ok) {
      cc = sc[c];
      ac = sa[c];
    } else if (c_aug) {  // AUG: Z_q c and Z_q a0
      for (int l = 0; l < m; ++l) {
        cc += sZ[(c - ja) * m + l] * sc[l];
        ac += sZ[(c - ja) * m + l] * sa[l];
      }
    }
    __syncthreads();  // scratch is dead: the loop buffers may be written
  }
  if (!AUG) {
    for (int e = lane; e < p * TILE; e += KF_WAVE) {
      const int q = e >> 4, k = e & 15;
      const double z = (k < m) ? gZ[q * m + k] : 0.0;
      ZL[e] = z;
      ZP[q * TILE + (k & 3) * 4 + (k >> 2)] = z;
    }
  }
  if (lane < p) {
    HD[lane] = A.H[b * (long)p * p + lane * p + lane];
    DL[lane] = A.d ? A.d[b * (long)p + lane] : 0.0;
  }
  const double H1 = P1 ? A.H[b] : 0.0;
  const double d1 = (P1 && A.d) ? A.d[b] : 0.0;

  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double joint_const = (A.filter == KF_FILTER_CHOLESKY) ? (double)p : 1.0;

  const bool vec_on = g == 0 && c_ok;
  const unsigned voff = (unsigned)(c * 8);
  double* o_fa = A.filt_a ? A.filt_a + b * (long)n * m : nullptr;
  double* o_pa = A.pred_a ? A.pred_a + b * (long)(n + 1) * m : nullptr;
  double* o_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* o_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;

  // predicted tape slot 0 = (a0, P0 as given)
  if (o_pa) {
    if (vec_on) st_off(o_pa, voff, ac);
    o_pa += m;
  }
  if (o_pP) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
      if (on[s]) st_off(o_pP, offb[s], gP[offb[s] >> 3]);
    o_pP += mm;
  }

  // covariance tapes: registers -> compact m x m image in LDS (tile STG) -> 16 bytes per lane,
  // consecutive lanes on consecutive addresses (full lines).  Staging and flushing are separate
  // so that the LDS round trip hides behind independent work (the MFMAs / the next update).
  const int nb128 = mm >> 1;
  auto stage_cov = [&](const double (&Preg)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
      if (on[s]) STG[offb[s] >> 3] = Preg[s];
  };
  // lane predicates fixed for the whole launch: plain exec-masked stores, no branch chains in the loop
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
  bool pred_pending = false;  // a predicted covariance sits in STG, flushed inside the next update

  double logp = 0.0;  // every lane accumulates the same total
  int info = 0;

  // evaluate the ll of the steps [tb, tb+cnt) held in the ring: one step per lane
  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
    const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * lane);
    const bool act = lane < cnt;
    double llv;
    if (P1) {
      // (F, v^2/F); F = -1: missing; F = -2: observed but skipped (Univariate, F == 0)
      const double Fv = act ? fq.x : -1.0;
      if (Fv == -1.0) llv = 0.0;
      else if (Fv == -2.0) llv = exact_const ? -0.5 * KF_LOG_2PI : 0.0;
      else if (univariate && !exact_const && Fv == 1.0 && fq.y == 0.0) llv = 0.0;  // :497-500
      else llv = -0.5 * (KF_LOG_2PI + log(Fv) + fq.y);
    } else {
      // (mantissa product, exponents ln 2 + quadratic form + n log 2 pi); -1: no contribution
      const double xv = act ? fq.x : -1.0;
      llv = (xv == -1.0) ? 0.0 : -0.5 * (log(xv) + fq.y);
    }
    if (act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
  };

  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L);

  for (int t = 0; t < n; ++t) {
    if ((t & 63) == 0) {
      const int cnt = ((n - t) < 64 ? (n - t) : 64) * p;
      for (int e = lane; e < cnt; e += KF_WAVE) YB[e] = ybase[(long)t * p + e];
      wave_lds_fence();
    }
    const double* yrow = YB + (t & 63) * p;
    double mant = 1.0, qs = 0.0;
    int ex = 0, cnt_ll = 0, nob = 0;
    bool bad = false;
#pragma unroll 1
    for (int q = 0; q < p; ++q) {
      const double yq = yrow[q];
      const bool obs = !(yq != yq);
      const double Hq = P1 ? H1 : HD[q], dq = P1 ? d1 : DL[q];
      double F, zs, Mc;
      if (P1) {
        // column 4 KS - 1 of the tile IS M = P Z'; its diagonal entry is Z P Z'
        constexpr int jst = 4 * KS - 1;
        Mc = group_bcast<3>(P[KS - 1]);
        if (g == 0) EXM[permc] = Mc;
        F = Hq + readlane_f64(Mc, jst);
        zs = readlane_f64(ac, jst);
      } else if (AUG) {
        const int jst = ja + q, gst = jst & 3, sst = jst >> 2;
        double psel = P[0];
#pragma unroll
        for (int s = 1; s < KS; ++s) psel = (sst == s) ? P[s] : psel;
        Mc = col_allsum(g == gst ? psel : 0.0);  // row ja+q, broadcast from its lane group
        if (g == 0) EXM[permc] = Mc;
        F = Hq + readlane_f64(Mc, jst);
        zs = readlane_f64(ac, jst);
      } else {
        const double zc = ZL[q * TILE + c];
        const double2_t z01 = *reinterpret_cast<const double2_t*>(ZP + q * TILE + 4 * g);
        const double2_t z23 = *reinterpret_cast<const double2_t*>(ZP + q * TILE + 4 * g + 2);
        const double zr[4] = {z01.x, z01.y, z23.x, z23.y};
        // ---- M_c = (P Z_q')_c: in-lane over r, then over the lane groups ----------------------------
        double pc = P[0] * zr[0];
#pragma unroll
        for (int s = 1; s < KS; ++s) pc = fma(P[s], zr[s], pc);
        Mc = col_allsum(pc);
        if (g == 0) EXM[permc] = Mc;
        F = Hq + row_allsum(zc * Mc);
        zs = row_allsum(zc * ac);
      }
      double v = (yq + dsign * dq) - zs;
      // missing -> skipped; UnivariateFilter also skips F == 0 (:468-471)
      const bool degenerate = !obs || (univariate && F == 0.0);
      const bool notpd = !degenerate && !(F > 0.0);
      if (notpd && info == 0) info = ((t + 1) << 8) | (q + 1);
      const double invF = degenerate ? 0.0 : fast_rcp(F);
      if (degenerate) v = 0.0;
      if (P1) {
        if (lane == 0) {
          double2_t fq = {degenerate ? (obs ? -2.0 : -1.0) : F, v * v * invF};
          *reinterpret_cast<double2_t*>(RG + 2 * (t & 63)) = fq;
        }
      } else {
        bad = bad || notpd;
        nob += obs ? 1 : 0;
        if (!degenerate) {
          int e2;
          mant *= frexp(F, &e2);
          ex += e2;
          qs = fma(v * v, invF, qs);
          cnt_ll += (F == 1.0 && v == 0.0) ? 0 : 1;  // :497-500 counts non-zero scalar terms
        }
      }
      ac = fma(Mc * invF, v, ac);
      if ((P1 || q == p - 1) && g == 0) EXA[permc] = ac;
      wave_lds_fence();
      double Mr[4];
      {
        const double2_t m01 = *reinterpret_cast<const double2_t*>(EXM + 4 * g);
        const double2_t m23 = *reinterpret_cast<const double2_t*>(EXM + 4 * g + 2);
        Mr[0] = m01.x; Mr[1] = m01.y; Mr[2] = m23.x; Mr[3] = m23.y;
      }
      if (q == 0 && pred_pending) {  // predicted covariance of the previous step: tile -> HBM
        flush_cov(o_pP);
        o_pP += mm;
        pred_pending = false;
      }
      // P -= M M'/F, bitwise symmetric (M_i * M_j commutes)
#pragma unroll
      for (int s = 0; s < KS; ++s) P[s] = fma(-(Mr[s] * Mc), invF, P[s]);
    }
    double ar[4];
    {
      const double2_t a01 = *reinterpret_cast<const double2_t*>(EXA + 4 * g);
      const double2_t a23 = *reinterpret_cast<const double2_t*>(EXA + 4 * g + 2);
      ar[0] = a01.x; ar[1] = a01.y; ar[2] = a23.x; ar[3] = a23.y;
    }
    if (!P1 && lane == 0) {
      double nconst, xv;
      if (univariate) {
        nconst = exact_const ? (double)nob : (double)cnt_ll;
        xv = mant;
      } else {
        nconst = exact_const ? (double)nob : joint_const;
        xv = (nob == 0) ? -1.0 : mant;  // all-missing step: ll = 0
      }
      if (bad) xv = __builtin_nan("");
      double2_t fq = {xv, fma(0.69314718055994530942, (double)ex, qs) + nconst * KF_LOG_2PI};
      if (!univariate && nob == 0) fq.y = 0.0;
      *reinterpret_cast<double2_t*>(RG + 2 * (t & 63)) = fq;
    }
    // ---- filtered tapes -----------------------------------------------------------------------------
    if (o_fa) {
      if (vec_on) st_off(o_fa, voff, ac);
      o_fa += m;
    }
    if (o_fP) stage_cov(P);  // flushed below, behind the MFMAs
    // ---- predict: a_hat = T a_f + c ; P_hat = sym(T P_f T' + RQR') ----------------------------------------
    double part = TA[0] * ar[0];
#pragma unroll
    for (int s = 1; s < KS; ++s) part = fma(TA[s], ar[s], part);
    ac = cc + col_allsum(part);
    double4_t W2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) W2 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], TA[s], W2, 0, 0, 0);
    double4_t S = {RQRh[0], KS > 1 ? RQRh[KS > 1 ? 1 : 0] : 0.0, KS > 2 ? RQRh[KS > 2 ? 2 : 0] : 0.0,
                   KS > 3 ? RQRh[KS > 3 ? 3 : 0] : 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) S = __builtin_amdgcn_mfma_f64_16x16x4f64(TAh[s], W2[s], S, 0, 0, 0);
    if (o_fP) {
      flush_cov(o_fP);
      o_fP += mm;
    }
    // transpose S/2 through LDS: element (i, j) at TR[i*16 + perm(j)]
#pragma unroll
    for (int s = 0; s < KS; ++s) TR[(4 * s + g) * TILE + permc] = S[s];
    wave_lds_fence();
    {
      const double2_t t01 = *reinterpret_cast<const double2_t*>(TR + c * TILE + 4 * g);
      P[0] = S[0] + t01.x;
      if (KS > 1) P[KS > 1 ? 1 : 0] = S[1] + t01.y;
      if (KS > 2) {
        const double2_t t23 = *reinterpret_cast<const double2_t*>(TR + c * TILE + 4 * g + 2);
        P[KS > 2 ? 2 : 0] = S[2] + t23.x;
        if (KS > 3) P[KS > 3 ? 3 : 0] = S[3] + t23.y;
      }
    }
    if (o_pa) {
      if (vec_on) st_off(o_pa, voff, ac);
      o_pa += m;
    }
    if (o_pP) {
      stage_cov(P);
      pred_pending = true;
    }
    if ((t & 63) == 63) flush_ring(t - 63, 64);
  }
  if (pred_pending) flush_cov(o_pP);
  if (n & 63) flush_ring(n & ~63, n & 63);
  if (lane == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

int kf_launch_forward_mfma(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  if (a.m > TILE || a.p > PMAXQ || a.p < 1) {
    KF_SET_ERR(h, "forward(mfma): m=%d p=%d out of range (m <= %d, p <= %d)", a.m, a.p, TILE, PMAXQ);
    return KF_ERR_UNSUPPORTED;
  }
  const bool aug = a.m + a.p <= TILE;  // room for the augmented state (alpha, Z alpha) in the padding
  const int lds_loop = mfma_lds_doubles(a.p, aug), lds_setup = mfma_setup_doubles(a.m, a.p, a.r);
  const size_t lds = sizeof(double) * (size_t)(lds_loop > lds_setup ? lds_loop : lds_setup);
  const int ks = ((aug ? a.m + a.p : a.m) + 3) / 4;
#define KF_MFMA_LAUNCH(AUG_, KS_, P1_) \
  hipLaunchKernelGGL((kf_fwd_mfma_kernel<AUG_, KS_, P1_>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a)
#define KF_MFMA_KS(AUG_, P1_)                       \
  switch (ks) {                                     \
    case 1: KF_MFMA_LAUNCH(AUG_, 1, P1_); break;    \
    case 2: KF_MFMA_LAUNCH(AUG_, 2, P1_); break;    \
    case 3: KF_MFMA_LAUNCH(AUG_, 3, P1_); break;    \
    default: KF_MFMA_LAUNCH(AUG_, 4, P1_); break;   \
  }
  if (aug && a.p == 1) {
    KF_MFMA_KS(true, true)
  } else if (aug) {
    KF_MFMA_KS(true, false)
  } else {
    KF_MFMA_KS(false, false)
  }
#undef KF_MFMA_KS
#undef KF_MFMA_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
