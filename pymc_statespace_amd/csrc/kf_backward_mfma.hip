// kf_backward_mfma.hip -- reverse-mode gradient of the Kalman log-likelihood with the covariance
// and its adjoint in MFMA accumulator layout (one wavefront per draw, k_states <= 16, p <= 8
// observations processed one at a time, static system matrices).  Companion of
// kf_forward_mfma.hip; consumes the predicted tapes of any forward kernel.
//
// Maths = kf_backward_rowsplit.hip's (derivation: oracle/kalman_adjoint_numpy.py): with scalar
// observations the update is P_f = P - (P z)(z'P)/F and its adjoint is a handful of rank-one
// terms; only the predict step keeps m x m x m products, and those run as MFMAs whose operands
// are the register tiles themselves (kf_tile.h: a tile's registers are X as B operand and X' as
// A operand):
//     W   = T P_f          A = T' tile (static)   B = P_f
//     Tb += 2 Gm W         A = Gm (symmetric)     B = W        accumulated in the MFMA C operand
//     X1  = Gm T           A = Gm                 B = T tile (static)
//     Sb  = T' X1          A = T tile             B = X1
// Gm = sym(adjoint of the predicted P_{t+1}).  The adjoint of a SYMMETRIC intermediate may be
// symmetrised freely (only its symmetric part reaches the gradients of the inputs), so the
// adjoint tile is kept symmetric:  Sb <- Sb + Fb z z' + (u z' + z u')/2,  u = (ab_f v - 2 Sb M)/F.
// Per time step (reverse): recompute the p scalar updates from the predicted tape (keeping
// M_q, F_q, v_q, a^q in LDS), run the predict adjoint, then walk the observations backwards,
// rebuilding P^q = P^{q+1} + M_q M_q'/F_q.
//
// Outputs: as kf_filter_backward documents.  g_P0 is exact (not symmetrised) for p == 1; for
// p > 1 it is the symmetric part (the joint Standard/Cholesky update and the sequential one are
// the same function of a symmetric P0, their entrywise partials differ by an antisymmetric
// matrix), and g_H carries the diagonal only (the caller asserted a diagonal H).
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int TILE = 16;
constexpr int PMAXQ = KF_MP_P_MAX;

// dynamic LDS (doubles)
__host__ __device__ inline int bwd_mfma_lds_doubles(int p) {
  return 16 /*EX*/ + 2 * PMAXQ /*HD, DL*/ + 2 * PMAXQ /*HB, DB*/ + p * (2 * 16 /*ZC, ZR*/ + 3 * 16 /*MC, MP, AQ*/ + 4 /*SC*/ + 16 /*ZB*/ + 64 /*YB*/) + 64 /*GL*/ + 256 /*final tile*/ + 8 * KF_WAVE /*ACC*/ + 64 /*GY*/;
}

// (capping the allocation at 128 VGPRs for a fourth wavefront per SIMD spills ~30 values and is
// slower: 5.2 vs 3.9 ms at the north-star shape)
// P1: p == 1 known at compile time -- Z, M, F, v of the step stay in registers between the
// forward recompute and the adjoint instead of going through LDS
// TV (round 5): time-varying T / Z / H / c / d ([B][T][..], KF_FLAG_TV_*; R and Q static): step t loads slice t of every flagged
// array -- T_t for the predict adjoint, Z_t / H_t / d_t for the update -- and its gradient is that step's term alone, written to slice
// t of the gradient tape instead of accumulated (the indexing of kf_backward_lds.hip, which this replaces for m <= 16 and scalar
// observations: 15 -> 4 ms at the headline shape).
// TP (round 6): cotangents of the filter's TAPE outputs (kf_filter_backward_tapes; KfBwdArgs::g_filt_a / g_pred_a / g_filt_P / g_pred_P,
// any of them NULL = 0) -- the predicted slots' join (abp, PB) at the end of each step (slot T before the sweep), the filtered ones'
// join T' ab' and Sb; covariance cotangents enter by their symmetric part (the adjoint tile is kept symmetric, see above), except
// predicted slot 0 = P0 as given, whose cotangent goes to g_P0 unchanged.  Loaded at the top of the step, used behind the forward
// recompute.  A separate instantiation: the log-likelihood-only adjoint is the kernel it was.
// W4 (TP only): four wavefronts per SIMD at the price of spills -- the cotangents are then loaded where they are used, not at the top
// of the step.  Pays when the batch does not fit one round of three (measured, profiles/r06/tape_grad_time.txt: headline 4096 draws
// 8.4 -> 6.7 ms; the 2048-draw C4 shard 3.9 -> 4.7, so only beyond 3 x SIMDs draws).
template <int KS, bool P1, bool TV = false, bool TP = false, bool W4 = false>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu((KS <= 3 && !(TV && KS >= 2) && (!TP || W4)) ? 4 : 3))) void kf_bwd_mfma_kernel(
    const KfBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const int permc = (c & 3) * 4 + (c >> 2);
  const long b = blockIdx.x;
  const int n = A.steps ? A.steps[b] : A.T;  // (per-draw step counts: the steady-state adjoint took the steps behind)
  const int nt = A.tape_T, r_ = A.r, m = A.m, p = P1 ? 1 : A.p, mm = m * m;  // nt: tape stride (>= n)

  double* EX = smem;                 // exchange: permuted write, two b128 reads give the row-indexed view
  double* HD = EX + 16;
  double* DL = HD + PMAXQ;
  double* HB = DL + PMAXQ;           // final g_H diagonal / g_d
  double* DB = HB + PMAXQ;
  double* ZC = DB + PMAXQ;           // Z_q, column-indexed (zero padded)
  double* ZR = ZC + 16 * p;          // Z_q, permuted
  double* MC = ZR + 16 * p;          // M_q of the current step, column-indexed
  double* MP = MC + 16 * p;          //   permuted
  double* AQ = MP + 16 * p;          // a^q (state before observation q)
  double* SC = AQ + 16 * p;          // F_q, v_q, 1/F_q
  double* ZB = SC + 4 * p;           // accumulated g_Z
  double* YB = ZB + 16 * p;          // y block of 64 steps
  double* GL = YB + 64 * p;          // g_ll_obs block
  double* FT = GL + 64;              // 16 x 16 tile: transpose of X1 per step, Gs at the end
  // per-lane accumulators that are touched once per step live in LDS, not in VGPRs (slot s of lane
  // l at ACC[s*64 + l]: conflict-free): Gs = sum Gm (slots 0..3), sum ab' a_f' (slots 4..7)
  double* ACC = FT + 256;
  double* GY = ACC + 8 * KF_WAVE;       // (g_yy requested) sum_t vb_q (y -+ d)_q', 8 x 8

  const bool c_ok = c < m;
  const unsigned tvf = TV ? (A.flags & KF_FLAG_TV_ANY) : 0u;
  const bool tv_T = TV && (tvf & KF_FLAG_TV_T), tv_Z = TV && (tvf & KF_FLAG_TV_Z), tv_H = TV && (tvf & KF_FLAG_TV_H),
             tv_C = TV && (tvf & KF_FLAG_TV_C), tv_D = TV && (tvf & KF_FLAG_TV_D);
  const long bT = TV ? b * (long)A.T : 0L;   // slice (b, t) of a time-varying array is (bT + t)
  double TA[KS], TB[KS];  // T' tile (TA[s] = T[c][4s+g]) and T tile (TB[s] = T[4s+g][c])
  bool on[KS];
  int off[KS];
  {
    const double* gT = A.Tm + (tv_T ? bT : b) * (long)mm;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + g;
      on[s] = c_ok && k < m;
      off[s] = k * m + c;
      TA[s] = on[s] ? gT[c * m + k] : 0.0;
      TB[s] = on[s] ? gT[k * m + c] : 0.0;
    }
    const double* gZ = A.Z + (tv_Z ? bT : b) * (long)p * m;
    for (int e = lane; e < p * TILE; e += KF_WAVE) {
      const int q = e >> 4, k = e & 15;
      const double z = (k < m) ? gZ[q * m + k] : 0.0;
      ZC[e] = z;
      ZR[q * TILE + (k & 3) * 4 + (k >> 2)] = z;
      ZB[e] = 0.0;
    }
    if (lane < p) {
      HD[lane] = A.H[(tv_H ? bT : b) * (long)p * p + lane * p + lane];
      DL[lane] = A.d ? A.d[(tv_D ? bT : b) * (long)p + lane] : 0.0;
    }
  }
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const bool y_per_draw = (A.flags & KF_FLAG_Y_PER_DRAW) != 0;
  const double* ybase = A.y + (y_per_draw ? b * (long)A.y_T * p : 0L);
  const double* gpa = A.pred_a + b * (long)(nt + 1) * m;
  const double* gpP = A.pred_P + b * (long)(nt + 1) * mm;
  const double g_all = A.g_logp ? A.g_logp[b] : 0.0;
  const double* gll = A.g_ll_obs ? A.g_ll_obs + b * (long)A.y_T : nullptr;

  double zc1 = 0.0, zr1[4] = {0.0, 0.0, 0.0, 0.0}, H1 = 0.0, d1 = 0.0;
  if (P1) {
    const double* gZ = A.Z + (tv_Z ? bT : b) * (long)m;
    zc1 = c_ok ? gZ[c] : 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) zr1[s] = (4 * s + g < m) ? gZ[4 * s + g] : 0.0;
    H1 = A.H[tv_H ? bT : b];
    d1 = A.d ? A.d[tv_D ? bT : b] : 0.0;
  }
  // adjoint state and accumulators
  double PB[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    PB[s] = (A.init_Pb && on[s]) ? A.init_Pb[b * (long)mm + off[s]] : 0.0;
    ACC[s * KF_WAVE + lane] = 0.0;
    ACC[(4 + s) * KF_WAVE + lane] = 0.0;
  }
  double4_t TBm = {0.0, 0.0, 0.0, 0.0};
  double abp = (A.init_ab && c_ok) ? A.init_ab[b * (long)m + c] : 0.0, cb = 0.0, hbv = 0.0, dbv = 0.0;
  const double* c_fa = (TP && A.g_filt_a) ? A.g_filt_a + b * (long)nt * m : nullptr;
  const double* c_pa = (TP && A.g_pred_a) ? A.g_pred_a + b * (long)(nt + 1) * m : nullptr;
  const double* c_fP = (TP && A.g_filt_P) ? A.g_filt_P + b * (long)nt * mm : nullptr;
  const double* c_pP = (TP && A.g_pred_P) ? A.g_pred_P + b * (long)(nt + 1) * mm : nullptr;
  auto sym_at = [&](const double* tape, long slot, int s) -> double {   // symmetric part of a covariance cotangent at [4 s + g][c]
    const double* q = tape + slot * mm;
    return on[s] ? 0.5 * (q[off[s]] + q[c * m + 4 * s + g]) : 0.0;
  };
  if (TP) {   // the cotangent of predicted slot T
    if (c_pa && c_ok) abp += c_pa[(long)n * m + c];
    if (c_pP) {
#pragma unroll
      for (int s = 0; s < KS; ++s) PB[s] += sym_at(c_pP, n, s);
    }
  }
  const bool acc_out = A.accumulate != 0;
  const bool want_yy = A.g_yy != nullptr;
  GY[lane] = 0.0;
  auto put = [&](double* dst, double v) { *dst = acc_out ? *dst + v : v; };

  double PN[KS], an;
  double TN[KS], zcN = 0.0, zrN[4] = {0.0, 0.0, 0.0, 0.0}, HN = 0.0, dN = 0.0;   // TV: slice t of T (and, p == 1, of Z / H / d)
  auto prefetch = [&](int t) {
    const double* rp = gpP + (long)t * mm;
#pragma unroll
    for (int s = 0; s < KS; ++s) PN[s] = on[s] ? rp[off[s]] : 0.0;
    an = c_ok ? gpa[(long)t * m + c] : 0.0;
    if (TV) {
      if (tv_T) {
        const double* tp = A.Tm + (bT + t) * (long)mm;
#pragma unroll
        for (int s = 0; s < KS; ++s) TN[s] = on[s] ? tp[off[s]] : 0.0;
      }
      if (P1) {
        if (tv_Z) {
          const double* zp = A.Z + (bT + t) * (long)m;
          zcN = c_ok ? zp[c] : 0.0;
#pragma unroll
          for (int s = 0; s < 4; ++s) zrN[s] = (4 * s + g < m) ? zp[4 * s + g] : 0.0;
        }
        if (tv_H) HN = A.H[bT + t];
        if (tv_D && A.d) dN = A.d[bT + t];
      }
    }
  };
  auto stage_block = [&](int t) {  // y and g_ll_obs of the 64-step block that holds t
    const int t0 = t & ~63;
    const int cnt = (n - t0 < 64 ? n - t0 : 64);
    for (int e = lane; e < cnt * p; e += KF_WAVE) YB[e] = ybase[(long)t0 * p + e];
    if (lane < cnt) GL[lane] = gll ? gll[t0 + lane] : 0.0;
    __syncthreads();
  };
  prefetch(n - 1);
  stage_block(n - 1);

  for (int t = n - 1; t >= 0; --t) {
    double P[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) P[s] = PN[s];
    double ac = an;
    if (TV) {
      if (tv_T) {
#pragma unroll
        for (int s = 0; s < KS; ++s) TB[s] = TN[s];
      }
      if (P1) {
        if (tv_Z) {
          zc1 = zcN;
#pragma unroll
          for (int s = 0; s < 4; ++s) zr1[s] = zrN[s];
        }
        if (tv_H) H1 = HN;
        if (tv_D) d1 = dN;
      } else if (tv_Z || tv_H || tv_D) {   // p > 1: slice t of Z / H / d into the LDS copies
        __syncthreads();
        if (tv_Z) {
          const double* zp = A.Z + (bT + t) * (long)p * m;
          for (int e = lane; e < p * TILE; e += KF_WAVE) {
            const int q = e >> 4, k = e & 15;
            const double z = (k < m) ? zp[q * m + k] : 0.0;
            ZC[e] = z;
            ZR[q * TILE + (k & 3) * 4 + (k >> 2)] = z;
          }
        }
        if (lane < p) {
          if (tv_H) HD[lane] = A.H[(bT + t) * (long)p * p + lane * p + lane];
          if (tv_D && A.d) DL[lane] = A.d[(bT + t) * (long)p + lane];
        }
        __syncthreads();
      }
    }
    if (t > 0) prefetch(t - 1);
    double k_fa = 0.0, k_pa = 0.0, k_fP[KS], k_pP[KS];   // (TP) this step's tape cotangents
    if (TP && !W4) {
      k_fa = (c_fa && c_ok) ? c_fa[(long)t * m + c] : 0.0;
      k_pa = (c_pa && c_ok) ? c_pa[(long)t * m + c] : 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        k_fP[s] = c_fP ? sym_at(c_fP, t, s) : 0.0;
        k_pP[s] = (c_pP && t > 0) ? sym_at(c_pP, t, s) : 0.0;   // (slot 0: straight into g_P0, below)
      }
    }
    if ((t & 63) == 63 && t != n - 1) stage_block(t);
    const double* yrow = YB + (t & 63) * p;
    const double gt = g_all + GL[t & 63];

    double k_Mc = 0.0, k_aq = 0.0, k_v = 0.0, k_invF = 0.0, k_Mr[4] = {0.0, 0.0, 0.0, 0.0};  // P1: kept in registers
    // ---- forward recompute of the p scalar updates (kf_forward_mfma.hip, non-augmented form) -------
#pragma unroll 1
    for (int q = 0; q < p; ++q) {
      const double yq = yrow[q];
      const bool obs = !(yq != yq);
      double zc, zr[4];
      if (P1) {
        zc = zc1;
#pragma unroll
        for (int s = 0; s < 4; ++s) zr[s] = zr1[s];
      } else {
        zc = ZC[q * TILE + c];
        const double2_t z01 = *reinterpret_cast<const double2_t*>(ZR + q * TILE + 4 * g);
        const double2_t z23 = *reinterpret_cast<const double2_t*>(ZR + q * TILE + 4 * g + 2);
        zr[0] = z01.x; zr[1] = z01.y; zr[2] = z23.x; zr[3] = z23.y;
      }
      double pc = P[0] * zr[0];
#pragma unroll
      for (int s = 1; s < KS; ++s) pc = fma(P[s], zr[s], pc);
      const double Mc = col_allsum(pc);
      const double F = (P1 ? H1 : HD[q]) + row_allsum(zc * Mc);
      const double zs = row_allsum(zc * ac);
      const bool degenerate = !obs || (univariate && F == 0.0);
      const double invF = degenerate ? 0.0 : fast_rcp(F);
      const double v = degenerate ? 0.0 : (yq + dsign * (P1 ? d1 : DL[q])) - zs;
      if (g == 0) {
        if (!P1) MC[q * TILE + c] = Mc;
        MP[q * TILE + permc] = Mc;
        if (!P1) AQ[q * TILE + c] = ac;
      }
      if (!P1 && lane == 0) {
        SC[4 * q] = F;
        SC[4 * q + 1] = v;
        SC[4 * q + 2] = invF;
      }
      __syncthreads();
      const double2_t m01 = *reinterpret_cast<const double2_t*>(MP + q * TILE + 4 * g);
      const double2_t m23 = *reinterpret_cast<const double2_t*>(MP + q * TILE + 4 * g + 2);
      const double Mr[4] = {m01.x, m01.y, m23.x, m23.y};
      if (P1) {
        k_Mc = Mc; k_aq = ac; k_v = v; k_invF = invF;
#pragma unroll
        for (int s = 0; s < 4; ++s) k_Mr[s] = Mr[s];
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) P[s] = fma(-(Mr[s] * Mc), invF, P[s]);
      ac = fma(Mc * invF, v, ac);
    }
    // P = P_f, ac = a_f

    // ---- adjoint of predict ---------------------------------------------------------------------------
    if (g == 0) EX[permc] = abp;
    __syncthreads();
    double abr[4];
    {
      const double2_t a01 = *reinterpret_cast<const double2_t*>(EX + 4 * g);
      const double2_t a23 = *reinterpret_cast<const double2_t*>(EX + 4 * g + 2);
      abr[0] = a01.x; abr[1] = a01.y; abr[2] = a23.x; abr[3] = a23.y;
    }
    if (tv_C) {
      if (g == 0 && c_ok && A.g_c) A.g_c[(bT + t) * (long)m + c] = abp;
    } else {
      cb += abp;
    }
    double part = TB[0] * abr[0];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (!tv_T) ACC[(4 + s) * KF_WAVE + lane] = fma(abr[s], ac, ACC[(4 + s) * KF_WAVE + lane]);  // ab' a_f'
      ACC[s * KF_WAVE + lane] += PB[s];                                                // adjoint of R Q R'
      if (s > 0) part = fma(TB[s], abr[s], part);
    }
    double abf = col_allsum(part);  // T' ab'
    if (TP && W4) {
      k_fa = (c_fa && c_ok) ? c_fa[(long)t * m + c] : 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) k_fP[s] = c_fP ? sym_at(c_fP, t, s) : 0.0;
    }
    if (TP) abf += k_fa;
    // X1 = Gm T ; Sb = T' X1 ; Tb += 2 X1 P_f  (A operand = X1' registers, from an LDS transpose:
    // cheaper than the fourth product W = T P_f, Gm W)
    double4_t X1 = {0.0, 0.0, 0.0, 0.0}, SBv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) X1 = __builtin_amdgcn_mfma_f64_16x16x4f64(PB[s], TB[s], X1, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) FT[(4 * s + g) * TILE + permc] = X1[s];
#pragma unroll
    for (int s = 0; s < KS; ++s) SBv = __builtin_amdgcn_mfma_f64_16x16x4f64(TB[s], X1[s], SBv, 0, 0, 0);
    __syncthreads();
    {
      // X1t registers: X1t[4s+g][c] = X1[c][4s+g]
      const double2_t t01 = *reinterpret_cast<const double2_t*>(FT + c * TILE + 4 * g);
      const double2_t t23 = *reinterpret_cast<const double2_t*>(FT + c * TILE + 4 * g + 2);
      const double X1t[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
      for (int s = 0; s < KS; ++s) TBm = __builtin_amdgcn_mfma_f64_16x16x4f64(X1t[s], P[s], TBm, 0, 0, 0);
    }
    if (tv_T) {   // g_T[t] = 2 X1' P_f + ab' a_f'  (this step's term alone)
      if (A.g_T) {
        double* o = A.g_T + (bT + t) * (long)mm;
#pragma unroll
        for (int s = 0; s < KS; ++s)
          if (on[s]) o[off[s]] = fma(2.0, TBm[s], abr[s] * ac);
      }
      TBm = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    double SB[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) SB[s] = TP ? SBv[s] + k_fP[s] : SBv[s];

    // ---- adjoint of the scalar updates, last observation first ---------------------------------------
#pragma unroll 1
    for (int q = p - 1; q >= 0; --q) {
      double Mc, aq, zc, Mr[4], zr[4], v, invF;
      if (P1) {
        Mc = k_Mc; aq = k_aq; zc = zc1; v = k_v; invF = k_invF;
#pragma unroll
        for (int s = 0; s < 4; ++s) { Mr[s] = k_Mr[s]; zr[s] = zr1[s]; }
      } else {
        Mc = MC[q * TILE + c];
        aq = AQ[q * TILE + c];
        zc = ZC[q * TILE + c];
        const double2_t m01 = *reinterpret_cast<const double2_t*>(MP + q * TILE + 4 * g);
        const double2_t m23 = *reinterpret_cast<const double2_t*>(MP + q * TILE + 4 * g + 2);
        const double2_t z01 = *reinterpret_cast<const double2_t*>(ZR + q * TILE + 4 * g);
        const double2_t z23 = *reinterpret_cast<const double2_t*>(ZR + q * TILE + 4 * g + 2);
        Mr[0] = m01.x; Mr[1] = m01.y; Mr[2] = m23.x; Mr[3] = m23.y;
        zr[0] = z01.x; zr[1] = z01.y; zr[2] = z23.x; zr[3] = z23.y;
        v = SC[4 * q + 1];
        invF = SC[4 * q + 2];
      }
      // P^q = P^{q+1} + M M'/F ;  s = Sb M
      double sp = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        P[s] = fma(Mr[s] * Mc, invF, P[s]);
        sp = fma(SB[s], Mr[s], sp);
      }
      const double sc = col_allsum(sp);
      const double beta = row_allsum(Mc * abf);
      const double gamma = row_allsum(Mc * sc);
      const double Fb = (gamma - beta * v) * invF * invF - 0.5 * gt * (invF - v * v * invF * invF);
      const double vb = (beta - gt * v) * invF;
      const double uc = (abf * v - 2.0 * sc) * invF;  // m1b + m2b
      if (g == 0) EX[permc] = uc;
      __syncthreads();
      const double2_t u01 = *reinterpret_cast<const double2_t*>(EX + 4 * g);
      const double2_t u23 = *reinterpret_cast<const double2_t*>(EX + 4 * g + 2);
      const double ur[4] = {u01.x, u01.y, u23.x, u23.y};
      double pp = P[0] * ur[0];
#pragma unroll
      for (int s = 1; s < KS; ++s) pp = fma(P[s], ur[s], pp);
      const double pdot = col_allsum(pp);  // P^q (m1b + m2b)
      if (tv_Z) {
        if (g == 0 && c_ok && A.g_Z) A.g_Z[((bT + t) * (long)p + q) * m + c] = fma(-aq, vb, fma(2.0 * Fb, Mc, pdot));
      } else if (g == 0) {
        ZB[q * TILE + c] += fma(-aq, vb, fma(2.0 * Fb, Mc, pdot));
      }
      if (p == 1 && t == 0 && A.g_P0) {  // (p == 1 at run time also takes this path)
        // exact (not symmetrised) entrywise gradient: Sb + (Fb z + m1b) z' - z s'/F
        __syncthreads();
        if (g == 0) EX[permc] = (abf * v - sc) * invF;  // m1b
        __syncthreads();
        const double2_t n01 = *reinterpret_cast<const double2_t*>(EX + 4 * g);
        const double2_t n23 = *reinterpret_cast<const double2_t*>(EX + 4 * g + 2);
        const double m1r[4] = {n01.x, n01.y, n23.x, n23.y};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          double gp = SB[s];
          gp = fma(fma(Fb, zr[s], m1r[s]), zc, gp);
          gp = fma(-zr[s] * invF, sc, gp);
          if (TP && c_pP && on[s]) gp += c_pP[off[s]];
          if (on[s]) A.g_P0[b * (long)mm + off[s]] = gp;
        }
      }
      // symmetrised adjoint of P^q ; adjoint of a^q
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        double sb = SB[s];
        sb = fma(Fb * zr[s], zc, sb);
        sb = fma(0.5 * ur[s], zc, sb);
        sb = fma(0.5 * zr[s], uc, sb);
        SB[s] = sb;
      }
      abf = fma(-zc, vb, abf);
      if (c == q) {
        hbv = tv_H ? Fb : hbv + Fb;
        dbv = tv_D ? dsign * vb : dbv + dsign * vb;
      }
      if (want_yy && g == 0 && c < p) {   // (vb = 0 on a missing / skipped observation: its y never counts)
        const double yc = yrow[c];
        GY[q * 8 + c] = fma(vb, (yc != yc) ? 0.0 : yc + dsign * DL[c], GY[q * 8 + c]);
      }
    }
    if (TP && W4) {
      k_pa = (c_pa && c_ok) ? c_pa[(long)t * m + c] : 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) k_pP[s] = (c_pP && t > 0) ? sym_at(c_pP, t, s) : 0.0;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) PB[s] = TP ? SB[s] + k_pP[s] : SB[s];
    abp = TP ? abf + k_pa : abf;
    if (tv_H || tv_D) {   // slice t of g_H (diagonal; the rest zeros) / g_d: lane (g = 0, c = q) holds this step's term
      if (g == 0 && c < p) {
        if (tv_D && A.g_d) A.g_d[(bT + t) * (long)p + c] = dbv;
        if (tv_H && A.g_H)
          for (int q2 = 0; q2 < p; ++q2) A.g_H[((bT + t) * (long)p + c) * p + q2] = (q2 == c) ? hbv : 0.0;
      }
    }
  }

  // ---- gradients out ----------------------------------------------------------------------------------------
  __syncthreads();
  if (g == 0 && c < p) {
    HB[c] = hbv;
    DB[c] = dbv;
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) FT[(4 * s + g) * TILE + c] = ACC[s * KF_WAVE + lane];
  __syncthreads();
  if (g == 0 && c_ok) {
    if (A.g_a0) A.g_a0[b * (long)m + c] = abp;
    if (A.g_c && !tv_C) put(A.g_c + b * (long)m + c, cb);
  }
  if (A.g_Z && !tv_Z)
    for (int e = lane; e < p * m; e += KF_WAVE) {
      const int q = e / m, k = e - q * m;
      put(A.g_Z + b * (long)p * m + e, ZB[q * TILE + k]);
    }
  if (A.g_H && !tv_H)
    for (int e = lane; e < p * p; e += KF_WAVE) {
      const int q1 = e / p, q2 = e - q1 * p;
      put(A.g_H + b * (long)p * p + e, (q1 == q2) ? HB[q1] : 0.0);
    }
  if (A.g_d && !tv_D && lane < p) put(A.g_d + b * (long)p + lane, DB[lane]);
  if (want_yy)
    for (int e = lane; e < p * p; e += KF_WAVE) put(A.g_yy + b * (long)p * p + e, GY[(e / p) * 8 + e % p]);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (!on[s]) continue;
    if (A.g_T && !tv_T) put(A.g_T + b * (long)mm + off[s], fma(2.0, TBm[s], ACC[(4 + s) * KF_WAVE + lane]));
    if (A.g_P0 && p > 1) A.g_P0[b * (long)mm + off[s]] = (TP && c_pP) ? PB[s] + c_pP[off[s]] : PB[s];
  }
  // R_b = Gs R (Q + Q') ;  Q_b = R' Gs R   (Gs in tile FT)
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  if (A.g_R)
    for (int e = lane; e < m * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double sacc = 0.0;
      for (int k = 0; k < r_; ++k) {
        double grk = 0.0;  // (Gs R)[i][k]
        for (int j = 0; j < m; ++j) grk = fma(FT[i * TILE + j], gR[j * r_ + k], grk);
        sacc = fma(grk, gQ[k * r_ + l] + gQ[l * r_ + k], sacc);
      }
      put(A.g_R + b * (long)m * r_ + e, sacc);
    }
  if (A.g_Q)
    for (int e = lane; e < r_ * r_; e += KF_WAVE) {
      const int k = e / r_, l = e - k * r_;
      double sacc = 0.0;
      for (int i = 0; i < m; ++i) {
        double grl = 0.0;
        for (int j = 0; j < m; ++j) grl = fma(FT[i * TILE + j], gR[j * r_ + l], grl);
        sacc = fma(gR[i * r_ + k], grl, sacc);
      }
      put(A.g_Q + b * (long)r_ * r_ + e, sacc);
    }
}

}  // namespace

int kf_launch_backward_mfma(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  if (a.m > TILE || a.p > PMAXQ || a.p < 1) {
    KF_SET_ERR(h, "backward(mfma): m=%d p=%d out of range (m <= %d, p <= %d)", a.m, a.p, TILE, PMAXQ);
    return KF_ERR_UNSUPPORTED;
  }
  const size_t lds = sizeof(double) * (size_t)bwd_mfma_lds_doubles(a.p);
  const int ks = (a.m + 3) / 4;
  const bool tv = (a.flags & KF_FLAG_TV_ANY) != 0;
  if (tv && ((a.flags & (KF_FLAG_TV_R | KF_FLAG_TV_Q)) || a.steps || a.accumulate)) {
    KF_SET_ERR(h, "backward(mfma): time-varying R / Q (and the steady-state hand-over) stay on the LDS adjoint");
    return KF_ERR_UNSUPPORTED;
  }
  const bool tp = a.g_filt_a || a.g_pred_a || a.g_filt_P || a.g_pred_P;
  if (tp && (tv || a.steps || a.accumulate || a.init_ab || a.init_Pb || a.g_yy)) {
    KF_SET_ERR(h, "backward(mfma): tape cotangents with time-varying matrices or a hand-over stay on the LDS adjoint");
    return KF_ERR_UNSUPPORTED;
  }
  const bool w4 = tp && ks <= 3 && a.B > 12L * (h->n_cu > 0 ? h->n_cu : 256);   // more draws than one round of three per SIMD
#define KF_BWD_MFMA_LAUNCH(KS_)                                                                        \
  if (tp && w4 && a.p == 1)                                                                            \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, true, false, true, (KS_ <= 3)>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a);  \
  else if (tp && w4)                                                                                   \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, false, false, true, (KS_ <= 3)>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a); \
  else if (tp && a.p == 1)                                                                             \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, true, false, true>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a);  \
  else if (tp)                                                                                         \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, false, false, true>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a); \
  else if (tv && a.p == 1)                                                                                  \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, true, true>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a);  \
  else if (tv)                                                                                         \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, false, true>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a); \
  else if (a.p == 1)                                                                                   \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, true>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a);  \
  else                                                                                                 \
    hipLaunchKernelGGL((kf_bwd_mfma_kernel<KS_, false>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a)
  switch (ks) {
    case 1: KF_BWD_MFMA_LAUNCH(1); break;
    case 2: KF_BWD_MFMA_LAUNCH(2); break;
    case 3: KF_BWD_MFMA_LAUNCH(3); break;
    default: KF_BWD_MFMA_LAUNCH(4); break;
  }
#undef KF_BWD_MFMA_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
