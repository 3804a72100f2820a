// kf_smoother_bf.hip -- smoothed moments from the FILTER's predicted tapes, without inverting the prediction covariance
// (one wavefront per draw on v_mfma_f64_4x4x4_4b_f64, scalar observation p == 1, k_states + 1 <= 16, static T, Z, H, d).
// Companion of kf_smoother_b4.hip (tile layout, MFMA conventions: kf_b4.h, kf_forward_b4.hip).
//
// What it replaces: KalmanSmoother.build_graph (pymc_statespace/filters/kalman_smoother.py:56-104) WHEN its tapes come from this
// library's filter on the same model -- the only way the reference itself ever calls it (core/statespace.py:176-197 hands the
// filter's outputs straight to the smoother).  The reference's recursion needs pinv(P_hat) at every step; at the north-star shape
// that inverse is what bounds kf_smoother_b4.hip (warm-started Newton-Schulz: 6 products + 1 - 4 iterations, and a launch as long as
// the chain of the draws whose filter never converges, profiles/r04/smooth_split_kernel_stats.txt).  With the filter's own
// quantities the same moments follow from the backward recursion of de Jong (1989) / Durbin & Koopman (2012, 4.4) -- what
// statsmodels, the oracle that pins this build's golden vectors, computes -- with no m x m inverse at all (p = 1: F is a scalar):
//     v_t = y_t - Z a_t - d      F_t = Z P_t Z' + H      K_t = P_t Z' / F_t      L_t = T (I - K_t Z)      (a_t, P_t: PREDICTED)
//     r_{t-1} = Z' v_t / F_t + L_t' r_t        N_{t-1} = Z' Z / F_t + L_t' N_t L_t        (r_{n-1} = 0, N_{n-1} = 0)
//     a_s[t] = a_t + P_t r_{t-1}               P_s[t] = P_t - P_t N_{t-1} P_t
// (a missing observation: L_t = T and no Z terms).  Algebraically the RTS moments (both are E / Var of alpha_t given all of y);
// measured against the reference's recursion restated in oracle/kalman_numpy.py: <= 5e-14 of a draw's largest entry on the BASELINE
// workloads (scripts/r05/bf_proto.py), and a numerically singular P_hat -- the reference's pinv case, a fallback chain of three
// kernels in kf_smoother_b4.hip -- is just another step here.
//
// Tiles are N4 x N4, N4 = 4 NB >= m + 1: indices 0 .. m-1 states, JC = N4 - 1 carries the VECTORS (T_aug = blockdiag(T, 1)).
// What is carried from step to step is  M_t = T_aug' N_aug,t T_aug  with N_aug = [[N, r], [r', .]], and the gain is folded into the
// static factors, so that the dependency chain of a step is two products and ONE pass through an LDS tile -- the forward kernel's
// shape (kf_forward_b4.hip), run backwards:
//     kappa = (K, -v/F)                 (Z P_t | Z a_t) is three MFMAs with Z as a one-row left factor; ONE LDS exchange gives kappa
//                                       by row index g (for G) and by column index q (for the rank-one step)
//     Pf_ext = P_ext - F kappa kappa'   the FILTERED moments in registers: state block P_f = (I - K Z) P_t, column / row JC a_f
//     Y = M_t Pf_ext (row JC left out)  S = Pf_ext - P_f Y        smooth_P[t] = P_f - P_f M_t P_f,  smooth_a[t]' = a_f' + row JC of Y
//     G = T_aug - K_aug (Z T)_aug       V = M_t G                 M_{t-1} = G' V + zeta zeta' / F,   zeta = ((Z T)', v)
// (P_t - P_t N_{t-1} P_t = P_f - P_f (T' N_t T) P_f because N_{t-1} = (I - K Z)' T' N_t T (I - K Z) + Z' Z / F and P_t Z' Z P_t / F =
// P_t - P_f.)  The left factor of M_t -- one pass of M through an LDS tile -- serves both Y and V; P_f's needs a second tile.  Per step
// 42 block MFMAs, 11 LDS writes, 17 LDS reads, no data-dependent branch: every draw costs the same.  A missing observation: kappa = 0,
// 1/F := 0 -- G = T_aug, Pf_ext = P_ext, nothing else changes.
#include "kf_b4.h"

namespace {

constexpr int BF_LB = 32;  // steps per observation block
// row_allsum (kf_tile.h) without the copies its update_dpp(x, x) form makes before every v_mov_dpp (all lanes are written: no old value)
template <int CTRL>
__device__ __forceinline__ double bf_dpp(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bf_row_allsum(double x) {
  asm volatile("" : "+v"(x));
  x += bf_dpp<0x128>(x);
  x += bf_dpp<0x124>(x);
  x += bf_dpp<0x122>(x);
  x += bf_dpp<0x121>(x);
  return x;
}

template <int NB>
struct BfLds {
  static constexpr int TILE_D = 4 * NB * TRS;
  static constexpr int IMG_D = NB <= 3 ? 128 : 240;  // m <= 11: 121 doubles; m <= 15: 225
  // TN (M), TP (P_f), UX (kappa: 16 | zeros 4), UZ (Z: 16 | zeros 4), UT (Z T: 16 | zeros 4), compact output image, its state row,
  // the observation block, one dump slot per lane (where lanes without an element write: no exec masks in the step)
  static constexpr int OFF_TP = TILE_D, OFF_UX = 2 * TILE_D, OFF_UZ = OFF_UX + 20, OFF_UT = OFF_UZ + 20, OFF_IMG = OFF_UT + 20,
                       OFF_IMA = OFF_IMG + IMG_D, OFF_YB = OFF_IMA + 16, OFF_DUMP = OFF_YB + 2 * BF_LB;
  static constexpr int TOTAL = OFF_DUMP + KF_WAVE;
};
static_assert(BfLds<3>::TOTAL <= 1280 && BfLds<4>::TOTAL <= 1280, "four wavefronts per SIMD need <= 10 KiB of LDS per wavefront");

template <int NB>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(NB <= 3 ? 4 : 3))) void kf_smooth_bf_kernel(
    const KfSmoothBfArgs A) {
  constexpr int N4 = 4 * NB, JC = N4 - 1;
  using L = BfLds<NB>;
  __shared__ __attribute__((aligned(16))) double smem[L::TOTAL];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15, q = lane & 3;
  const int fb = q * 4 + ((lane >> 2) & 3);
  const long b = blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m;
  double* TN = smem;
  double* TP = smem + L::OFF_TP;
  double* UX = smem + L::OFF_UX;
  double* UZ = smem + L::OFF_UZ;
  double* UT = smem + L::OFF_UT;
  double* IMG = smem + L::OFF_IMG;
  double* IMA = smem + L::OFF_IMA;
  double* YB = smem + L::OFF_YB;
  double* DUMP = smem + L::OFF_DUMP;

  const double* gT = A.Tm + b * (long)mm;
  const double* gZ = A.Z + b * (long)m;
  for (int e = lane; e < L::TOTAL; e += KF_WAVE) smem[e] = 0.0;
  wave_lds_fence();
  // Z and Z T by column index, in exchange layout (element j at 4 (j & 3) + (j >> 2)); entries >= m stay zero
  if (lane < m) {
    double zt = 0.0;
    for (int l = 0; l < m; ++l) zt = fma(gZ[l], gT[l * m + lane], zt);
    const int pos = 4 * (lane & 3) + (lane >> 2);
    UZ[pos] = gZ[lane];
    UT[pos] = zt;
  }
  wave_lds_fence();
  // ---- per-draw constants ---------------------------------------------------------------------------------------------------
  double Tt[NB][NB];  // left-factor blocks of T_aug': lane (g = k, any b, q = i) holds T_aug[4K+g][4I+q]
  double Tr[NB];      // row-block registers of T_aug
  auto load_T = [&]() {   // (again behind a steady phase: its constants take these registers meanwhile)
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g;
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const int ti = 4 * I + q, tk = 4 * K + g;
        Tt[I][K] = (ti < m && tk < m) ? gT[tk * m + ti] : ((ti == JC && tk == JC) ? 1.0 : 0.0);
      }
      Tr[I] = (rr < m && bq < m) ? gT[rr * m + bq] : ((rr == JC && bq == JC) ? 1.0 : 0.0);
    }
  };
  load_T();
  const double* src[NB];
  int stp[NB];
  bool on_tp[NB];
  unsigned img_off[NB];  // LDS byte offset of this lane's element in the compact output image (its dump slot if none)
  const double* gpa = A.pred_a + b * (long)(n + 1) * m;
  const double* gpP = A.pred_P + b * (long)(n + 1) * mm;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int rr = 4 * I + g;
    // tape image of step t: the state block P_t, a_t in column JC and in row JC; zero elsewhere
    on_tp[I] = true;
    if (rr < m && bq < m) { src[I] = gpP + rr * m + bq; stp[I] = mm; }
    else if (rr < m && bq == JC) { src[I] = gpa + rr; stp[I] = m; }
    else if (rr == JC && bq < m) { src[I] = gpa + bq; stp[I] = m; }
    else { src[I] = gpa; stp[I] = 0; on_tp[I] = false; }
    src[I] += (long)(n - 1) * stp[I];
    img_off[I] = (unsigned)(((rr < m && bq < m) ? L::OFF_IMG + rr * m + bq
                             : ((rr == JC && bq < m) ? L::OFF_IMA + bq : L::OFF_DUMP + lane)) * 8);
  }
  // Z by column (F's reduction), Z T by column, by row index g and by column index q (all from the exchange-layout copies)
  const double Zb = UZ[4 * (bq & 3) + (bq >> 2)];
  const double ZTb = UT[4 * (bq & 3) + (bq >> 2)];
  // Z T by row index g and by column index q, Z as a one-row left factor (lane (g = k, q = 0) holds Z[4K+g]): re-read from LDS in
  // every step (three reads off the dependency chain) rather than held in 18 registers -- the step would spill them
  const double* ut_g = UT + 4 * g;
  const double* ut_q = UT + 4 * q;
  const double* uz_l = (q == 0) ? UZ + 4 * g : UZ + 16;
  auto read4 = [&](const double* p, double (&o)[4]) {
    const double2_t a01 = *reinterpret_cast<const double2_t*>(p);
    double2_t a23 = {0.0, 0.0};
    if (NB > 2) a23 = *reinterpret_cast<const double2_t*>(p + 2);
    o[0] = a01.x; o[1] = a01.y; o[2] = a23.x; o[3] = a23.y;
  };
  const bool row_jc = g == 3;           // tile row JC: DPP row 3 of the last register
  const bool col_jc = bq == JC;
  const bool l_jc0 = lane == JC;        // entry JC of a vector held along DPP row 0
  const double H1 = A.H[b];
  const double d1 = A.d ? A.d[b] : 0.0;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // kalman_filter.py:335 quirk: y_hat = Z a - d
  // kappa's exchange: DPP row 0 writes element bq at 4 q + b; by row index every lane reads {kappa[g], kappa[4+g], ..}, by column
  // index the lanes of DPP row 0 read {kappa[q], kappa[4+q], ..} (the others zeros: the rank-one step contracts over k = 0 only)
  double* ux_w = (g == 0) ? UX + fb : DUMP + lane;
  const double* ux_g = UX + 4 * g;
  const double* ux_q = (g == 0) ? UX + 4 * q : UX + 16;

  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
  auto load_y_block = [&](int tb) {  // steps [tb, tb + BF_LB): {y -+ d (0 when missing), observed ? 1 : 0}
    const int cnt = (n - tb) < BF_LB ? (n - tb) : BF_LB;
    if (lane < cnt) {
      const double yv = ybase[(long)tb + lane];
      const bool ob = !(yv != yv);
      double2_t e = {ob ? yv + dsign * d1 : 0.0, ob ? 1.0 : 0.0};
      *reinterpret_cast<double2_t*>(YB + 2 * lane) = e;
    }
    wave_lds_fence();
  };
  double PN[NB];
  auto prefetch = [&]() {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      PN[I] = *src[I];
      src[I] -= stp[I];
    }
  };
  const int nb128 = mm >> 1;
  const bool st1 = lane < nb128, st1b = (mm & 1) && lane == nb128;
  const bool st2 = lane + KF_WAVE < nb128, st2b = (mm & 1) && lane + KF_WAVE == nb128;
  auto stage_image = [&](const double (&S)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) *reinterpret_cast<double*>(reinterpret_cast<char*>(smem) + img_off[I]) = S[I];
    wave_lds_fence();
  };
  auto emit_image = [&](int t) {
    double* dst = A.smooth_P + (b * (long)n + t) * mm;
    const double2_t v2 = *reinterpret_cast<const double2_t*>(IMG + 2 * lane);
    if (st1) *(reinterpret_cast<double2_t*>(dst) + lane) = v2;
    else if (st1b) dst[mm - 1] = v2.x;
    if (NB > 3 && mm > 2 * KF_WAVE) {  // (m >= 12)
      const double2_t w2 = *reinterpret_cast<const double2_t*>(IMG + 2 * (lane + KF_WAVE));
      if (st2) *(reinterpret_cast<double2_t*>(dst) + lane + KF_WAVE) = w2;
      else if (st2b) dst[mm - 1] = w2.x;
    }
    if (lane < m) A.smooth_a[(b * (long)n + t) * m + lane] = IMA[lane];
    wave_lds_fence();
  };
  auto store_moments = [&](int t, const double (&S)[NB]) {
    stage_image(S);
    emit_image(t);
  };
  // left factor X(I, K) of a tile in LDS: lane (g = k, any b, q = i) gets X[4I+q][4K+g], K = 0 .. 3
  auto left_factor = [&](const double* tile, double (&Lf)[NB][4]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const double* ls = tile + (4 * I + q) * TRS + 4 * g;
      const double2_t d01 = *reinterpret_cast<const double2_t*>(ls);
      double2_t d23 = {0.0, 0.0};
      if (NB > 2) d23 = *reinterpret_cast<const double2_t*>(ls + 2);
      Lf[I][0] = d01.x; Lf[I][1] = d01.y; Lf[I][2] = d23.x; Lf[I][3] = d23.y;
    }
  };

  double nl[NB][4];  // left factor of M_t (M_{n-1} = 0)
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int K = 0; K < 4; ++K) nl[I][K] = 0.0;
  prefetch();
  for (int t = n - 1; t >= 0; --t) {
    if (t == n - 1 || (t & (BF_LB - 1)) == BF_LB - 1) load_y_block(t & ~(BF_LB - 1));
    double Sf[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) Sf[I] = on_tp[I] ? PN[I] : 0.0;
    if (t > 0) prefetch();
    const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (t & (BF_LB - 1)));  // {y -+ d, observed}
    // ---- the filter's quantities of step t: (Z P_t | Z a_t) along DPP row 0, F, v, kappa -----------------------------------------
    double Zl[4], ZTg[4], ZTq[4];
    read4(uz_l, Zl);
    read4(ut_g, ZTg);
    read4(ut_q, ZTq);
    double zP = mfma4(Zl[0], Sf[0], 0.0);
#pragma unroll
    for (int K = 1; K < NB; ++K) zP = mfma4(Zl[K], Sf[K], zP);
    const double F = H1 + readlane_f64(bf_row_allsum(zP * Zb), 0);
    const bool upd = yo.y != 0.0 && F > 0.0;  // (F <= 0: the filter flagged the draw; UnivariateFilter skips F == 0, :468-471)
    const double rF = upd ? fast_rcp(F) : 0.0;
    const double v = yo.x - readlane_f64(zP, JC);
    const double kap = (l_jc0 ? -v : zP) * rF;  // (K, -v / F) along DPP row 0, zero in the other rows and on a missing step
    *ux_w = kap;
    wave_lds_fence();
    double kg[4], kq[4];
    read4(ux_g, kg);
    read4(ux_q, kq);
    wave_lds_fence();
    if (row_jc) kg[NB - 1] = 0.0;  // K_aug[JC] = 0 (kappa[JC] = -v / F belongs to the rank-one step only)
    // ---- Pf_ext = P_ext - F kappa kappa' ------------------------------------------------------------------------------------------
    double Pf[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) Pf[I] = mfma4_nega(F * kq[I], kap, Sf[I]);
    // ---- V = M_t G (G = T_aug - K_aug (Z T)_aug),  Y = M_t Pf_ext: both from the left factor of M_t ------------------------------
    double V[NB], Y[NB];
    {
      double Gr[NB], Pb[NB];
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        Gr[K] = fma(-kg[K], ZTb, Tr[K]);
        Pb[K] = (K == NB - 1 && row_jc) ? 0.0 : Pf[K];  // right factor: row JC (a_f') left out
      }
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        double acc = mfma4(nl[I][0], Gr[0], 0.0);
        double acy = mfma4(nl[I][0], Pb[0], 0.0);
#pragma unroll
        for (int K = 1; K < NB; ++K) {
          acc = mfma4(nl[I][K], Gr[K], acc);
          acy = mfma4(nl[I][K], Pb[K], acy);
        }
        V[I] = acc;
        Y[I] = acy;
      }
    }
    // ---- M_{t-1} = G' V + zeta zeta' / F -----------------------------------------------------------------------------------------
    double Mn[NB];
    {
      const double zc = (col_jc ? v : ZTb) * rF;
#pragma unroll
      for (int I = 0; I < NB; ++I) Mn[I] = ((I == NB - 1 && row_jc) ? v : ZTg[I]) * zc;
#pragma unroll
      for (int K = 0; K < NB; ++K)
#pragma unroll
        for (int I = 0; I < NB; ++I) Mn[I] = mfma4(fma(-kg[K], ZTq[I], Tt[I][K]), V[K], Mn[I]);
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) TN[(4 * I + g) * TRS + fb] = Mn[I];
    // Pf_ext as a left factor.  Its column JC (a_f) must not take part: row JC of Y is set aside (a_s needs it) and zeroed instead
    // -- one select, not one per register
#pragma unroll
    for (int I = 0; I < NB; ++I) TP[(4 * I + g) * TRS + fb] = Pf[I];
    const double yjc = Y[NB - 1];
    if (row_jc) Y[NB - 1] = 0.0;
    wave_lds_fence();
    left_factor(TN, nl);  // (the next step's)
    double S[NB];
    {
      double pl[NB][4];
      left_factor(TP, pl);
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        double acc = Pf[I];
#pragma unroll
        for (int K = 0; K < NB; ++K) acc = mfma4_nega(pl[I][K], Y[K], acc);
        S[I] = acc;
      }
    }
    if (row_jc) S[NB - 1] = Pf[NB - 1] + yjc;  // row JC: a_s' = a_f' + (P_f T' r_t)'
    wave_lds_fence();
    store_moments(t, S);
  }
}

}  // namespace

bool kf_smoother_bf_ok(int m, int p, unsigned flags) { return p == 1 && m >= 1 && m + 1 <= 16 && !(flags & KF_FLAG_TV_ANY); }

int kf_launch_smoother_bf(kf_handle* h, const KfSmoothBfArgs& a, hipStream_t stream) {
  if (!kf_smoother_bf_ok(a.m, 1, a.flags)) {
    KF_SET_ERR(h, "smoother(from filter): needs p == 1, m <= 15 and static matrices");
    return KF_ERR_UNSUPPORTED;
  }
  const int nb = (a.m + 1 + 3) / 4;
#define KF_SBF(NB_) hipLaunchKernelGGL((kf_smooth_bf_kernel<NB_>), dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a)
  switch (nb) {
    case 1: KF_SBF(1); break;
    case 2: KF_SBF(2); break;
    case 3: KF_SBF(3); break;
    default: KF_SBF(4); break;
  }
#undef KF_SBF
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
