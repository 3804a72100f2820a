// kf_backward_mfma_wg.hip -- reverse-mode gradient of the Kalman log-likelihood for
// 48 < k_states <= 64: one workgroup of four wavefronts per draw, wavefront J owning column block J
// of the covariance, of its adjoint and of the T-gradient accumulator (register X[s] =
// X[4 s + g][16 J + c], s < KSR = ceil(m / 4)).  Companion of kf_forward_mfma_wg.hip; consumes the
// predicted tapes of any forward kernel; p <= 8 scalar observations, static system matrices.
//
// Maths = kf_backward_mfma.hip's (derivation: oracle/kalman_adjoint_numpy.py).  Per step, reverse:
//   recompute the p scalar updates from the predicted tape (all-gather of M_q through LDS);
//     X1(:, J)  = Gm T(:, J)        A operand = Gm[I][ks] from the LDS image of the adjoint tile
//     Sb(:, J)  = T' X1(:, J)       A operand = T'[I][ks] from a static LDS image
//     Tb(:, J) += X1 P_f(:, J)      A operand = X1'[I][ks] from the LDS transpose tile of X1
//   then the rank-one adjoints of the scalar updates, last observation first; the two scalars
//   that need all columns (beta = M' ab, gamma = M' Sb M) are summed across the wavefronts
//   through LDS.  3 x 4 x KSR MFMAs per wavefront and step.
// Why: the HBM-workspace adjoint (kf_backward_big.hip) takes 0.53 - 0.74 s per 1000 steps at
// 4096 draws for m = 49 .. 64.
//
// Outputs as kf_filter_backward documents; g_P0 exact for p == 1, symmetric part for p > 1; g_H
// carries the diagonal (the caller asserted scalar observations).
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int PMAXQ = KF_MP_P_MAX;
constexpr int NBW = 4;
constexpr int NT = NBW * KF_WAVE;

template <int KSR>
struct WgBwdLayout {
  static constexpr int KSRP = KSR + (KSR & 1);
  static constexpr int LD = 4 * KSRP + 2;
  static constexpr int NCOL = 16 * NBW;
  static constexpr int IMG = NBW * KSR * KF_WAVE;
  static constexpr int OFF_PBL = IMG;                    // TBL: static A-operand image of T'
  static constexpr int OFF_TR = OFF_PBL + IMG;           // PBL: image of the adjoint tile
  static constexpr int OFF_EX = OFF_TR + 4 * KSR * LD;   // TR: transpose tile / final Gs image
  static constexpr int OFF_EXU = OFF_EX + 4 * KSRP;      // EX: ab' by row
  static constexpr int OFF_RED = OFF_EXU + 4 * KSRP;     // EXU: u by row
  static constexpr int OFF_HD = OFF_RED + 2 * NBW;       // RED: per-wavefront partial sums
  static constexpr int OFF_PQ = OFF_HD + 4 * PMAXQ;      // HD, DL, HB, DB
  static constexpr int PER_Q = 4 * NCOL + 2 * 4 * KSRP + 4;  // ZC, MC, AQ, ZB | ZR, MP | SC
  static int doubles(int p) { return OFF_PQ + p * PER_Q + 64 * p + 64; }
  static bool fin_fits(int m, int r) { return m * r <= IMG; }  // Gs R staged in PBL at the end
};

template <int KSR>
__global__ __launch_bounds__(NT) void kf_bwd_mfma_wg_kernel(const KfBwdArgs A) {
  using L = WgBwdLayout<KSR>;
  constexpr int KSRP = L::KSRP, LD = L::LD, NCOL = L::NCOL;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, J = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, p = A.p, mm = m * m;

  double* TBL = smem;
  double* PBL = smem + L::OFF_PBL;
  double* TR = smem + L::OFF_TR;
  double* EX = smem + L::OFF_EX;
  double* EXU = smem + L::OFF_EXU;
  double* RED = smem + L::OFF_RED;
  double* HD = smem + L::OFF_HD;
  double* DL = HD + PMAXQ;
  double* HB = DL + PMAXQ;
  double* DB = HB + PMAXQ;
  double* ZC = smem + L::OFF_PQ;     // Z_q by column index            p x NCOL
  double* MC = ZC + p * NCOL;        // M_q of the current step
  double* AQ = MC + p * NCOL;        // a^q (state before observation q)
  double* ZB = AQ + p * NCOL;        // accumulated g_Z
  double* ZR = ZB + p * NCOL;        // Z_q by row index, per lane group  p x 4 KSRP
  double* MP = ZR + p * 4 * KSRP;    // M_q by row index
  double* SC = MP + p * 4 * KSRP;    // F_q, v_q, 1 / F_q
  double* YB = SC + 4 * p;           // y block of 64 steps
  double* GL = YB + 64 * p;          // g_ll_obs block

  const int j = 16 * J + c;
  const bool cok = j < m;
  auto rok = [&](int s) { return 4 * s + g < m; };
  auto pos = [&](int jj) { return (jj & 3) * KSRP + (jj >> 2); };
  const int obase = g * m + j;  // offset of element (4 s + g, j): obase + 4 s m
  auto read_rows = [&](const double* src, double (&x)[KSRP]) {
#pragma unroll
    for (int s = 0; s < KSRP; s += 2) {
      const double2_t v2 = *reinterpret_cast<const double2_t*>(src + g * KSRP + s);
      x[s] = v2.x;
      x[s + 1] = v2.y;
    }
  };

  double TB[KSR];  // T tile, own column block: TB[s] = T[4 s + g][j]
  {
    const double* gT = A.Tm + b * (long)mm;
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      TB[s] = (cok && rok(s)) ? gT[obase + 4 * s * m] : 0.0;
      TBL[(J * KSR + s) * KF_WAVE + lane] = TB[s];
    }
    const double* gZ = A.Z + b * (long)p * m;
    for (int e = tid; e < p * NCOL; e += NT) {
      const int q = e / NCOL, k = e - q * NCOL;
      ZC[e] = (k < m) ? gZ[q * m + k] : 0.0;
      ZB[e] = 0.0;
      MC[e] = 0.0;
      AQ[e] = 0.0;
    }
    for (int e = tid; e < p * 4 * KSRP; e += NT) {
      const int q = e / (4 * KSRP), w = e - q * 4 * KSRP;
      const int gg = w / KSRP, s = w - gg * KSRP, k = 4 * s + gg;
      ZR[e] = (s < KSR && k < m) ? gZ[q * m + k] : 0.0;
      MP[e] = 0.0;
    }
    if (tid < 4 * KSRP) {
      EX[tid] = 0.0;
      EXU[tid] = 0.0;
    }
    if (tid < p) {
      HD[tid] = A.H[b * (long)p * p + tid * p + tid];
      DL[tid] = A.d ? A.d[b * (long)p + tid] : 0.0;
    }
  }
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const bool y_per_draw = (A.flags & KF_FLAG_Y_PER_DRAW) != 0;
  const double* ybase = A.y + (y_per_draw ? b * (long)n * p : 0L);
  const double* gpa = A.pred_a + b * (long)(n + 1) * m;
  const double* gpP = A.pred_P + b * (long)(n + 1) * mm;
  const double g_all = A.g_logp ? A.g_logp[b] : 0.0;
  const double* gll = A.g_ll_obs ? A.g_ll_obs + b * (long)n : nullptr;

  // adjoint state and accumulators (own column block)
  double PB[KSR], GS[KSR];
  double4_t TBm[NBW];  // X1 P_f + ab' a_f' / 2 (g_T = twice this)
#pragma unroll
  for (int s = 0; s < KSR; ++s) PB[s] = GS[s] = 0.0;
#pragma unroll
  for (int I = 0; I < NBW; ++I) TBm[I] = double4_t{0.0, 0.0, 0.0, 0.0};
  double abp = 0.0, cb = 0.0, hbv = 0.0, dbv = 0.0;

  auto stage_block = [&](int t) {  // y and g_ll_obs of the 64-step block that holds t
    const int t0 = t & ~63;
    const int cnt = (n - t0 < 64 ? n - t0 : 64);
    __syncthreads();
    for (int e = tid; e < cnt * p; e += NT) YB[e] = ybase[(long)t0 * p + e];
    if (tid < cnt) GL[tid] = gll ? gll[t0 + tid] : 0.0;
    __syncthreads();
  };
  stage_block(n - 1);

  for (int t = n - 1; t >= 0; --t) {
    double P[KSR], ar[KSRP];
    {
      const double* rp = gpP + (long)t * mm;
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        P[s] = (cok && rok(s)) ? rp[obase + 4 * s * m] : 0.0;
        ar[s] = rok(s) ? gpa[(long)t * m + 4 * s + g] : 0.0;
      }
      if (KSRP > KSR) ar[KSRP - 1] = 0.0;
    }
    double ac = cok ? gpa[(long)t * m + j] : 0.0;
    if ((t & 63) == 63 && t != n - 1) stage_block(t);
    const double* yrow = YB + (t & 63) * p;
    const double gt = g_all + GL[t & 63];

    // ---- forward recompute of the p scalar updates (kf_forward_mfma_wg.hip) ---------------------------
#pragma unroll 1
    for (int q = 0; q < p; ++q) {
      const double yq = yrow[q];
      const bool obs = !(yq != yq);
      double zr[KSRP];
      read_rows(ZR + q * 4 * KSRP, zr);
      double pc = P[0] * zr[0];
#pragma unroll
      for (int s = 1; s < KSR; ++s) pc = fma(P[s], zr[s], pc);
      const double Mc = col_allsum(pc);
      if (g == 0) {
        if (j < 4 * KSR) MP[q * 4 * KSRP + pos(j)] = Mc;
        MC[q * NCOL + j] = Mc;
        AQ[q * NCOL + j] = ac;
      }
      __syncthreads();
      double Mr[KSRP];
      read_rows(MP + q * 4 * KSRP, Mr);
      double fz = zr[0] * Mr[0], az = zr[0] * ar[0];
#pragma unroll
      for (int s = 1; s < KSR; ++s) {
        fz = fma(zr[s], Mr[s], fz);
        az = fma(zr[s], ar[s], az);
      }
      const double F = HD[q] + col_allsum(fz);
      const double zs = col_allsum(az);
      const bool degenerate = !obs || (univariate && F == 0.0);
      const double invF = degenerate ? 0.0 : fast_rcp(F);
      const double v = degenerate ? 0.0 : (yq + dsign * DL[q]) - zs;
      if (tid == 0) {
        SC[4 * q] = F;
        SC[4 * q + 1] = v;
        SC[4 * q + 2] = invF;
      }
      const double vf = v * invF;
      ac = fma(Mc, vf, ac);
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        ar[s] = fma(Mr[s], vf, ar[s]);
        P[s] = fma(-(Mr[s] * Mc), invF, P[s]);
      }
    }
    // P = P_f (own columns), ac = a_f (own columns)

    // ---- adjoint of predict ---------------------------------------------------------------------------
    if (g == 0 && j < 4 * KSR) EX[pos(j)] = abp;
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      PBL[(J * KSR + s) * KF_WAVE + lane] = PB[s];
      GS[s] += PB[s];  // adjoint of R Q R'
    }
    __syncthreads();
    double abr[KSRP];
    read_rows(EX, abr);
    cb += abp;
    double part = TB[0] * abr[0];
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      TBm[s >> 2][s & 3] = fma(0.5 * abr[s], ac, TBm[s >> 2][s & 3]);  // ab' a_f' / 2
      if (s > 0) part = fma(TB[s], abr[s], part);
    }
    double abf = col_allsum(part);  // T' ab', own columns
    double4_t X1[NBW];
#pragma unroll
    for (int I = 0; I < NBW; ++I) X1[I] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR)
          X1[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(PBL[(I * KSR + ks) * KF_WAVE + lane], TB[ks], X1[I], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KSR; ++s)
      if (j < 4 * KSR) TR[(4 * s + g) * LD + pos(j)] = X1[s >> 2][s & 3];
    double4_t SBv[NBW];
#pragma unroll
    for (int I = 0; I < NBW; ++I) SBv[I] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR)
          SBv[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(TBL[(I * KSR + ks) * KF_WAVE + lane], X1[ks >> 2][ks & 3], SBv[I], 0, 0, 0);
    __syncthreads();
    // Tb(:, J) += X1 P_f(:, J) : A operand X1'[I][ks] (lane (g, c): X1[16 I + c][4 ks + g]) from the transpose tile
#pragma unroll
    for (int I = 0; I < NBW; ++I) {
      if (4 * I >= KSR) continue;
      const int i = 16 * I + c;
      double x1t[KSRP];
      read_rows(TR + (i < 4 * KSR ? i : 4 * KSR - 1) * LD, x1t);
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks)
        TBm[I] = __builtin_amdgcn_mfma_f64_16x16x4f64((i < 4 * KSR) ? x1t[ks] : 0.0, P[ks], TBm[I], 0, 0, 0);
    }
    double SB[KSR];
#pragma unroll
    for (int s = 0; s < KSR; ++s) SB[s] = SBv[s >> 2][s & 3];

    // ---- adjoint of the scalar updates, last observation first ---------------------------------------
#pragma unroll 1
    for (int q = p - 1; q >= 0; --q) {
      double Mr[KSRP], zr[KSRP];
      read_rows(MP + q * 4 * KSRP, Mr);
      read_rows(ZR + q * 4 * KSRP, zr);
      const double v = SC[4 * q + 1], invF = SC[4 * q + 2];
      const double Mc = MC[q * NCOL + j], zc = ZC[q * NCOL + j];
      // P^q = P^{q+1} + M M'/F ;  s = Sb M
      double sp = 0.0;
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        P[s] = fma(Mr[s] * Mc, invF, P[s]);
        sp = fma(SB[s], Mr[s], sp);
      }
      const double sc = col_allsum(sp);
      // beta = M' ab_f, gamma = M' Sb M: partial sums over the own columns, then over the wavefronts
      const double pb = row_allsum(Mc * abf), pg = row_allsum(Mc * sc);
      if (lane == 0) {
        RED[2 * J] = pb;
        RED[2 * J + 1] = pg;
      }
      __syncthreads();
      const double beta = (RED[0] + RED[2]) + (RED[4] + RED[6]);
      const double gamma = (RED[1] + RED[3]) + (RED[5] + RED[7]);
      const double Fb = (gamma - beta * v) * invF * invF - 0.5 * gt * (invF - v * v * invF * invF);
      const double vb = (beta - gt * v) * invF;
      const double uc = (abf * v - 2.0 * sc) * invF;  // m1b + m2b
      if (g == 0 && j < 4 * KSR) EXU[pos(j)] = uc;
      __syncthreads();
      double ur[KSRP];
      read_rows(EXU, ur);
      double pp = P[0] * ur[0];
#pragma unroll
      for (int s = 1; s < KSR; ++s) pp = fma(P[s], ur[s], pp);
      const double pdot = col_allsum(pp);  // P^q (m1b + m2b)
      if (g == 0) ZB[q * NCOL + j] += fma(-AQ[q * NCOL + j], vb, fma(2.0 * Fb, Mc, pdot));
      if (p == 1 && t == 0 && A.g_P0) {
        // exact (not symmetrised) entrywise gradient: Sb + (Fb z + m1b) z' - z s'/F
        __syncthreads();
        if (g == 0 && j < 4 * KSR) EXU[pos(j)] = (abf * v - sc) * invF;  // m1b
        __syncthreads();
        double m1r[KSRP];
        read_rows(EXU, m1r);
#pragma unroll
        for (int s = 0; s < KSR; ++s) {
          double gp = SB[s];
          gp = fma(fma(Fb, zr[s], m1r[s]), zc, gp);
          gp = fma(-zr[s] * invF, sc, gp);
          if (cok && rok(s)) A.g_P0[b * (long)mm + obase + 4 * s * m] = gp;
        }
      }
      // symmetrised adjoint of P^q ; adjoint of a^q
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        double sb = SB[s];
        sb = fma(Fb * zr[s], zc, sb);
        sb = fma(0.5 * ur[s], zc, sb);
        sb = fma(0.5 * zr[s], uc, sb);
        SB[s] = sb;
      }
      abf = fma(-zc, vb, abf);
      if (c == q) {
        hbv += Fb;
        dbv += dsign * vb;
      }
    }
#pragma unroll
    for (int s = 0; s < KSR; ++s) PB[s] = SB[s];
    abp = abf;
  }

  // ---- gradients out ----------------------------------------------------------------------------------------
  __syncthreads();
  if (J == 0 && g == 0 && c < p) {
    HB[c] = hbv;
    DB[c] = dbv;
  }
  // Gs as a plain row-major image (row stride LD >= m) in TR
#pragma unroll
  for (int s = 0; s < KSR; ++s)
    if (cok && rok(s)) TR[(4 * s + g) * LD + j] = GS[s];
  if (g == 0 && cok) {
    if (A.g_a0) A.g_a0[b * (long)m + j] = abp;
    if (A.g_c) A.g_c[b * (long)m + j] = cb;
  }
#pragma unroll
  for (int s = 0; s < KSR; ++s) {
    if (!(cok && rok(s))) continue;
    const long o = b * (long)mm + obase + 4 * s * m;
    if (A.g_T) A.g_T[o] = 2.0 * TBm[s >> 2][s & 3];
    if (A.g_P0 && p > 1) A.g_P0[o] = PB[s];
  }
  __syncthreads();
  if (A.g_Z)
    for (int e = tid; e < p * m; e += NT) {
      const int q = e / m, k = e - q * m;
      A.g_Z[b * (long)p * m + e] = ZB[q * NCOL + k];
    }
  if (A.g_H)
    for (int e = tid; e < p * p; e += NT) {
      const int q1 = e / p, q2 = e - q1 * p;
      A.g_H[b * (long)p * p + e] = (q1 == q2) ? HB[q1] : 0.0;
    }
  if (A.g_d && tid < p) A.g_d[b * (long)p + tid] = DB[tid];
  // R_b = Gs R (Q + Q') ;  Q_b = R' Gs R   (Gs in TR, Gs R staged in PBL)
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  double* GRS = PBL;
  if (A.g_R || A.g_Q) {
    for (int e = tid; e < m * r_; e += NT) {
      const int i = e / r_, k = e - i * r_;
      double grk = 0.0;
      for (int jj = 0; jj < m; ++jj) grk = fma(TR[i * LD + jj], gR[jj * r_ + k], grk);
      GRS[e] = grk;
    }
    __syncthreads();
  }
  if (A.g_R)
    for (int e = tid; e < m * r_; e += NT) {
      const int i = e / r_, l = e - i * r_;
      double sacc = 0.0;
      for (int k = 0; k < r_; ++k) sacc = fma(GRS[i * r_ + k], gQ[k * r_ + l] + gQ[l * r_ + k], sacc);
      A.g_R[b * (long)m * r_ + e] = sacc;
    }
  if (A.g_Q)
    for (int e = tid; e < r_ * r_; e += NT) {
      const int k = e / r_, l = e - k * r_;
      double sacc = 0.0;
      for (int i = 0; i < m; ++i) sacc = fma(gR[i * r_ + k], GRS[i * r_ + l], sacc);
      A.g_Q[b * (long)r_ * r_ + e] = sacc;
    }
}

template <int KSR>
int launch_bwd_wg(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)WgBwdLayout<KSR>::doubles(a.p);
  static KfAttrOnce attr_set;  // the attribute is per device
  if (attr_set.need(h->device)) {
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&kf_bwd_mfma_wg_kernel<KSR>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, KF_LDS_BYTES_MAX));
    attr_set.mark(h->device);
  }
  hipLaunchKernelGGL((kf_bwd_mfma_wg_kernel<KSR>), dim3((unsigned)a.B), dim3(NT), lds, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

bool kf_backward_mfma_wg_ok(int m, int p, int r) {
  if (!(m > KF_NB_M_MAX && m <= KF_WG_M_MAX && p >= 1 && p <= PMAXQ)) return false;
  if (sizeof(double) * (size_t)WgBwdLayout<16>::doubles(p) > KF_LDS_BYTES_MAX) return false;
  return m * r <= NBW * ((m + 3) / 4) * KF_WAVE;
}

int kf_launch_backward_mfma_wg(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  if (!kf_backward_mfma_wg_ok(a.m, a.p, a.r)) {
    KF_SET_ERR(h, "backward(mfma_wg): m=%d p=%d r=%d out of range (%d < m <= %d, p <= %d)", a.m, a.p, a.r,
               KF_NB_M_MAX, KF_WG_M_MAX, PMAXQ);
    return KF_ERR_UNSUPPORTED;
  }
  switch ((a.m + 3) / 4) {
    case 13: return launch_bwd_wg<13>(h, a, stream);
    case 14: return launch_bwd_wg<14>(h, a, stream);
    case 15: return launch_bwd_wg<15>(h, a, stream);
    default: return launch_bwd_wg<16>(h, a, stream);
  }
}
