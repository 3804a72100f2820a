// kf_backward_mfma_nb.hip -- reverse-mode gradient of the Kalman log-likelihood for
// 16 < k_states <= 48: covariance and adjoint as NB x NB blocks of 16 x 16 in MFMA accumulator
// layout, one wavefront per draw, p <= 8 scalar observations, static system matrices.  Companion of
// kf_forward_mfma_nb.hip (same register layout: X[J][s] = X[4 s + g][16 J + c]); consumes the
// predicted tapes of any forward kernel.
//
// Maths = kf_backward_mfma.hip's (derivation: oracle/kalman_adjoint_numpy.py).  Per step, reverse:
//   recompute the p scalar updates from the predicted tape (M_q, F_q, v_q, a^q kept in LDS), then
//     X1  = Gm T            X1(I, J) += mfma(A = Gm[I][ks] (symmetric), B = T[J][ks])
//     Sb  = T' X1           Sb(I, J) += mfma(A = T[I][ks],  B = X1[J][ks])      blocks I <= J, mirrored
//     Tb += X1 P_f          Tb(I, J) += mfma(A = X1'[I][ks] (LDS transpose), B = P_f[J][ks])
//   (T[J][s] = T[4 s + g][16 J + c] serves as B operand "T" and as A operand "T'"), then the
//   rank-one adjoints of the scalar updates, last observation first, rebuilding
//   P^q = P^{q+1} + M_q M_q' / F_q.
// Why: before this kernel the range was served by the generic LDS-resident adjoint: 144 ms (m = 20)
// to 2.6 s (m = 40) per 1000 steps at 4096 draws.
//
// Outputs as kf_filter_backward documents; g_P0 exact for p == 1, symmetric part for p > 1; g_H
// carries the diagonal (the caller asserted scalar observations).
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int PMAXQ = KF_MP_P_MAX;

// Gs accumulator slots: only the blocks on and above the block diagonal (Gs is symmetric)
__host__ __device__ constexpr int nb_slots_before(int J, int KSR) {
  int n = 0;
  for (int j = 0; j < J; ++j) n += (4 * (j + 1) < KSR) ? 4 * (j + 1) : KSR;
  return n;
}

template <int NB, int KSR>
struct NbBwdLayout {
  static constexpr int KSRP = KSR + (KSR & 1);
  static constexpr int LD = 4 * KSRP + 2;
  static constexpr int NCOL = 16 * NB;
  static constexpr int NSLOT = nb_slots_before(NB, KSR);
  static constexpr int OFF_ACC = 4 * KSR * LD;            // FT: transpose tile / final Gs image
  static constexpr int OFF_EX = OFF_ACC + NSLOT * KF_WAVE;
  static constexpr int OFF_HD = OFF_EX + 4 * KSRP;
  static constexpr int OFF_PQ = OFF_HD + 4 * PMAXQ;       // HD, DL, HB, DB
  static constexpr int PER_Q = 4 * NCOL + 2 * 4 * KSRP + 4;  // ZC, MC, AQ, ZB | ZR, MP | SC
  static int doubles(int p, int m, int r) {
    const int loop = OFF_PQ + p * PER_Q + 64 * p + 64, fin = OFF_ACC + m * r;  // Gs R aliases everything behind FT
    return loop > fin ? loop : fin;
  }
};

// occupancy: two wavefronts per SIMD (256 registers) only while that does not spill -- measured
// (4096 draws x 1000 steps): m = 20: 12.7 ms at two, 17.8 ms at one; m = 28: 22.8 / 21.8; m = 32
// (63 spilled registers at two): 29.7 / 23.0 ms
template <int NB, int KSR>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu((NB == 2 && KSR <= 6) ? 2 : 1))) void kf_bwd_mfma_nb_kernel(
    const KfBwdArgs A) {
  using L = NbBwdLayout<NB, KSR>;
  constexpr int KSRP = L::KSRP, LD = L::LD, NCOL = L::NCOL;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, p = A.p, mm = m * m;

  double* FT = smem;
  double* ACC = smem + L::OFF_ACC;   // Gs = sum of the predicted-covariance adjoints, slot (J, s) of lane l at [(J KSR + s) 64 + l]
  double* EX = smem + L::OFF_EX;
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
  double* GRS = ACC;                 // Gs R at the end (m x r): the loop buffers are dead by then
  // Gs slot of (J, s), defined for s >> 2 <= J
  auto slot = [&](int J, int s) { return (nb_slots_before(J, KSR) + s) * KF_WAVE + lane; };

  bool cok[NB];
#pragma unroll
  for (int J = 0; J < NB; ++J) cok[J] = 16 * J + c < m;
  auto rok = [&](int s) { return 4 * s + g < m; };
  auto pos = [&](int j) { return (j & 3) * KSRP + (j >> 2); };
  const int obase = g * m + c;  // tape / gradient offset of element (4 s + g, 16 J + c): obase + 4 s m + 16 J

  double TB[NB][KSR];  // T tile: TB[J][s] = T[4 s + g][16 J + c]
  {
    const double* gT = A.Tm + b * (long)mm;
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        TB[J][s] = (cok[J] && rok(s)) ? gT[obase + 4 * s * m + 16 * J] : 0.0;
        if ((s >> 2) <= J) ACC[slot(J, s)] = 0.0;
      }
    const double* gZ = A.Z + b * (long)p * m;
    for (int e = lane; e < p * NCOL; e += KF_WAVE) {
      const int q = e / NCOL, k = e - q * NCOL;
      ZC[e] = (k < m) ? gZ[q * m + k] : 0.0;
      ZB[e] = 0.0;
    }
    for (int e = lane; e < p * 4 * KSRP; e += KF_WAVE) {
      const int q = e / (4 * KSRP), w = e - q * 4 * KSRP;
      const int gg = w / KSRP, s = w - gg * KSRP, k = 4 * s + gg;
      ZR[e] = (s < KSR && k < m) ? gZ[q * m + k] : 0.0;
      MP[e] = 0.0;
    }
    if (lane < 4 * KSRP) EX[lane] = 0.0;
    if (lane < p) {
      HD[lane] = A.H[b * (long)p * p + lane * p + lane];
      DL[lane] = A.d ? A.d[b * (long)p + lane] : 0.0;
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

  // adjoint state and accumulators
  double PB[NB][KSR];
  double4_t TBm[NB][NB];  // X1 P_f + ab' a_f' / 2 (g_T = twice this)
#pragma unroll
  for (int J = 0; J < NB; ++J) {
#pragma unroll
    for (int s = 0; s < KSR; ++s) PB[J][s] = 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I) TBm[I][J] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
  double abp[NB], cb[NB], hbv = 0.0, dbv = 0.0;
#pragma unroll
  for (int J = 0; J < NB; ++J) abp[J] = cb[J] = 0.0;

  auto stage_block = [&](int t) {  // y and g_ll_obs of the 64-step block that holds t
    const int t0 = t & ~63;
    const int cnt = (n - t0 < 64 ? n - t0 : 64);
    wave_lds_fence();
    for (int e = lane; e < cnt * p; e += KF_WAVE) YB[e] = ybase[(long)t0 * p + e];
    if (lane < cnt) GL[lane] = gll ? gll[t0 + lane] : 0.0;
    wave_lds_fence();
  };
  // row-indexed view (KSR values per lane) of a vector exchanged through LDS in `pos` order
  auto read_rows = [&](const double* src, double (&x)[KSRP]) {
#pragma unroll
    for (int s = 0; s < KSRP; s += 2) {
      const double2_t v2 = *reinterpret_cast<const double2_t*>(src + g * KSRP + s);
      x[s] = v2.x;
      x[s + 1] = v2.y;
    }
  };
  auto write_cols = [&](double* dst, const double (&x)[NB]) {
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
      if (g == 0 && j < 4 * KSR) dst[pos(j)] = x[J];
    }
  };
  stage_block(n - 1);

  for (int t = n - 1; t >= 0; --t) {
    double P[NB][KSR], ac[NB];
    {
      const double* rp = gpP + (long)t * mm;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
#pragma unroll
        for (int s = 0; s < KSR; ++s) P[J][s] = (cok[J] && rok(s)) ? rp[obase + 4 * s * m + 16 * J] : 0.0;
        ac[J] = cok[J] ? gpa[(long)t * m + 16 * J + c] : 0.0;
      }
    }
    if ((t & 63) == 63 && t != n - 1) stage_block(t);
    const double* yrow = YB + (t & 63) * p;
    const double gt = g_all + GL[t & 63];

    // ---- forward recompute of the p scalar updates (kf_forward_mfma_nb.hip) ---------------------------
#pragma unroll 1
    for (int q = 0; q < p; ++q) {
      const double yq = yrow[q];
      const bool obs = !(yq != yq);
      double zr[KSRP];
      read_rows(ZR + q * 4 * KSRP, zr);
      double Mc[NB], fz = 0.0, az = 0.0;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        double pc = P[J][0] * zr[0];
#pragma unroll
        for (int s = 1; s < KSR; ++s) pc = fma(P[J][s], zr[s], pc);
        Mc[J] = col_allsum(pc);
        const double zc = ZC[q * NCOL + 16 * J + c];
        fz = (J == 0) ? zc * Mc[J] : fma(zc, Mc[J], fz);
        az = (J == 0) ? zc * ac[J] : fma(zc, ac[J], az);
        if (g == 0) {
          MC[q * NCOL + 16 * J + c] = Mc[J];
          AQ[q * NCOL + 16 * J + c] = ac[J];
        }
      }
      write_cols(MP + q * 4 * KSRP, Mc);
      const double F = HD[q] + row_allsum(fz);
      const double zs = row_allsum(az);
      const bool degenerate = !obs || (univariate && F == 0.0);
      const double invF = degenerate ? 0.0 : fast_rcp(F);
      const double v = degenerate ? 0.0 : (yq + dsign * DL[q]) - zs;
      if (lane == 0) {
        SC[4 * q] = F;
        SC[4 * q + 1] = v;
        SC[4 * q + 2] = invF;
      }
      wave_lds_fence();
      double Mr[KSRP];
      read_rows(MP + q * 4 * KSRP, Mr);
#pragma unroll
      for (int J = 0; J < NB; ++J) {
#pragma unroll
        for (int s = 0; s < KSR; ++s) P[J][s] = fma(-(Mr[s] * Mc[J]), invF, P[J][s]);
        ac[J] = fma(Mc[J] * invF, v, ac[J]);
      }
    }
    // P = P_f, ac = a_f

    // ---- adjoint of predict ---------------------------------------------------------------------------
    write_cols(EX, abp);
    wave_lds_fence();
    double abr[KSRP];
    read_rows(EX, abr);
    double abf[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      cb[J] += abp[J];
      double part = TB[J][0] * abr[0];
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        TBm[s >> 2][J][s & 3] = fma(0.5 * abr[s], ac[J], TBm[s >> 2][J][s & 3]);  // ab' a_f' / 2
        if ((s >> 2) <= J) ACC[slot(J, s)] += PB[J][s];                             // adjoint of R Q R'
        if (s > 0) part = fma(TB[J][s], abr[s], part);
      }
      abf[J] = col_allsum(part);  // T' ab'
    }
    double4_t X1[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) X1[I][J] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J)
          X1[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(PB[I][ks], TB[J][ks], X1[I][J], 0, 0, 0);
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
#pragma unroll
      for (int s = 0; s < KSR; ++s)
        if (j < 4 * KSR) FT[(4 * s + g) * LD + pos(j)] = X1[s >> 2][J][s & 3];
    }
    double4_t SBv[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J) SBv[I][J] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J)
          SBv[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(TB[I][ks], X1[ks >> 2][J][ks & 3], SBv[I][J], 0, 0, 0);
    wave_lds_fence();
    // Tb += X1 P_f : A operand X1'[I][ks] (lane (g, c): X1[16 I + c][4 ks + g]) from the transpose tile
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int i = 16 * I + c;
      const double* row = FT + (i < 4 * KSR ? i : 4 * KSR - 1) * LD + g * KSRP;
      double x1t[KSRP];
#pragma unroll
      for (int s = 0; s < KSRP; s += 2) {
        const double2_t v2 = *reinterpret_cast<const double2_t*>(row + s);
        x1t[s] = (i < 4 * KSR) ? v2.x : 0.0;
        x1t[s + 1] = (i < 4 * KSR) ? v2.y : 0.0;
      }
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
        for (int J = 0; J < NB; ++J)
          TBm[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1t[ks], P[J][ks], TBm[I][J], 0, 0, 0);
    }
    wave_lds_fence();
    // Sb = T' Gm T is symmetric up to rounding: blocks below the diagonal are the mirror images
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
#pragma unroll
      for (int s = 0; s < KSR; ++s)
        if ((s >> 2) <= J && j < 4 * KSR) FT[(4 * s + g) * LD + pos(j)] = SBv[s >> 2][J][s & 3];
    }
    wave_lds_fence();
    double SB[NB][KSR];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
      const double* row = FT + (j < 4 * KSR ? j : 4 * KSR - 1) * LD + g * KSRP;
#pragma unroll
      for (int s = 0; s < 4 * J && s < KSR; ++s) SB[J][s] = SBv[s >> 2][J][s & 3];
#pragma unroll
      for (int s = 4 * J; s < KSRP; s += 2) {
        const double2_t t2 = *reinterpret_cast<const double2_t*>(row + s);
        const bool dg0 = (s >> 2) == J, dg1 = ((s + 1) >> 2) == J;
        const double v0 = dg0 ? 0.5 * (SBv[dg0 ? J : 0][J][s & 3] + t2.x) : t2.x;
        SB[J][s] = cok[J] ? v0 : 0.0;
        if (s + 1 < KSR) {
          const double v1 = dg1 ? 0.5 * (SBv[dg1 ? J : 0][J][(s + 1) & 3] + t2.y) : t2.y;
          SB[J][s + 1 < KSR ? s + 1 : 0] = cok[J] ? v1 : 0.0;
        }
      }
    }
    wave_lds_fence();

    // ---- adjoint of the scalar updates, last observation first ---------------------------------------
#pragma unroll 1
    for (int q = p - 1; q >= 0; --q) {
      double Mr[KSRP], zr[KSRP], Mc[NB], zc[NB];
      read_rows(MP + q * 4 * KSRP, Mr);
      read_rows(ZR + q * 4 * KSRP, zr);
      const double v = SC[4 * q + 1], invF = SC[4 * q + 2];
      // P^q = P^{q+1} + M M'/F ;  s = Sb M
      double sc[NB], bz = 0.0, gz = 0.0;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        Mc[J] = MC[q * NCOL + 16 * J + c];
        zc[J] = ZC[q * NCOL + 16 * J + c];
        double sp = 0.0;
#pragma unroll
        for (int s = 0; s < KSR; ++s) {
          P[J][s] = fma(Mr[s] * Mc[J], invF, P[J][s]);
          sp = fma(SB[J][s], Mr[s], sp);
        }
        sc[J] = col_allsum(sp);
        bz = (J == 0) ? Mc[J] * abf[J] : fma(Mc[J], abf[J], bz);
        gz = (J == 0) ? Mc[J] * sc[J] : fma(Mc[J], sc[J], gz);
      }
      const double beta = row_allsum(bz);
      const double gamma = row_allsum(gz);
      const double Fb = (gamma - beta * v) * invF * invF - 0.5 * gt * (invF - v * v * invF * invF);
      const double vb = (beta - gt * v) * invF;
      double uc[NB];
#pragma unroll
      for (int J = 0; J < NB; ++J) uc[J] = (abf[J] * v - 2.0 * sc[J]) * invF;  // m1b + m2b
      wave_lds_fence();
      write_cols(EX, uc);
      wave_lds_fence();
      double ur[KSRP];
      read_rows(EX, ur);
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        double pp = P[J][0] * ur[0];
#pragma unroll
        for (int s = 1; s < KSR; ++s) pp = fma(P[J][s], ur[s], pp);
        const double pdot = col_allsum(pp);  // P^q (m1b + m2b)
        if (g == 0)
          ZB[q * NCOL + 16 * J + c] += fma(-AQ[q * NCOL + 16 * J + c], vb, fma(2.0 * Fb, Mc[J], pdot));
      }
      if (p == 1 && t == 0 && A.g_P0) {
        // exact (not symmetrised) entrywise gradient: Sb + (Fb z + m1b) z' - z s'/F
        double m1c[NB], m1r[KSRP];
#pragma unroll
        for (int J = 0; J < NB; ++J) m1c[J] = (abf[J] * v - sc[J]) * invF;  // m1b
        wave_lds_fence();
        write_cols(EX, m1c);
        wave_lds_fence();
        read_rows(EX, m1r);
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
          for (int s = 0; s < KSR; ++s) {
            double gp = SB[J][s];
            gp = fma(fma(Fb, zr[s], m1r[s]), zc[J], gp);
            gp = fma(-zr[s] * invF, sc[J], gp);
            if (cok[J] && rok(s)) A.g_P0[b * (long)mm + obase + 4 * s * m + 16 * J] = gp;
          }
      }
      // symmetrised adjoint of P^q ; adjoint of a^q
#pragma unroll
      for (int J = 0; J < NB; ++J) {
#pragma unroll
        for (int s = 0; s < KSR; ++s) {
          double sb = SB[J][s];
          sb = fma(Fb * zr[s], zc[J], sb);
          sb = fma(0.5 * ur[s], zc[J], sb);
          sb = fma(0.5 * zr[s], uc[J], sb);
          SB[J][s] = sb;
        }
        abf[J] = fma(-zc[J], vb, abf[J]);
      }
      if (c == q) {
        hbv += Fb;
        dbv += dsign * vb;
      }
    }
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
      for (int s = 0; s < KSR; ++s) PB[J][s] = SB[J][s];
      abp[J] = abf[J];
    }
  }

  // ---- gradients out ----------------------------------------------------------------------------------------
  wave_lds_fence();
  if (g == 0 && c < p) {
    HB[c] = hbv;
    DB[c] = dbv;
  }
  // Gs as a plain row-major image (row stride LD >= m) in FT, mirrored below the block diagonal
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int s = 0; s < KSR; ++s)
      if ((s >> 2) <= J && cok[J] && rok(s)) {
        const double gs = ACC[slot(J, s)];
        FT[(4 * s + g) * LD + 16 * J + c] = gs;
        if ((s >> 2) < J) FT[(16 * J + c) * LD + 4 * s + g] = gs;
      }
  wave_lds_fence();
#pragma unroll
  for (int J = 0; J < NB; ++J)
    if (g == 0 && cok[J]) {
      if (A.g_a0) A.g_a0[b * (long)m + 16 * J + c] = abp[J];
      if (A.g_c) A.g_c[b * (long)m + 16 * J + c] = cb[J];
    }
  if (A.g_Z)
    for (int e = lane; e < p * m; e += KF_WAVE) {
      const int q = e / m, k = e - q * m;
      A.g_Z[b * (long)p * m + e] = ZB[q * NCOL + k];
    }
  if (A.g_H)
    for (int e = lane; e < p * p; e += KF_WAVE) {
      const int q1 = e / p, q2 = e - q1 * p;
      A.g_H[b * (long)p * p + e] = (q1 == q2) ? HB[q1] : 0.0;
    }
  if (A.g_d && lane < p) A.g_d[b * (long)p + lane] = DB[lane];
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      if (!(cok[J] && rok(s))) continue;
      const long o = b * (long)mm + obase + 4 * s * m + 16 * J;
      if (A.g_T) A.g_T[o] = 2.0 * TBm[s >> 2][J][s & 3];
      if (A.g_P0 && p > 1) A.g_P0[o] = PB[J][s];
    }
  // R_b = Gs R (Q + Q') ;  Q_b = R' Gs R   (Gs in FT, Gs R staged in GRS)
  wave_lds_fence();
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  if (A.g_R || A.g_Q) {
    for (int e = lane; e < m * r_; e += KF_WAVE) {
      const int i = e / r_, k = e - i * r_;
      double grk = 0.0;
      for (int j = 0; j < m; ++j) grk = fma(FT[i * LD + j], gR[j * r_ + k], grk);
      GRS[e] = grk;
    }
    wave_lds_fence();
  }
  if (A.g_R)
    for (int e = lane; e < m * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double sacc = 0.0;
      for (int k = 0; k < r_; ++k) sacc = fma(GRS[i * r_ + k], gQ[k * r_ + l] + gQ[l * r_ + k], sacc);
      A.g_R[b * (long)m * r_ + e] = sacc;
    }
  if (A.g_Q)
    for (int e = lane; e < r_ * r_; e += KF_WAVE) {
      const int k = e / r_, l = e - k * r_;
      double sacc = 0.0;
      for (int i = 0; i < m; ++i) sacc = fma(gR[i * r_ + k], GRS[i * r_ + l], sacc);
      A.g_Q[b * (long)r_ * r_ + e] = sacc;
    }
}

template <int NB, int KSR>
int launch_bwd_nb(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)NbBwdLayout<NB, KSR>::doubles(a.p, a.m, a.r);
  hipLaunchKernelGGL((kf_bwd_mfma_nb_kernel<NB, KSR>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

bool kf_backward_mfma_nb_ok(int m, int p, int r) {
  if (!(m > 16 && m <= KF_NB_M_MAX && p >= 1 && p <= PMAXQ)) return false;
  const int d = m <= 32 ? NbBwdLayout<2, 8>::doubles(p, m, r) : NbBwdLayout<3, 12>::doubles(p, m, r);
  return sizeof(double) * (size_t)d <= 64 * 1024;
}

int kf_launch_backward_mfma_nb(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  if (!kf_backward_mfma_nb_ok(a.m, a.p, a.r)) {
    KF_SET_ERR(h, "backward(mfma_nb): m=%d p=%d r=%d out of range (16 < m <= %d, p <= %d)", a.m, a.p, a.r,
               KF_NB_M_MAX, PMAXQ);
    return KF_ERR_UNSUPPORTED;
  }
  switch ((a.m + 3) / 4) {
    case 5: return launch_bwd_nb<2, 5>(h, a, stream);
    case 6: return launch_bwd_nb<2, 6>(h, a, stream);
    case 7: return launch_bwd_nb<2, 7>(h, a, stream);
    case 8: return launch_bwd_nb<2, 8>(h, a, stream);
    case 9: return launch_bwd_nb<3, 9>(h, a, stream);
    case 10: return launch_bwd_nb<3, 10>(h, a, stream);
    case 11: return launch_bwd_nb<3, 11>(h, a, stream);
    default: return launch_bwd_nb<3, 12>(h, a, stream);
  }
}
