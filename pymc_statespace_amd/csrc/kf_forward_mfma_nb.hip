// kf_forward_mfma_nb.hip -- forward Kalman filter for 16 < k_states <= 48: the covariance as
// NB x NB blocks of 16 x 16 in the accumulator layout of v_mfma_f64_16x16x4_f64, one wavefront
// per draw, static system matrices, p <= 8 scalar observations processed one at a time
// (reference kalman_filter.py:255-351 for p == 1; :444-505 sequential processing, which with a
// diagonal H is algebraically the joint Standard / Cholesky update).
//
// Why: between the single-tile kernel (kf_forward_mfma.hip, m <= 16) and the HBM-workspace GEMM path
// (kf_forward_big.hip, m > ~45) the only kernel was the generic LDS-resident one, whose m^3 products
// run on LDS broadcast reads: 25 ms (m = 17) to 670 ms (m = 40) per 1000 steps at 4096 draws.
//
// Layout.  Row index i = 4 s + g (s < KSR = ceil(m / 4) "slices", g = lane >> 4), column index
// j = 16 J + c (J < NB, c = lane & 15): register X[J][s] holds X[4 s + g][16 J + c].  For the MFMA
// a 16 x 16 output block (I, J) is the four registers s = 4 I .. 4 I + 3 of column block J.
// With P bitwise symmetric, register P[I][ks] is at once the A operand "rows 16 I + c, k = 4 ks + g"
// and the B operand "k = 4 ks + g, columns 16 I + c", so
//     W2 = P_f T'        : W2(I, J) += mfma(A = P[I][ks],  B = T[J][ks])      NB^2 KSR MFMAs
//     S  = T W2 + R Q R' : S(I, J)  += mfma(A = T[I][ks],  B = W2[J][ks])     NB (NB + 1) / 2 KSR MFMAs
//                          (blocks I <= J only; the blocks below the diagonal are mirrored)
// take their operands straight from registers (T[J][s] = T[16 J + c][4 s + g] is static).  One LDS
// transpose gives P_hat = (S + S') / 2 (:221).  Update, likelihood bookkeeping and the tape
// stores follow kf_forward_mfma.hip (non-augmented variant), with vectors held per column block.
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int PMAXQ = KF_MP_P_MAX;

template <int NB, int KSR>
struct NbLayout {
  static constexpr int KSRP = KSR + (KSR & 1);   // slices per lane group, padded to whole double2
  static constexpr int LD = 4 * KSRP + 2;        // row stride of the transpose tile (16 B aligned, bank-skewed)
  static constexpr int NCOL = 16 * NB;
  static constexpr int OFF_EXM = 4 * KSR * LD;   // transpose tile, also the m x m tape staging image
  static constexpr int OFF_EXA = OFF_EXM + 4 * KSRP;
  static constexpr int OFF_RG = OFF_EXA + 4 * KSRP;
  static constexpr int OFF_HD = OFF_RG + 128;
  static constexpr int OFF_DL = OFF_HD + PMAXQ;
  static constexpr int OFF_YB = OFF_DL + PMAXQ;
  static int loop_doubles(int p) { return OFF_YB + 64 * p + p * NCOL + p * 4 * KSRP; }
};
inline int nb_setup_doubles(int m, int r) { return 2 * m * r + r * r; }

template <int NB, int KSR>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(NB == 2 ? 2 : 1))) void kf_fwd_mfma_nb_kernel(const KfFwdArgs A) {
  using L = NbLayout<NB, KSR>;
  constexpr int KSRP = L::KSRP, LD = L::LD, NCOL = L::NCOL;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, p = A.p, mm = m * m;

  double* TR = smem;  // transpose tile; between uses the compact m x m tape image (STG)
  double* STG = smem;
  double* EXM = smem + L::OFF_EXM;
  double* EXA = smem + L::OFF_EXA;
  double* RG = smem + L::OFF_RG;
  double* HD = smem + L::OFF_HD;
  double* DL = smem + L::OFF_DL;
  double* YB = smem + L::OFF_YB;
  double* ZL = YB + 64 * p;      // Z_q by column index          p x NCOL
  double* ZP = ZL + p * NCOL;    // Z_q by row index, per group  p x 4 KSRP: [g * KSRP + s] = Z_q[4 s + g]

  bool cok[NB];
#pragma unroll
  for (int J = 0; J < NB; ++J) cok[J] = 16 * J + c < m;
  auto rok = [&](int s) { return 4 * s + g < m; };
  // position of column index j in the "4 s + g" ordering of an exchange vector / transpose row
  auto pos = [&](int j) { return (j & 3) * KSRP + (j >> 2); };

  const double* gP = A.P0 + b * (long)mm;
  const double* gZ = A.Z + b * (long)p * m;
  double TA[NB][KSR], RQR[NB][KSR], P[NB][KSR];
  double cc[NB], ac[NB];
  {
    // ---- R Q R' through an LDS scratch that aliases the loop buffers --------------------------------
    double* sR = smem;              // R     m x r
    double* sQ = sR + m * r_;       // Q     r x r
    double* sRQ = sQ + r_ * r_;     // R Q   m x r
    const double* gR = A.R + b * (long)m * r_;
    const double* gQ = A.Q + b * (long)r_ * r_;
    for (int e = lane; e < m * r_; e += KF_WAVE) sR[e] = gR[e];
    for (int e = lane; e < r_ * r_; e += KF_WAVE) sQ[e] = gQ[e];
    __syncthreads();
    for (int e = lane; e < m * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      for (int kk = 0; kk < r_; ++kk) v += sR[i * r_ + kk] * sQ[kk * r_ + l];
      sRQ[e] = v;
    }
    __syncthreads();
    const double* gT = A.Tm + b * (long)mm;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        const int k = 4 * s + g;
        const bool on = cok[J] && k < m;
        double acc = 0.0;
        if (on)
          for (int l = 0; l < r_; ++l) acc += sRQ[k * r_ + l] * sR[j * r_ + l];
        RQR[J][s] = acc;
        TA[J][s] = on ? gT[j * m + k] : 0.0;
        // P0 as the reference reads it through the symmetric recursion: upper triangle
        P[J][s] = on ? ((j >= k) ? gP[k * m + j] : gP[j * m + k]) : 0.0;
      }
      cc[J] = (cok[J] && A.c) ? A.c[b * (long)m + j] : 0.0;
      ac[J] = cok[J] ? A.a0[b * (long)m + j] : 0.0;
    }
    __syncthreads();  // scratch is dead: the loop buffers may be written
  }
  for (int e = lane; e < p * NCOL; e += KF_WAVE) {
    const int q = e / NCOL, k = e - q * NCOL;
    ZL[e] = (k < m) ? gZ[q * m + k] : 0.0;
  }
  for (int e = lane; e < p * 4 * KSRP; e += KF_WAVE) {
    const int q = e / (4 * KSRP), w = e - q * 4 * KSRP;
    const int gg = w / KSRP, s = w - gg * KSRP, k = 4 * s + gg;
    ZP[e] = (s < KSR && k < m) ? gZ[q * m + k] : 0.0;
  }
  if (lane < p) {
    HD[lane] = A.H[b * (long)p * p + lane * p + lane];
    DL[lane] = A.d ? A.d[b * (long)p + lane] : 0.0;
  }
  wave_lds_fence();

  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double joint_const = (A.filter == KF_FILTER_CHOLESKY) ? (double)p : 1.0;

  double* o_fa = A.filt_a ? A.filt_a + b * (long)n * m : nullptr;
  double* o_pa = A.pred_a ? A.pred_a + b * (long)(n + 1) * m : nullptr;
  double* o_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* o_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;

  auto store_vec = [&](double* dst, const double (&x)[NB]) {
#pragma unroll
    for (int J = 0; J < NB; ++J)
      if (g == 0 && cok[J]) st_stream(dst + 16 * J + c, x[J]);
  };
  // covariance tapes: registers -> compact m x m image in LDS -> 16 bytes per lane on consecutive
  // addresses (full lines)
  const int sbase = g * m + c;
  auto stage_cov = [&](const double (&X)[NB][KSR]) {
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s)
        if (cok[J] && rok(s)) STG[sbase + 4 * s * m + 16 * J] = X[J][s];
  };
  const int nb128 = mm >> 1;
  auto flush_cov = [&](double* dst) {
    wave_lds_fence();
    for (int e = lane; e < nb128; e += KF_WAVE)
      st_stream(reinterpret_cast<double2_t*>(dst) + e, *reinterpret_cast<const double2_t*>(STG + 2 * e));
    if ((mm & 1) && lane == 0) st_stream(dst + mm - 1, STG[mm - 1]);
    wave_lds_fence();
  };

  // predicted tape slot 0 = (a0, P0 as given)
  if (o_pa) {
    store_vec(o_pa, ac);
    o_pa += m;
  }
  if (o_pP) {
    for (int e = lane; e < mm; e += KF_WAVE) st_stream(o_pP + e, gP[e]);
    o_pP += mm;
  }
  bool pred_pending = false;  // a predicted covariance sits in STG, flushed inside the next update

  double logp = 0.0;  // every lane accumulates the same total
  int info = 0;

  // evaluate the ll of the steps [tb, tb+cnt) held in the ring: one step per lane
  auto flush_ring = [&](int tb, int cnt) {
    wave_lds_fence();
    const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * lane);
    const bool act = lane < cnt;
    // (mantissa product, exponents ln 2 + quadratic form + n log 2 pi); -1: no contribution
    const double xv = act ? fq.x : -1.0;
    const double llv = (xv == -1.0) ? 0.0 : -0.5 * (log(xv) + fq.y);
    if (act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
    wave_lds_fence();
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
      const double Hq = HD[q], dq = DL[q];
      double zr[KSRP];
#pragma unroll
      for (int s = 0; s < KSRP; s += 2) {
        const double2_t z2 = *reinterpret_cast<const double2_t*>(ZP + q * 4 * KSRP + g * KSRP + s);
        zr[s] = z2.x;
        zr[s + 1] = z2.y;
      }
      // ---- M = P Z_q' by column block: in-lane over the slices, then over the lane groups ------------
      double Mc[NB], fz = 0.0, az = 0.0;
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        double pc = P[J][0] * zr[0];
#pragma unroll
        for (int s = 1; s < KSR; ++s) pc = fma(P[J][s], zr[s], pc);
        Mc[J] = col_allsum(pc);
        const int j = 16 * J + c;
        if (g == 0 && j < 4 * KSR) EXM[pos(j)] = Mc[J];
        const double zc = ZL[q * NCOL + j];
        fz = (J == 0) ? zc * Mc[J] : fma(zc, Mc[J], fz);
        az = (J == 0) ? zc * ac[J] : fma(zc, ac[J], az);
      }
      const double F = Hq + row_allsum(fz);
      const double zs = row_allsum(az);
      double v = (yq + dsign * dq) - zs;
      // missing -> skipped; UnivariateFilter also skips F == 0 (:468-471)
      const bool degenerate = !obs || (univariate && F == 0.0);
      const bool notpd = !degenerate && !(F > 0.0);
      if (notpd && info == 0) info = ((t + 1) << 8) | (q + 1);
      const double invF = degenerate ? 0.0 : fast_rcp(F);
      if (degenerate) v = 0.0;
      bad = bad || notpd;
      nob += obs ? 1 : 0;
      if (!degenerate) {
        int e2;
        mant *= frexp(F, &e2);
        ex += e2;
        qs = fma(v * v, invF, qs);
        cnt_ll += (F == 1.0 && v == 0.0) ? 0 : 1;  // :497-500 counts non-zero scalar terms
      }
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        ac[J] = fma(Mc[J] * invF, v, ac[J]);
        const int j = 16 * J + c;
        if (q == p - 1 && g == 0 && j < 4 * KSR) EXA[pos(j)] = ac[J];
      }
      wave_lds_fence();
      double Mr[KSRP];
#pragma unroll
      for (int s = 0; s < KSRP; s += 2) {
        const double2_t m2 = *reinterpret_cast<const double2_t*>(EXM + g * KSRP + s);
        Mr[s] = m2.x;
        Mr[s + 1] = m2.y;
      }
      if (q == 0 && pred_pending) {  // predicted covariance of the previous step: LDS image -> HBM
        flush_cov(o_pP);
        o_pP += mm;
        pred_pending = false;
      }
      // P -= M M'/F, bitwise symmetric (M_i * M_j commutes)
#pragma unroll
      for (int J = 0; J < NB; ++J)
#pragma unroll
        for (int s = 0; s < KSR; ++s) P[J][s] = fma(-(Mr[s] * Mc[J]), invF, P[J][s]);
      wave_lds_fence();  // EXM is rewritten by the next observation
    }
    double ar[KSRP];
#pragma unroll
    for (int s = 0; s < KSRP; s += 2) {
      const double2_t a2 = *reinterpret_cast<const double2_t*>(EXA + g * KSRP + s);
      ar[s] = a2.x;
      ar[s + 1] = a2.y;
    }
    if (lane == 0) {
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
      store_vec(o_fa, ac);
      o_fa += m;
    }
    if (o_fP) stage_cov(P);  // flushed below, behind the MFMAs
    // ---- predict: a_hat = T a_f + c ; P_hat = sym(T P_f T' + RQR') ----------------------------------------
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      double part = TA[J][0] * ar[0];
#pragma unroll
      for (int s = 1; s < KSR; ++s) part = fma(TA[J][s], ar[s], part);
      ac[J] = cc[J] + col_allsum(part);
    }
    double4_t W2[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) W2[I][J] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J)
          if (4 * I < KSR) W2[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[I][ks], TA[J][ks], W2[I][J], 0, 0, 0);
    if (o_fP) {
      flush_cov(o_fP);
      o_fP += mm;
    }
    // S = T W2 + R Q R' is symmetric up to rounding: only the blocks on and above the block diagonal
    // are computed; the blocks below are their mirror images (through the LDS transpose)
    double4_t S[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J)
#pragma unroll
        for (int u = 0; u < 4; ++u) S[I][J][u] = (4 * I + u < KSR) ? RQR[J][(4 * I + u < KSR) ? 4 * I + u : 0] : 0.0;
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J)
          S[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(TA[I][ks], W2[ks >> 2][J][ks & 3], S[I][J], 0, 0, 0);
    // transpose through LDS: element (i, j) at TR[i * LD + pos(j)]
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
#pragma unroll
      for (int s = 0; s < KSR; ++s)
        if ((s >> 2) <= J && j < 4 * KSR) TR[(4 * s + g) * LD + pos(j)] = S[s >> 2][J][s & 3];
    }
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
      const double* row = TR + (j < 4 * KSR ? j : 4 * KSR - 1) * LD + g * KSRP;
#pragma unroll
      for (int s = 0; s < 4 * J && s < KSR; ++s) P[J][s] = S[s >> 2][J][s & 3];  // above the diagonal: as computed
#pragma unroll
      for (int s = 4 * J; s < KSRP; s += 2) {
        const double2_t t2 = *reinterpret_cast<const double2_t*>(row + s);
        const bool dg0 = (s >> 2) == J, dg1 = ((s + 1) >> 2) == J;  // diagonal block: (S + S') / 2 (:221)
        const double v0 = dg0 ? 0.5 * (S[dg0 ? J : 0][J][s & 3] + t2.x) : t2.x;
        P[J][s] = cok[J] ? v0 : 0.0;
        if (s + 1 < KSR) {
          const double v1 = dg1 ? 0.5 * (S[dg1 ? J : 0][J][(s + 1) & 3] + t2.y) : t2.y;
          P[J][s + 1 < KSR ? s + 1 : 0] = cok[J] ? v1 : 0.0;
        }
      }
    }
    wave_lds_fence();
    if (o_pa) {
      store_vec(o_pa, ac);
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

template <int NB, int KSR>
int launch_nb(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  const int loop = NbLayout<NB, KSR>::loop_doubles(a.p), setup = nb_setup_doubles(a.m, a.r);
  const size_t lds = sizeof(double) * (size_t)(loop > setup ? loop : setup);
  hipLaunchKernelGGL((kf_fwd_mfma_nb_kernel<NB, KSR>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

bool kf_forward_mfma_nb_ok(int m, int p, int r) {
  return m > 16 && m <= KF_NB_M_MAX && p >= 1 && p <= PMAXQ &&
         sizeof(double) * (size_t)nb_setup_doubles(m, r) <= 64 * 1024;
}

int kf_launch_forward_mfma_nb(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  if (!kf_forward_mfma_nb_ok(a.m, a.p, a.r)) {
    KF_SET_ERR(h, "forward(mfma_nb): m=%d p=%d r=%d out of range (16 < m <= %d, p <= %d)", a.m, a.p, a.r,
               KF_NB_M_MAX, PMAXQ);
    return KF_ERR_UNSUPPORTED;
  }
  switch ((a.m + 3) / 4) {
    case 5: return launch_nb<2, 5>(h, a, stream);
    case 6: return launch_nb<2, 6>(h, a, stream);
    case 7: return launch_nb<2, 7>(h, a, stream);
    case 8: return launch_nb<2, 8>(h, a, stream);
    case 9: return launch_nb<3, 9>(h, a, stream);
    case 10: return launch_nb<3, 10>(h, a, stream);
    case 11: return launch_nb<3, 11>(h, a, stream);
    default: return launch_nb<3, 12>(h, a, stream);
  }
}
