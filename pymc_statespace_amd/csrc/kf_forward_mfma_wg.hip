// kf_forward_mfma_wg.hip -- forward Kalman filter for 48 < k_states <= 64: one WORKGROUP of four
// wavefronts per draw, wavefront J owning column block J (columns 16 J .. 16 J + 15) of every
// m x m matrix in MFMA accumulator layout (register X[s] = X[4 s + g][16 J + c], s < KSR =
// ceil(m / 4)).  Static system matrices, p <= 8 scalar observations (reference
// kalman_filter.py:255-351 for p == 1; :444-505 sequential processing, algebraically the joint
// update when H is diagonal).
//
// Why: a single wavefront cannot hold a 64 x 64 covariance plus its products (4 x 64 doubles per
// lane), and the HBM-workspace path (kf_forward_big.hip) streams ~10 m^2-sized passes per step
// through HBM: 0.62 s per 1000 steps at 4096 draws, m = 64; the LDS-resident kernel 2 - 4 s.
//
// Per step:  update per observation: M = P z (own columns: in-lane + permlane sums), all-gather of
// M through LDS (one barrier), F, v, the state by ROW index and the likelihood bookkeeping
// redundantly in every wavefront; predict:
//     W2(:, J) = P_f T'(:, J)   A operand = P_f[I][ks] from an LDS image every wavefront writes its
//                               column block of (P_f symmetric), B = T[J][ks] (own registers)
//     S(:, J)  = T W2(:, J) + RQR'(:, J)   A operand = T[I][ks] from its static LDS image
//     P_hat = (S + S') / 2 (:221) through a 64 x 64 LDS transpose tile that reuses the image's
//     buffer (p + 3 barriers per step).  (4 + 3) x KSR MFMAs per wavefront and step (S is symmetric:
//     one block of every mirrored pair), operands never
//     leave the CU; ~70 KB of LDS per draw: two draws per CU hide each other's barrier latency.
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int PMAXQ = KF_MP_P_MAX;
constexpr int NBW = 4;  // wavefronts per draw = column blocks

template <int KSR>
struct WgLayout {
  static constexpr int KSRP = KSR + (KSR & 1);
  static constexpr int LD = 4 * KSRP + 2;
  static constexpr int NCOL = 16 * NBW;
  static constexpr int IMG = NBW * KSR * KF_WAVE;
  static constexpr int BUF = (4 * KSR * LD > IMG) ? 4 * KSR * LD : IMG;
  static constexpr int OFF_PL = IMG;                           // TL: T operand image
  static constexpr int OFF_TR = OFF_PL;                        // PL: P_f operand image, then (one barrier later)
  static constexpr int OFF_EXM = OFF_PL + BUF;                 // TR: the transpose tile -- ~70 KB: two draws per CU
  static constexpr int OFF_EXA = OFF_EXM + 2 * 4 * KSRP;       // EXM: double buffered
  static constexpr int OFF_RG = OFF_EXA + 4 * KSRP;
  static constexpr int OFF_HD = OFF_RG + 128;
  static constexpr int OFF_DL = OFF_HD + PMAXQ;
  static constexpr int OFF_YB = OFF_DL + PMAXQ;
  static int loop_doubles(int p) { return OFF_YB + 64 * p + p * 4 * KSRP; }
  // setup scratch (R, R Q) aliases the whole allocation
  static int doubles(int p, int m, int r) {
    const int loop = loop_doubles(p), setup = 2 * m * r;
    return loop > setup ? loop : setup;
  }
  static bool setup_fits(int p, int m, int r) { return sizeof(double) * (size_t)doubles(p, m, r) <= 80 * 1024; }
};

template <int KSR>
__global__ __launch_bounds__(NBW* KF_WAVE) __attribute__((amdgpu_waves_per_eu(2))) void kf_fwd_mfma_wg_kernel(const KfFwdArgs A) {
  using L = WgLayout<KSR>;
  constexpr int KSRP = L::KSRP, LD = L::LD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, J = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, p = A.p, mm = m * m;

  double* TL = smem;
  double* PL = smem + L::OFF_PL;
  double* TR = smem + L::OFF_TR;
  double* EXM = smem + L::OFF_EXM;
  double* EXA = smem + L::OFF_EXA;
  double* RG = smem + L::OFF_RG;
  double* HD = smem + L::OFF_HD;
  double* DL = smem + L::OFF_DL;
  double* YB = smem + L::OFF_YB;
  double* ZP = YB + 64 * p;  // Z_q by row index, per lane group  p x 4 KSRP: [g * KSRP + s] = Z_q[4 s + g]

  const int j = 16 * J + c;  // this lane's column
  const bool cok = j < m;
  // row blocks of S this wavefront computes (bit I): {0,1,2}, {1,2,3}, {2,3}, {3,0} -- see the predict step
  const unsigned mine = (J == 0) ? 0x7u : (J == 1) ? 0xEu : (J == 2) ? 0xCu : 0x9u;
  auto rok = [&](int s) { return 4 * s + g < m; };
  auto pos = [&](int jj) { return (jj & 3) * KSRP + (jj >> 2); };
  auto read_rows = [&](const double* src, double (&x)[KSRP]) {
#pragma unroll
    for (int s = 0; s < KSRP; s += 2) {
      const double2_t v2 = *reinterpret_cast<const double2_t*>(src + g * KSRP + s);
      x[s] = v2.x;
      x[s + 1] = v2.y;
    }
  };

  const double* gP = A.P0 + b * (long)mm;
  const double* gZ = A.Z + b * (long)p * m;
  double TA[KSR], RQR[KSR], P[KSR];
  {
    // ---- R Q R' through an LDS scratch that aliases the loop buffers (Q is read from global memory) ----
    double* sR = smem;              // R     m x r
    double* sRQ = sR + m * r_;      // R Q   m x r
    const double* gR = A.R + b * (long)m * r_;
    const double* gQ = A.Q + b * (long)r_ * r_;
    for (int e = threadIdx.x; e < m * r_; e += NBW * KF_WAVE) sR[e] = gR[e];
    __syncthreads();
    for (int e = threadIdx.x; e < m * r_; e += NBW * KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      for (int kk = 0; kk < r_; ++kk) v += sR[i * r_ + kk] * gQ[kk * r_ + l];
      sRQ[e] = v;
    }
    __syncthreads();
    const double* gT = A.Tm + b * (long)mm;
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      const int k = 4 * s + g;
      const bool on = cok && k < m;
      double acc = 0.0;
      if (on)
        for (int l = 0; l < r_; ++l) acc += sRQ[k * r_ + l] * sR[j * r_ + l];
      RQR[s] = acc;
      TA[s] = on ? gT[j * m + k] : 0.0;
      // P0 as the reference reads it through the symmetric recursion: upper triangle
      P[s] = on ? ((j >= k) ? gP[k * m + j] : gP[j * m + k]) : 0.0;
    }
    __syncthreads();  // scratch is dead
#pragma unroll
    for (int s = 0; s < KSR; ++s) TL[(J * KSR + s) * KF_WAVE + lane] = TA[s];
  }
  const double cc = (cok && A.c) ? A.c[b * (long)m + j] : 0.0;
  double ac = cok ? A.a0[b * (long)m + j] : 0.0;
  double ar[KSRP];  // the state by row index, kept redundantly in every wavefront
#pragma unroll
  for (int s = 0; s < KSRP; ++s) ar[s] = (s < KSR && rok(s)) ? A.a0[b * (long)m + 4 * s + g] : 0.0;
  for (int e = threadIdx.x; e < p * 4 * KSRP; e += NBW * KF_WAVE) {
    const int q = e / (4 * KSRP), w = e - q * 4 * KSRP;
    const int gg = w / KSRP, s = w - gg * KSRP, k = 4 * s + gg;
    ZP[e] = (s < KSR && k < m) ? gZ[q * m + k] : 0.0;
  }
  if (threadIdx.x < p) {
    HD[threadIdx.x] = A.H[b * (long)p * p + threadIdx.x * p + threadIdx.x];
    DL[threadIdx.x] = A.d ? A.d[b * (long)p + threadIdx.x] : 0.0;
  }
  if (threadIdx.x < 2 * 4 * KSRP) EXM[threadIdx.x] = 0.0;
  if (threadIdx.x < 4 * KSRP) EXA[threadIdx.x] = 0.0;

  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double joint_const = (A.filter == KF_FILTER_CHOLESKY) ? (double)p : 1.0;

  double* o_fa = A.filt_a ? A.filt_a + b * (long)n * m : nullptr;
  double* o_pa = A.pred_a ? A.pred_a + b * (long)(n + 1) * m : nullptr;
  double* o_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* o_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;

  // covariance tapes straight from the registers: per instruction 4 rows x 16 consecutive doubles
  const int obase = g * m + j;
  auto store_cov = [&](double* dst, const double (&X)[KSR]) {
#pragma unroll
    for (int s = 0; s < KSR; ++s)
      if (cok && rok(s)) st_stream(dst + obase + 4 * s * m, X[s]);
  };

  // predicted tape slot 0 = (a0, P0 as given)
  if (o_pa) {
    if (g == 0 && cok) st_stream(o_pa + j, ac);
    o_pa += m;
  }
  if (o_pP) {
    for (int e = threadIdx.x; e < mm; e += NBW * KF_WAVE) st_stream(o_pP + e, gP[e]);
    o_pP += mm;
  }

  double logp = 0.0;  // every lane accumulates the same total
  int info = 0;
  auto flush_ring = [&](int tb, int cnt) {  // ll of the steps [tb, tb+cnt) held in the ring: one step per lane
    const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * lane);
    const bool act = lane < cnt;
    const double xv = act ? fq.x : -1.0;
    const double llv = (xv == -1.0) ? 0.0 : -0.5 * (log(xv) + fq.y);
    if (J == 0 && act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
  };

  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L);
  int par = 0;  // EXM buffer parity
  __syncthreads();

  for (int t = 0; t < n; ++t) {
    if ((t & 63) == 0) {
      // (the barriers of the previous step separate these writes from its last reads of YB / RG)
      const int cnt = ((n - t) < 64 ? (n - t) : 64) * p;
      for (int e = threadIdx.x; e < cnt; e += NBW * KF_WAVE) YB[e] = ybase[(long)t * p + e];
      __syncthreads();
    }
    const double* yrow = YB + (t & 63) * p;
    double mant = 1.0, qs = 0.0;
    int ex = 0, cnt_ll = 0, nob = 0;
    bool bad = false;
#pragma unroll 1
    for (int q = 0; q < p; ++q) {
      const double yq = yrow[q];
      const bool obs = !(yq != yq);
      double zr[KSRP];
      read_rows(ZP + q * 4 * KSRP, zr);
      // ---- M = P Z_q': own columns, then all-gather by row index --------------------------------------
      double pc = P[0] * zr[0];
#pragma unroll
      for (int s = 1; s < KSR; ++s) pc = fma(P[s], zr[s], pc);
      const double Mc = col_allsum(pc);
      double* exm = EXM + par * 4 * KSRP;
      par ^= 1;
      if (g == 0 && j < 4 * KSR) exm[pos(j)] = Mc;
      __syncthreads();
      double Mr[KSRP];
      read_rows(exm, Mr);
      double fz = zr[0] * Mr[0], az = zr[0] * ar[0];
#pragma unroll
      for (int s = 1; s < KSR; ++s) {
        fz = fma(zr[s], Mr[s], fz);
        az = fma(zr[s], ar[s], az);
      }
      const double F = HD[q] + col_allsum(fz);
      const double zs = col_allsum(az);
      double v = (yq + dsign * DL[q]) - zs;
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
      const double vf = v * invF;
      ac = fma(Mc, vf, ac);
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        ar[s] = fma(Mr[s], vf, ar[s]);
        P[s] = fma(-(Mr[s] * Mc), invF, P[s]);  // P -= M M'/F, bitwise symmetric
      }
    }
    if (threadIdx.x == 0) {
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
    // ---- filtered tapes; image of P_f; a_hat = T a_f + c (own columns) -----------------------------------
    if (o_fa) {
      if (g == 0 && cok) st_stream(o_fa + j, ac);
      o_fa += m;
    }
    if (o_fP) {
      store_cov(o_fP, P);
      o_fP += mm;
    }
    double part = TA[0] * ar[0];
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      PL[(J * KSR + s) * KF_WAVE + lane] = P[s];
      if (s > 0) part = fma(TA[s], ar[s], part);
    }
    ac = cc + col_allsum(part);
    if (g == 0 && j < 4 * KSR) EXA[pos(j)] = ac;
    __syncthreads();
    // ---- P_hat = sym(T P_f T' + RQR'), column block J -------------------------------------------------------
    double4_t W2[NBW];
#pragma unroll
    for (int I = 0; I < NBW; ++I) W2[I] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR)
          W2[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(PL[(I * KSR + ks) * KF_WAVE + lane], TA[ks], W2[I], 0, 0, 0);
    __syncthreads();  // every wavefront has read the image: its buffer becomes the transpose tile
    // S is symmetric up to rounding: of every block pair {(I, J), (J, I)} only one is computed, the other
    // is its mirror image through the transpose tile; diagonal blocks are averaged (:221).  `mine`:
    // the row blocks I of column block J this wavefront computes -- 3, 3, 2, 2 blocks instead of 4 each
    double4_t S[NBW];
#pragma unroll
    for (int I = 0; I < NBW; ++I)
#pragma unroll
      for (int u = 0; u < 4; ++u) S[I][u] = (4 * I + u < KSR) ? RQR[(4 * I + u < KSR) ? 4 * I + u : 0] : 0.0;
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR && ((mine >> I) & 1))
          S[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(TL[(I * KSR + ks) * KF_WAVE + lane], W2[ks >> 2][ks & 3], S[I], 0, 0, 0);
    // transpose through LDS: element (i, j) at TR[i * LD + pos(j)]
#pragma unroll
    for (int s = 0; s < KSR; ++s)
      if (((mine >> (s >> 2)) & 1) && j < 4 * KSR) TR[(4 * s + g) * LD + pos(j)] = S[s >> 2][s & 3];
    __syncthreads();
    {
      const double* row = TR + (j < 4 * KSR ? j : 4 * KSR - 1) * LD + g * KSRP;
      double tw[KSRP];
#pragma unroll
      for (int s = 0; s < KSRP; s += 2) {
        const double2_t t2 = *reinterpret_cast<const double2_t*>(row + s);
        tw[s] = t2.x;
        tw[s + 1] = t2.y;
      }
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        const int I = s >> 2;
        const double sv = S[I][s & 3];
        // own diagonal block: average; computed here: as is; computed by the other wavefront: its mirror
        const double v = (I == J) ? 0.5 * (sv + tw[s]) : (((mine >> I) & 1) ? sv : tw[s]);
        P[s] = cok ? v : 0.0;
      }
    }
    read_rows(EXA, ar);
    if (o_pa) {
      if (g == 0 && cok) st_stream(o_pa + j, ac);
      o_pa += m;
    }
    if (o_pP) {
      store_cov(o_pP, P);
      o_pP += mm;
    }
    if ((t & 63) == 63) flush_ring(t - 63, 64);
  }
  if (n & 63) {
    __syncthreads();
    flush_ring(n & ~63, n & 63);
  }
  if (threadIdx.x == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

template <int KSR>
int launch_wg(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)WgLayout<KSR>::doubles(a.p, a.m, a.r);
  static KfAttrOnce attr_set;  // the attribute is per device
  if (attr_set.need(h->device)) {
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&kf_fwd_mfma_wg_kernel<KSR>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, KF_LDS_BYTES_MAX));
    attr_set.mark(h->device);
  }
  hipLaunchKernelGGL((kf_fwd_mfma_wg_kernel<KSR>), dim3((unsigned)a.B), dim3(NBW * KF_WAVE), lds, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

bool kf_forward_mfma_wg_ok(int m, int p, int r) {
  if (!(m > KF_NB_M_MAX && m <= KF_WG_M_MAX && p >= 1 && p <= PMAXQ)) return false;
  switch ((m + 3) / 4) {
    case 13: return WgLayout<13>::setup_fits(p, m, r);
    case 14: return WgLayout<14>::setup_fits(p, m, r);
    case 15: return WgLayout<15>::setup_fits(p, m, r);
    default: return WgLayout<16>::setup_fits(p, m, r);
  }
}

int kf_launch_forward_mfma_wg(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  if (!kf_forward_mfma_wg_ok(a.m, a.p, a.r)) {
    KF_SET_ERR(h, "forward(mfma_wg): m=%d p=%d r=%d out of range (%d < m <= %d, p <= %d)", a.m, a.p, a.r,
               KF_NB_M_MAX, KF_WG_M_MAX, PMAXQ);
    return KF_ERR_UNSUPPORTED;
  }
  switch ((a.m + 3) / 4) {
    case 13: return launch_wg<13>(h, a, stream);
    case 14: return launch_wg<14>(h, a, stream);
    case 15: return launch_wg<15>(h, a, stream);
    default: return launch_wg<16>(h, a, stream);
  }
}
