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
#include "kf_internal.h"

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  __builtin_nontemporal_store(v, p);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// all-reduce over the 16 lanes of a DPP row (same g): every lane gets the bit-identical sum
__device__ __forceinline__ double row_allsum(double x) {
  // no FMA contraction with the caller's product: fma(a, b, other lane's ROUNDED a*b) would give
  // the two lanes of a pair different bits (-ffp-contract=fast fuses in the backend whatever the
  // pragma says, so the value is pinned with an empty asm)
  asm volatile("" : "+v"(x));
  x += dpp_mov<0x128>(x);  // row_ror:8
  x += dpp_mov<0x124>(x);  // row_ror:4
  x += dpp_mov<0x122>(x);  // row_ror:2
  x += dpp_mov<0x121>(x);  // row_ror:1
  return x;
}

// all-reduce over the 4 lane groups g (lanes c, c+16, c+32, c+48): (x0 + x1) + (x2 + x3) everywhere
__device__ __forceinline__ double col_allsum(double x) {
  asm volatile("" : "+v"(x));
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(x);
  hi = __double2hiint(x);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}

constexpr int TILE = 16;
constexpr int PMAXQ = KF_MP_P_MAX;
// LDS (doubles): transpose tile | M exchange | a exchange | ring | y block | Z | Z permuted | diag H | d
constexpr int OFF_EXM = TILE * TILE;
constexpr int OFF_EXA = OFF_EXM + TILE;
constexpr int OFF_RG = OFF_EXA + TILE;
constexpr int OFF_YB = OFF_RG + 128;
constexpr int OFF_ZL = OFF_YB + 64 * PMAXQ;
constexpr int OFF_ZP = OFF_ZL + TILE * PMAXQ;
constexpr int OFF_HD = OFF_ZP + TILE * PMAXQ;
constexpr int OFF_DL = OFF_HD + PMAXQ;
constexpr int LDS_DOUBLES = OFF_DL + PMAXQ;

template <bool P1>
__global__ __launch_bounds__(KF_WAVE) void kf_fwd_mfma_kernel(const KfFwdArgs A) {
  __shared__ __attribute__((aligned(16))) double smem[LDS_DOUBLES];
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const int permc = (c & 3) * 4 + (c >> 2);  // position of index c in the "g + 4r" ordering
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, p = P1 ? 1 : A.p, mm = m * m;

  double* TR = smem;
  double* EXM = smem + OFF_EXM;
  double* EXA = smem + OFF_EXA;
  double* RG = smem + OFF_RG;
  double* YB = smem + OFF_YB;
  double* ZL = smem + OFF_ZL;
  double* ZP = smem + OFF_ZP;
  double* HD = smem + OFF_HD;
  double* DL = smem + OFF_DL;

  // ---- per-draw constants -----------------------------------------------------------------------
  double TA[4], TAh[4], RQRh[4], P[4];
  bool on[4];
  int off[4];
  const bool c_ok = c < m;
  {
    const double* gT = A.Tm + b * (long)mm;
    const double* gP = A.P0 + b * (long)mm;
    const double* gR = A.R + b * (long)m * r_;
    const double* gQ = A.Q + b * (long)r_ * r_;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = 4 * s + g;  // row index of accumulator register s / k index of MFMA step s
      const bool ok = c_ok && k < m;
      on[s] = ok;
      off[s] = k * m + c;
      TA[s] = ok ? gT[c * m + k] : 0.0;
      TAh[s] = 0.5 * TA[s];
      P[s] = ok ? ((c >= k) ? gP[k * m + c] : gP[c * m + k]) : 0.0;  // upper triangle: bitwise symmetric
      double sacc = 0.0;
      if (ok) {
        for (int l = 0; l < r_; ++l) {
          double rq = 0.0;
          for (int kk = 0; kk < r_; ++kk) rq += gR[k * r_ + kk] * gQ[kk * r_ + l];
          sacc += rq * gR[c * r_ + l];
        }
      }
      RQRh[s] = 0.5 * sacc;
    }
  }
  double Zc1 = 0.0, Zr1[4] = {0.0, 0.0, 0.0, 0.0}, H1 = 0.0, d1 = 0.0;
  if (P1) {
    const double* gZ = A.Z + b * (long)m;
    Zc1 = c_ok ? gZ[c] : 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) Zr1[s] = (4 * s + g < m) ? gZ[4 * s + g] : 0.0;
    H1 = A.H[b];
    d1 = A.d ? A.d[b] : 0.0;
  } else {
    const double* gZ = A.Z + b * (long)p * m;
    for (int e = lane; e < p * TILE; e += KF_WAVE) {
      const int q = e >> 4, k = e & 15;
      const double z = (k < m) ? gZ[q * m + k] : 0.0;
      ZL[e] = z;
      ZP[q * TILE + (k & 3) * 4 + (k >> 2)] = z;
    }
    if (lane < p) {
      HD[lane] = A.H[b * (long)p * p + lane * p + lane];
      DL[lane] = A.d ? A.d[b * (long)p + lane] : 0.0;
    }
  }
  const double cc = (A.c && c_ok) ? A.c[b * (long)m + c] : 0.0;
  double ac = c_ok ? A.a0[b * (long)m + c] : 0.0;

  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double joint_const = (A.filter == KF_FILTER_CHOLESKY) ? (double)p : 1.0;

  const bool vec_on = g == 0 && c_ok;
  double* o_fa = (A.filt_a && vec_on) ? A.filt_a + b * (long)n * m + c : nullptr;
  double* o_pa = (A.pred_a && vec_on) ? A.pred_a + b * (long)(n + 1) * m + c : nullptr;
  double* o_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* o_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;

  // predicted tape slot 0 = (a0, P0 as given)
  if (o_pa) { *o_pa = ac; o_pa += m; }
  if (o_pP) {
    const double* gP = A.P0 + b * (long)mm;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (on[s]) st_stream(o_pP + off[s], gP[off[s]]);
    o_pP += mm;
  }

  double logp = 0.0;  // every lane accumulates the same total
  int info = 0;

  // evaluate the ll of the steps [tb, tb+cnt) held in the ring: one step per lane
  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
    const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * lane);
    const bool act = lane < cnt;
    const double xv = act ? fq.x : -1.0;
    // -1 marks a step without likelihood contribution; NaN a not-PD innovation covariance
    const double llv = (xv == -1.0) ? 0.0 : -0.5 * (log(xv) + fq.y);
    if (act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
  };

  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L);

  for (int t = 0; t < n; ++t) {
    if ((t & 63) == 0) {
      const int cnt = ((n - t) < 64 ? (n - t) : 64) * p;
      for (int e = lane; e < cnt; e += KF_WAVE) YB[e] = ybase[(long)t * p + e];
      __syncthreads();
    }
    const double* yrow = YB + (t & 63) * p;
    double mant = 1.0, qs = 0.0;
    int ex = 0, cnt_ll = 0, nob = 0;
    bool bad = false;
    double Mr[4], invF = 0.0, Mc = 0.0;
#pragma unroll 1
    for (int q = 0; q < p; ++q) {
      const double yq = yrow[q];
      const bool obs = !(yq != yq);
      double zc, zr[4], Hq, dq;
      if (P1) {
        zc = Zc1;
#pragma unroll
        for (int s = 0; s < 4; ++s) zr[s] = Zr1[s];
        Hq = H1;
        dq = d1;
      } else {
        zc = ZL[q * TILE + c];
        const double2_t z01 = *reinterpret_cast<const double2_t*>(ZP + q * TILE + 4 * g);
        const double2_t z23 = *reinterpret_cast<const double2_t*>(ZP + q * TILE + 4 * g + 2);
        zr[0] = z01.x; zr[1] = z01.y; zr[2] = z23.x; zr[3] = z23.y;
        Hq = HD[q];
        dq = DL[q];
      }
      // ---- M_c = (P Z_q')_c: in-lane over r, then over the lane groups ------------------------------
      double pc = P[0] * zr[0];
      pc = fma(P[1], zr[1], pc);
      pc = fma(P[2], zr[2], pc);
      pc = fma(P[3], zr[3], pc);
      Mc = col_allsum(pc);
      if (g == 0) EXM[permc] = Mc;
      const double F = Hq + row_allsum(zc * Mc);
      const double zs = row_allsum(zc * ac);
      double v = (yq + dsign * dq) - zs;
      // missing -> skipped; UnivariateFilter also skips F == 0 (:468-471)
      const bool degenerate = !obs || (univariate && F == 0.0);
      const bool notpd = !degenerate && !(F > 0.0);
      if (notpd && info == 0) info = ((t + 1) << 8) | (q + 1);
      bad = bad || notpd;
      invF = degenerate ? 0.0 : 1.0 / F;
      if (degenerate) v = 0.0;
      nob += obs ? 1 : 0;
      if (!degenerate) {
        int e2;
        mant *= frexp(F, &e2);
        ex += e2;
        qs = fma(v * v, invF, qs);
        cnt_ll += (F == 1.0 && v == 0.0) ? 0 : 1;  // :497-500 counts non-zero scalar terms
      }
      ac = fma(Mc * invF, v, ac);
      if (q == p - 1 && g == 0) EXA[permc] = ac;
      __syncthreads();
      {
        const double2_t m01 = *reinterpret_cast<const double2_t*>(EXM + 4 * g);
        const double2_t m23 = *reinterpret_cast<const double2_t*>(EXM + 4 * g + 2);
        Mr[0] = m01.x; Mr[1] = m01.y; Mr[2] = m23.x; Mr[3] = m23.y;
      }
      // P -= M M'/F, bitwise symmetric (M_i * M_j commutes)
#pragma unroll
      for (int s = 0; s < 4; ++s) P[s] = fma(-(Mr[s] * Mc), invF, P[s]);
    }
    double ar[4];
    {
      const double2_t a01 = *reinterpret_cast<const double2_t*>(EXA + 4 * g);
      const double2_t a23 = *reinterpret_cast<const double2_t*>(EXA + 4 * g + 2);
      ar[0] = a01.x; ar[1] = a01.y; ar[2] = a23.x; ar[3] = a23.y;
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
    if (o_fa) { *o_fa = ac; o_fa += m; }
    if (o_fP) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (on[s]) st_stream(o_fP + off[s], P[s]);
      o_fP += mm;
    }
    // ---- predict: a_hat = T a_f + c ; P_hat = sym(T P_f T' + RQR') ----------------------------------------
    double part = TA[0] * ar[0];
    part = fma(TA[1], ar[1], part);
    part = fma(TA[2], ar[2], part);
    part = fma(TA[3], ar[3], part);
    ac = cc + col_allsum(part);
    double4_t W2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) W2 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], TA[s], W2, 0, 0, 0);
    double4_t S = {RQRh[0], RQRh[1], RQRh[2], RQRh[3]};
#pragma unroll
    for (int s = 0; s < 4; ++s) S = __builtin_amdgcn_mfma_f64_16x16x4f64(TAh[s], W2[s], S, 0, 0, 0);
    // transpose S/2 through LDS: element (i, j) at TR[i*16 + perm(j)]
#pragma unroll
    for (int s = 0; s < 4; ++s) TR[(4 * s + g) * TILE + permc] = S[s];
    __syncthreads();
    {
      const double2_t t01 = *reinterpret_cast<const double2_t*>(TR + c * TILE + 4 * g);
      const double2_t t23 = *reinterpret_cast<const double2_t*>(TR + c * TILE + 4 * g + 2);
      P[0] = S[0] + t01.x;
      P[1] = S[1] + t01.y;
      P[2] = S[2] + t23.x;
      P[3] = S[3] + t23.y;
    }
    if (o_pa) { *o_pa = ac; o_pa += m; }
    if (o_pP) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (on[s]) st_stream(o_pP + off[s], P[s]);
      o_pP += mm;
    }
    if ((t & 63) == 63) flush_ring(t - 63, 64);
  }
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
  if (a.p == 1)
    hipLaunchKernelGGL(kf_fwd_mfma_kernel<true>, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a);
  else
    hipLaunchKernelGGL(kf_fwd_mfma_kernel<false>, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
