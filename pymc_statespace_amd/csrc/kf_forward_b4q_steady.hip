// kf_forward_b4q_steady.hip -- the steady-state phase of the multi-observation block kernel (kf_forward_b4q.hip: 3 <= p <= 5 scalar
// observations, m <= 15) as SEPARATE kernels: per-draw constants, the constant covariance tapes, and the mean recursion as a prefix
// scan over time.  Round 2 ran the phase inside the recursion kernel: per wavefront first the bulk fill of both covariance tapes,
// then 33 MFMAs + ~40 other instructions per step for the mean -- C4 (16384 x 1000): 6.1 ms for the log-likelihood alone, 13.0 ms with
// the 63 GB of tapes (4.8 TB/s).  Here the recursion kernel leaves at the convergence point (kf_fwd_b4q_kernel with hand-over
// buffers: tile, partial log-likelihood, step) and
//   kf_b4q_steady_prep_kernel   per draw: F = Z P Z' + H = L D L' (the sequential filter's F_q are the pivots d_q), W = F^-1,
//                               K = P Z' W, G = T K, A = T - G Z and A^(2^k), the filtered covariance P - K F K', both tape images
//   kf_b4q_steady_fill_kernel   element i of a covariance tape = image[i mod m^2], whole aligned lines (the fill of kf_b4_split.h)
//   kf_b4q_steady_scan_kernel   a_{t+1} = A a_t + G y~_t + c as a prefix scan, one step per lane, 63 steps per block (the scheme
//                               of kf_forward_b4_steady.hip), v_t = y~_t - Z a_t, a_filt = a_t + K v_t, and the step's
//                               log-likelihood from the sequential innovations w = L^-1 v: -1/2 (sum w_q^2 / d_q + sum log d_q + k log 2 pi)
//                               with the per-filter count k of kf_forward_b4q.hip (kalman_filter.py:281, :314, :497-503)
// run behind it, the fill and the scan on the handle's two streams.  Same function of the inputs as the in-kernel phase up to
// rounding (joint update with the constant gain instead of p sequential ones).
#include <stdlib.h>

#include "kf_b4_split.h"

namespace {

// per-draw constants in the workspace (doubles); matrices zero-padded to 16 x 16 / 16 x 8 / 8 x 16 / 8 x 8
constexpr int QS_META = 0;                     // [0] sum log d_q  [1] 1 if some pivot is not positive (flag the draw)  [2] #degenerate (Univariate F_q == 0)
constexpr int QS_AP = 8;                       // A^(2^k), k = 0..5: 6 x 256
constexpr int QS_G = QS_AP + 6 * 256;          // G = T K        16 x 8
constexpr int QS_K = QS_G + 128;               // K              16 x 8
constexpr int QS_Z = QS_K + 128;               // Z              8 x 16
constexpr int QS_LI = QS_Z + 128;              // L^-1           8 x 8 (unit lower; rows of degenerate observations zero)
constexpr int QS_D = QS_LI + 64;               // d_q            8
constexpr int QS_DI = QS_D + 8;                // 1 / d_q (0 for a degenerate observation)   8
constexpr int QS_C = QS_DI + 8;                // c              16
constexpr int QS_A = QS_C + 16;                // a at the hand-over step   16
constexpr int QS_DV = QS_A + 16;               // d (observation intercept) 8
constexpr int QS_IMF = QS_DV + 8;              // filtered covariance, compact m x m (+ 1 wrap element)   256
constexpr int QS_IMP = QS_IMF + 256;           // predicted covariance image                              256
constexpr int QS_TOTAL = QS_IMP + 256;
}  // namespace
int kf_b4q_sp_stride() { return QS_TOTAL; }
namespace {

__global__ __launch_bounds__(KF_WAVE) void kf_b4q_steady_prep_kernel(const KfFwdArgs A, double* __restrict__ sp_all) {
  __shared__ __attribute__((aligned(16))) double sm[1792];
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const int n = A.T, m = A.m, p = A.p, mm = m * m;
  const int ts = A.hs_ts[b];
  if (ts >= n) return;
  const int N4 = 4 * ((m + 1 + 3) / 4), JC = N4 - 1;
  double* sp = sp_all + b * (long)QS_TOTAL;
  const double* ht = A.hs_tile + b * (long)B4_HS_STRIDE;
  const double* gT = A.Tm + b * (long)mm;
  const double* gZ = A.Z + b * (long)p * m;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  double* sP = sm;            // 16 x 16
  double* sT = sm + 256;      // 16 x 16
  double* sZ = sm + 512;      // 8 x 16
  double* sM = sm + 640;      // 16 x 8: P Z'
  double* sK = sm + 768;      // 16 x 8
  double* sGm = sm + 896;     // 16 x 8: T K
  double* sF = sm + 1024;     // 8 x 8
  double* sL = sm + 1088;     // 8 x 8
  double* sW = sm + 1152;     // 8 x 8
  double* sA = sm + 1216;     // 16 x 16 (current power)
  double* sB = sm + 1472;     // 16 x 16 (next)  -> 1728
  double* sDinv = sm + 1728;  // 8
  for (int e = lane; e < 1792; e += KF_WAVE) sm[e] = 0.0;
  for (int e = lane; e < QS_TOTAL; e += KF_WAVE) sp[e] = 0.0;
  wave_lds_fence();
  for (int e = lane; e < mm; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    sP[i * 16 + j] = 0.5 * (ht[i * 16 + j] + ht[j * 16 + i]);
    sT[i * 16 + j] = gT[e];
  }
  for (int e = lane; e < p * m; e += KF_WAVE) sZ[(e / m) * 16 + e % m] = gZ[e];
  wave_lds_fence();
  for (int e = lane; e < 128; e += KF_WAVE) {  // M = P Z'
    const int i = e >> 3, q = e & 7;
    double s = 0.0;
    for (int j = 0; j < 16; ++j) s = fma(sP[i * 16 + j], sZ[q * 16 + j], s);
    sM[e] = s;
  }
  wave_lds_fence();
  {  // F = Z M + diag(H), symmetric
    const int q = lane >> 3, r2 = lane & 7;
    double s = (q == r2 && q < p) ? A.H[b * (long)p * p + q * p + q] : 0.0;
    for (int i = 0; i < 16; ++i) s = fma(sZ[q * 16 + i], sM[i * 8 + r2], s);
    sF[lane] = s;
  }
  wave_lds_fence();
  // F = L D L' by lane 0 (p <= 8): the pivots d_q are the F_q of the sequential (Univariate) filter; L^-1 alongside
  if (lane == 0) {
    double Lm[8][8], Li[8][8], dv[8];
    bool bad = false;
    int ndeg = 0;
    for (int q = 0; q < 8; ++q)
      for (int r2 = 0; r2 < 8; ++r2) {
        Lm[q][r2] = (q == r2) ? 1.0 : 0.0;
        Li[q][r2] = (q == r2) ? 1.0 : 0.0;
      }
    for (int q = 0; q < p; ++q) {
      double dq = 0.5 * (sF[q * 8 + q] + sF[q * 8 + q]);
      for (int k = 0; k < q; ++k) dq = fma(-Lm[q][k] * Lm[q][k], dv[k], dq);
      dv[q] = dq;
      const bool deg = univariate && dq == 0.0;  // :468-471: the observation updates nothing
      if (deg) ++ndeg;
      if (!deg && !(dq > 0.0)) bad = true;
      const double idq = (deg || !(dq > 0.0)) ? 0.0 : 1.0 / dq;
      for (int i = q + 1; i < p; ++i) {
        double s = 0.5 * (sF[i * 8 + q] + sF[q * 8 + i]);
        for (int k = 0; k < q; ++k) s = fma(-Lm[i][k] * Lm[q][k], dv[k], s);
        Lm[i][q] = s * idq;
      }
    }
    // L^-1 (unit lower): forward substitution column by column
    for (int cidx = 0; cidx < p; ++cidx)
      for (int i = cidx + 1; i < p; ++i) {
        double s = 0.0;
        for (int k = cidx; k < i; ++k) s = fma(-Lm[i][k], Li[k][cidx], s);
        Li[i][cidx] = s;
      }
    double mant = 1.0;
    int ex = 0;
    for (int q = 0; q < 8; ++q) {
      const bool live = q < p && !(univariate && dv[q] == 0.0);
      sp[QS_D + q] = q < p ? dv[q] : 0.0;
      sDinv[q] = (live && dv[q] > 0.0) ? 1.0 / dv[q] : 0.0;
      sp[QS_DI + q] = sDinv[q];
      if (live) {
        int e2;
        mant *= frexp(dv[q], &e2);
        ex += e2;
      }
      for (int r2 = 0; r2 < 8; ++r2) {
        sL[q * 8 + r2] = (q < p && r2 < p) ? Li[q][r2] : 0.0;
        sp[QS_LI + q * 8 + r2] = (q < p && r2 < p && live) ? Li[q][r2] : 0.0;
      }
    }
    sp[QS_META + 0] = log(mant) + 0.69314718055994530942 * (double)ex;
    sp[QS_META + 1] = bad ? 1.0 : 0.0;
    sp[QS_META + 2] = (double)ndeg;
  }
  wave_lds_fence();
  {  // W = F^-1 = L^-T D^-1 L^-1 (degenerate / non-positive pivots contribute nothing)
    const int q = lane >> 3, r2 = lane & 7;
    double s = 0.0;
    for (int k = 0; k < 8; ++k) {
      const double dk = sDinv[k];
      s = fma(sL[k * 8 + q] * dk, sL[k * 8 + r2], s);
    }
    sW[lane] = (q < p && r2 < p) ? s : 0.0;
  }
  wave_lds_fence();
  for (int e = lane; e < 128; e += KF_WAVE) {  // K = M W
    const int i = e >> 3, q = e & 7;
    double s = 0.0;
    for (int r2 = 0; r2 < 8; ++r2) s = fma(sM[i * 8 + r2], sW[r2 * 8 + q], s);
    sK[e] = s;
    sp[QS_K + e] = s;
  }
  wave_lds_fence();
  for (int e = lane; e < 128; e += KF_WAVE) {  // G = T K
    const int i = e >> 3, q = e & 7;
    double s = 0.0;
    for (int j = 0; j < 16; ++j) s = fma(sT[i * 16 + j], sK[j * 8 + q], s);
    sGm[e] = s;
    sp[QS_G + e] = s;
    sp[QS_Z + e] = sZ[e];
  }
  wave_lds_fence();
  for (int e = lane; e < 256; e += KF_WAVE) {  // A = T - G Z ; images
    const int i = e >> 4, j = e & 15;
    double s = sT[e];
    for (int q = 0; q < 8; ++q) s = fma(-sGm[i * 8 + q], sZ[q * 16 + j], s);
    sA[e] = s;
    sp[QS_AP + e] = s;
  }
  for (int x = lane; x <= mm; x += KF_WAVE) {
    const int xx = x == mm ? 0 : x;
    const int i = xx / m, j = xx - i * m;
    double pf = sP[i * 16 + j];
    for (int q = 0; q < 8; ++q) pf = fma(-sK[i * 8 + q], sM[j * 8 + q], pf);  // P - K M' (= P - K F K')
    sp[QS_IMF + x] = pf;
    sp[QS_IMP + x] = ht[i * 16 + j];  // (the predicted tape continues with the tile as it is)
  }
  if (lane < 16) {
    sp[QS_C + lane] = (lane < m && A.c) ? A.c[b * (long)m + lane] : 0.0;
    sp[QS_A + lane] = lane < m ? 0.5 * ht[lane * 16 + JC] : 0.0;
  }
  if (lane < 8) sp[QS_DV + lane] = (lane < p && A.d) ? A.d[b * (long)p + lane] : 0.0;
  wave_lds_fence();
  for (int k = 1; k < 6; ++k) {
    for (int e = lane; e < 256; e += KF_WAVE) {
      const int i = e >> 4, j = e & 15;
      double v = 0.0;
      for (int l = 0; l < 16; ++l) v = fma(sA[i * 16 + l], sA[l * 16 + j], v);
      sB[e] = v;
      sp[QS_AP + k * 256 + e] = v;
    }
    wave_lds_fence();
    double* t = sA;
    sA = sB;
    sB = t;
  }
}

// symmetrise the filtered-covariance image in place?  No: P - K M' with M = P Z', K = M W is symmetric to rounding, and the tape of the
// in-kernel phase is no more symmetric than that.

__global__ __launch_bounds__(KF_WAVE) void kf_b4q_steady_fill_kernel(const KfFwdArgs A, const double* __restrict__ sp_all, int nslice) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, m = A.m, mm = m * m, n = A.T;
  double* imgF = smem;
  double* imgP = smem + mm + 2;
  const long items = A.B * (long)nslice;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const long b = it / nslice;
    const int sl = (int)(it - b * nslice);
    const int ts = A.hs_ts[b];
    if (ts >= n) continue;
    const double* sp = sp_all + b * (long)QS_TOTAL;
    wave_lds_fence();
    for (int x = lane; x <= mm; x += KF_WAVE) {
      imgF[x] = sp[QS_IMF + x];
      imgP[x] = sp[QS_IMP + x];
    }
    wave_lds_fence();
    const long cnt = (long)(n - ts) * mm;
    if (A.filt_P) {
      double* dst = A.filt_P + (b * (long)n + ts) * mm;
      const long a0 = (long)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15ull);
      const long lead = ((16 - a0) & 15) < cnt ? ((16 - a0) & 15) : cnt;
      const long nw = (cnt - lead + 127) / 128;
      fill_periodic(dst, cnt, imgF, mm, 0, lane, nw * sl / nslice, nw * (sl + 1) / nslice);
    }
    if (A.pred_P) {
      double* dst = A.pred_P + (b * (long)(n + 1) + ts + 1) * mm;
      const long a0 = (long)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15ull);
      const long lead = ((16 - a0) & 15) < cnt ? ((16 - a0) & 15) : cnt;
      const long nw = (cnt - lead + 127) / 128;
      fill_periodic(dst, cnt, imgP, mm, 0, lane, nw * sl / nslice, nw * (sl + 1) / nslice);
    }
  }
}

// rows of one block, [row][M] in LDS -> the draw's tape, 16 bytes per lane where the address allows
__device__ __forceinline__ void store_rows_q(double* dst, const double* rows, int count, int lane) {
  int lead = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & 1ull);
  if (lead > count) lead = count;
  if (lead && lane == 0) dst[0] = rows[0];
  for (int pos = lead + 2 * lane; pos < count; pos += 2 * KF_WAVE) {
    if (pos + 1 < count) {
      double2_t v = {rows[pos], rows[pos + 1]};
      *reinterpret_cast<double2_t*>(dst + pos) = v;
    } else {
      dst[pos] = rows[pos];
    }
  }
}

// MP >= m (rows computed per lane), PP >= p: compile-time loop bounds; the workspace matrices are zero-padded
template <int MP, int PP>
__global__ __launch_bounds__(KF_WAVE) void kf_b4q_steady_scan_kernel(const KfFwdArgs A, const double* __restrict__ sp_all) {
  constexpr int STEPS = KF_WAVE - 1;
  __shared__ __attribute__((aligned(16))) double sAP[6 * 256];
  __shared__ __attribute__((aligned(16))) double sV[128 + 128 + 128 + 64 + 64];  // G | K | Z | L^-1 | d^-1, c, a, d
  __shared__ __attribute__((aligned(16))) double rows[KF_WAVE * MP];  // one tape's rows at a time
  const int lane = threadIdx.x, n = A.T, m = A.m, p = A.p;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double joint_const = (A.filter == KF_FILTER_CHOLESKY) ? (double)p : 1.0;
  double* sG = sV;
  double* sK = sV + 128;
  double* sZ = sV + 256;
  double* sLi = sV + 384;
  double* sDi = sV + 448;      // 8
  double* sC = sV + 456;       // 16
  double* sA0 = sV + 472;      // 16
  double* sDv = sV + 488;      // 8
  double* sD = sV + 496;       // 8
  for (long b = blockIdx.x; b < A.B; b += gridDim.x) {
    const int ts = A.hs_ts[b];
    if (ts >= n) continue;
    const double* sp = sp_all + b * (long)QS_TOTAL;
    wave_lds_fence();
    for (int e = lane; e < 6 * 256; e += KF_WAVE) sAP[e] = sp[QS_AP + e];
    for (int e = lane; e < 128; e += KF_WAVE) {
      sG[e] = sp[QS_G + e];
      sK[e] = sp[QS_K + e];
      sZ[e] = sp[QS_Z + e];
    }
    sLi[lane] = sp[QS_LI + lane];
    if (lane < 8) {
      sDi[lane] = sp[QS_DI + lane];
      sDv[lane] = sp[QS_DV + lane];
      sD[lane] = sp[QS_D + lane];
    }
    if (lane < 16) {
      sC[lane] = sp[QS_C + lane];
      sA0[lane] = sp[QS_A + lane];
    }
    const double logdet = sp[QS_META + 0];
    const bool notpd = sp[QS_META + 1] != 0.0;
    const int ndeg = (int)sp[QS_META + 2];
    wave_lds_fence();
    const bool ypd = (A.flags & KF_FLAG_Y_PER_DRAW) != 0;
    const double* ybase = A.y + (ypd ? b * (long)n * p : 0L);
    double acur[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) acur[i] = sA0[i];
    double acc = 0.0;
    for (int tb = ts; tb < n; tb += STEPS) {
      const int t = tb + lane;
      const int cnt = (n - tb) < STEPS ? (n - tb) : STEPS;
      const bool act = lane < cnt;
      double yt[PP];
#pragma unroll
      for (int q = 0; q < PP; ++q) yt[q] = (t < n && q < p) ? ybase[(long)t * p + q] + dsign * sDv[q] : 0.0;
      double x[MP], xs[MP];
      {
        double yp[PP];
#pragma unroll
        for (int q = 0; q < PP; ++q) yp[q] = __shfl_up(yt[q], 1);
#pragma unroll
        for (int i = 0; i < MP; ++i) {
          double s = sC[i];
#pragma unroll
          for (int q = 0; q < PP; ++q) s = fma(sG[i * 8 + q], yp[q], s);
          x[i] = lane == 0 ? acur[i] : s;
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int s = 1 << k;
#pragma unroll
        for (int i = 0; i < MP; ++i) {
          const double up = __shfl_up(x[i], s);
          xs[i] = lane >= s ? up : 0.0;
        }
#pragma unroll
        for (int i = 0; i < MP; ++i) {
          double v = x[i];
#pragma unroll
          for (int l = 0; l < MP; ++l) v = fma(sAP[k * 256 + i * 16 + l], xs[l], v);
          x[i] = v;
        }
      }
      // lane j holds a_{tb+j}: innovations, filtered mean, the step's log-likelihood
      double v[PP];
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        double s = yt[q];
#pragma unroll
        for (int i = 0; i < MP; ++i) s = fma(-sZ[q * 16 + i], x[i], s);
        v[q] = s;
      }
      double af[MP];
#pragma unroll
      for (int i = 0; i < MP; ++i) {
        const double nx = __shfl_down(x[i], 1);
        double f2 = x[i];
#pragma unroll
        for (int q = 0; q < PP; ++q) f2 = fma(sK[i * 8 + q], v[q], f2);
        af[i] = f2;
        rows[lane * MP + i] = nx;
      }
      double qs = 0.0;
      int cnt_ll = 0;
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        double w = 0.0;
#pragma unroll
        for (int r2 = 0; r2 <= q; ++r2) w = fma(sLi[q * 8 + r2], v[r2], w);
        const double quad = w * w * sDi[q];
        qs += quad;
        if (q < p && sDi[q] != 0.0) cnt_ll += (sD[q] == 1.0 && quad == 0.0) ? 0 : 1;  // :497-500
      }
      double nconst;
      if (univariate) nconst = exact_const ? (double)p : (double)cnt_ll;
      else nconst = exact_const ? (double)p : joint_const;
      (void)ndeg;
      double llv = -0.5 * (logdet + qs + nconst * KF_LOG_2PI);
      if (notpd) llv = __builtin_nan("");
      if (act) {
        acc += llv;
        if (A.ll_obs) A.ll_obs[b * (long)n + t] = llv;
      }
      wave_lds_fence();
      if (A.pred_a) {
        double* dst = A.pred_a + (b * (long)(n + 1) + tb + 1) * m;
        if (MP == m) store_rows_q(dst, rows, cnt * m, lane);
        else
          for (int e = lane; e < cnt * m; e += KF_WAVE) dst[e] = rows[(e / m) * MP + e % m];
      }
      wave_lds_fence();
      if (A.filt_a) {
#pragma unroll
        for (int i = 0; i < MP; ++i) rows[lane * MP + i] = af[i];
        wave_lds_fence();
        double* dst = A.filt_a + (b * (long)n + tb) * m;
        if (MP == m) store_rows_q(dst, rows, cnt * m, lane);
        else
          for (int e = lane; e < cnt * m; e += KF_WAVE) dst[e] = rows[(e / m) * MP + e % m];
      }
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < MP; ++i) acur[i] = __shfl(x[i], KF_WAVE - 1);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
      A.logp[b] = A.hs_logp[b] + acc;
      int info = A.hs_info[b];
      if (notpd && info == 0) info = ((ts + 1) << 8) | 1;
      if (A.info) A.info[b] = info;
    }
  }
}

}  // namespace

// workspace of the split steady-state phase: hs_tile | hs_logp | sp | hs_info | hs_ts
size_t kf_b4q_split_workspace_bytes(long B) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return up(8 * (size_t)B * B4_HS_STRIDE) + up(8 * (size_t)B) + up(8 * (size_t)B * QS_TOTAL) + 2 * up(4 * (size_t)B) + 512;
}

void kf_b4q_split_carve(void* base, long B, KfFwdArgs* a, double** sp) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char* p = (char*)(((uintptr_t)base + 255) & ~(uintptr_t)255);
  a->hs_tile = (double*)p; p += up(8 * (size_t)B * B4_HS_STRIDE);
  a->hs_logp = (double*)p; p += up(8 * (size_t)B);
  *sp = (double*)p; p += up(8 * (size_t)B * QS_TOTAL);
  a->hs_info = (int32_t*)p; p += up(4 * (size_t)B);
  if (!a->hs_ts) a->hs_ts = (int32_t*)p;  // (kf_filter_value_and_grad passes its own ts array)
}

// behind one recursion launch (event ev_rec on the caller's stream): the constants on s_prep, the fill on s_fill and the scan on
// s_scan behind the constants.  Nothing joins here: the caller makes its stream wait for ev_fill / ev_scan of the LAST chunk (both
// streams run their launches in order)
int kf_b4q_split_steady(kf_handle* h, const KfFwdArgs& a, double* sp, hipStream_t s_prep, hipStream_t s_fill, hipStream_t s_scan,
                        hipEvent_t ev_rec, hipEvent_t ev_prep, hipEvent_t ev_fill, hipEvent_t ev_scan) {
  const int m = a.m, mm = m * m;
  // (the constants on their own stream: those of the next chunk are computed beside this chunk's fill)
  KF_HIP_CHECK(h, hipStreamWaitEvent(s_prep, ev_rec, 0));
  hipLaunchKernelGGL(kf_b4q_steady_prep_kernel, dim3((unsigned)a.B), dim3(KF_WAVE), 0, s_prep, a, sp);
  KF_HIP_CHECK(h, hipGetLastError());
  KF_HIP_CHECK(h, hipEventRecord(ev_prep, s_prep));
  KF_HIP_CHECK(h, hipStreamWaitEvent(s_fill, ev_prep, 0));
  if ((a.filt_P || a.pred_P) && !a.skip_steady_cov) {
    long ns = ((long)a.T * mm * 8 + 65535) / 65536;
    const int nslice = (int)(ns < 1 ? 1 : ns > 32 ? 32 : ns);
    hipLaunchKernelGGL(kf_b4q_steady_fill_kernel, dim3(2048), dim3(KF_WAVE), sizeof(double) * (size_t)(2 * mm + 4), s_fill, a, sp, nslice);
    KF_HIP_CHECK(h, hipGetLastError());
  }
  KF_HIP_CHECK(h, hipEventRecord(ev_fill, s_fill));
  KF_HIP_CHECK(h, hipStreamWaitEvent(s_scan, ev_prep, 0));
  const unsigned grid = (unsigned)(a.B < 4096 ? a.B : 4096);
  if (m <= 12 && a.p <= 5) hipLaunchKernelGGL((kf_b4q_steady_scan_kernel<12, 5>), dim3(grid), dim3(KF_WAVE), 0, s_scan, a, sp);
  else if (a.p <= 5) hipLaunchKernelGGL((kf_b4q_steady_scan_kernel<15, 5>), dim3(grid), dim3(KF_WAVE), 0, s_scan, a, sp);
  else hipLaunchKernelGGL((kf_b4q_steady_scan_kernel<15, 8>), dim3(grid), dim3(KF_WAVE), 0, s_scan, a, sp);
  KF_HIP_CHECK(h, hipGetLastError());
  KF_HIP_CHECK(h, hipEventRecord(ev_scan, s_scan));
  return KF_OK;
}
