// kf_forward_b4.hip -- forward Kalman filter on v_mfma_f64_4x4x4_4b_f64 (one wavefront per draw,
// scalar observation p == 1, k_states + 2 <= 16, static system matrices).
//
// Why a second matrix-core kernel: measured on MI355X (scripts/probes/f64_ops.hip, 4 wavefronts per SIMD)
// v_mfma_f64_16x16x4_f64 holds the fp64 pipe for 65 cycles, v_mfma_f64_4x4x4_4b_f64 (four independent
// 4x4x4 block products) for 10.2: 2.5 cycles per 4x4x4 block product against 4.1, and a 12 x 12 tile
// (the north-star shape: m = 10) is 3 x 3 blocks instead of a zero-padded 16 x 16 -- the two m^3
// products of a time step take 18 x 10.2 = 184 cycles instead of 6 x 65 = 390.
//
// Layout.  lane = 16 g + 4 b + q.  The instruction computes, per block b,
//     D_b[i][j] (lane g=i, q=j)  +=  sum_k A_b[i][k] (lane g=k, q=i) * B_b[k][j] (lane g=k, q=j)
// and with cbsz = 2, abid = I every block uses the A matrix of block I.  A symmetric N4 x N4 matrix X
// (N4 = 4 NB) lives in NB registers, register I holding X[4I+g][4b+q] ("row-block layout": slot b of
// register I is the 4 x 4 block X(I, b)).  Such a register is a valid B operand as it stands and,
// read as an A operand, the blockwise TRANSPOSE -- for symmetric X block (b, I), so that
//     W(I, :) = sum_K P(I, K) T'(K, :)   : mfma(A = P[K] abid = I, B = T'[K])      NB^2 instructions
//     S(I, :) = sum_K T/2(I, K) W(K, :)  : mfma(A = T'[K]/2 abid = I, B = W[K], C = RQR'/2)
// need no data movement at all between the two products.  P_hat = S + S' (kalman_filter.py:221) is one
// LDS transpose as in kf_forward_mfma.hip.
//
// The tile is the augmented second-moment matrix of (alpha, Z alpha, "1"):
//     index 0..m-1 : states          index JA = N4-2 : Z alpha (kf_forward_mfma.hip's AUG trick)
//     index JC = N4-1 : carrier of the MEAN -- row/column JC holds 2 a (and 2 Z a at JA), with
//     T_ext[JC][JC] = 1, C_ext[.][JC] = 2 c, so the predict products move the state vector along with
//     the covariance (a_hat = T a + c, kalman_filter.py:219) and no separate mean recursion exists.
// Update (kalman_filter.py:255-351, p = 1): with u = row JA of the tile, u_JC replaced by 2 Z a - 2 y
// (= -2 v) and F = H + u_JA, the whole update -- P_f = P - M M'/F, a_f = a + M v / F -- is ONE rank-one
// step  tile -= (u / sqrt F)(u / sqrt F)',  issued as NB more MFMAs (A = -u_hat, B = u_hat, k = 0 only);
// u / sqrt F keeps the tile bitwise symmetric.  Reference operation order differs (Joseph form); results
// agree to rounding (tests: 1e-8 on logp, 1e-9 on tapes).
#include <type_traits>

#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int TRS = 18;                 // transpose tile row stride (doubles): conflict-free 16-byte reads
constexpr int B4_OFF_STG = 16 * TRS;    // compact m x m tape image (<= 196 doubles) + 64 dummy slots
constexpr int B4_OFF_RG = B4_OFF_STG + 256 + 64;
constexpr int B4_OFF_YB = B4_OFF_RG + 128;
constexpr int B4_LOOP_DOUBLES = B4_OFF_YB + 64;
// setup scratch (aliases the loop buffers): T, P0, R, Q, Z, a0, c, R_ext, R_ext Q, Z T, P0 Z', 3 scalars
inline int b4_setup_doubles(int m, int r) { return 2 * m * m + m * r + r * r + 5 * m + 32 * r + 4; }

template <int ABID>
__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, ABID, 0);
}
template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// rows {2, 3} of the wavefront (lanes 32..63) copied over rows {0, 1}: v_permlane32_swap
__device__ __forceinline__ double high_half_everywhere(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[1], a[1]);
}

// 1/sqrt(x) to rounding accuracy: v_rsq_f64 + two Newton steps (x finite, normal, > 0)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  r = fma(r, fma(-hx * r, r, 0.5), r);
  r = fma(r, fma(-hx * r, r, 0.5), r);
  return r;
}

template <int NB>
__global__ __launch_bounds__(KF_WAVE) void kf_fwd_b4_kernel(const KfFwdArgs A) {
  constexpr int N4 = 4 * NB, JA = N4 - 2, JC = N4 - 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15;       // bq = 4 b + q: column index of the lane
  const int fb = (lane & 3) * 4 + ((lane >> 2) & 3);  // 4 q + b: position of column bq in a transpose-tile row
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, mm = m * m;

  double* TR = smem;
  double* STG = smem + B4_OFF_STG;
  double* RG = smem + B4_OFF_RG;
  double* YB = smem + B4_OFF_YB;

  // ---- per-draw constants through an LDS scratch that aliases the loop buffers ----------------------------
  double Tt[NB], Th[NB], Ch[NB], P[NB];
  unsigned stg_off[NB];  // LDS byte offset of this lane's element in the compact tape image (dummy slot if none)
  const double* gP = A.P0 + b * (long)mm;
  {
    double* sT = smem;                 // T    m x m
    double* sP = sT + mm;              // P0   m x m (as given; the upper triangle is used)
    double* sR = sP + mm;              // R    m x r
    double* sQ = sR + m * r_;          // Q    r x r
    double* sZ = sQ + r_ * r_;         // Z    m
    double* sa = sZ + m;               // a0
    double* sc = sa + m;               // c
    double* sZT = sc + m;              // Z T
    double* sPZ = sZT + m;             // P0 Z'
    double* sS = sPZ + m;              // [0] Z P0 Z'   [1] Z a0   [2] Z c
    double* sRe = sS + 4;              // R_ext   16 x r  rows: R | 0 | Z R (row JA)
    double* sRQ = sRe + 16 * r_;       // R_ext Q
    {
      const double* gT = A.Tm + b * (long)mm;
      const double* gR = A.R + b * (long)m * r_;
      const double* gQ = A.Q + b * (long)r_ * r_;
      const double* gZ = A.Z + b * (long)m;
      for (int e = lane; e < mm; e += KF_WAVE) {
        sT[e] = gT[e];
        sP[e] = gP[e];
      }
      for (int e = lane; e < m * r_; e += KF_WAVE) sR[e] = gR[e];
      for (int e = lane; e < r_ * r_; e += KF_WAVE) sQ[e] = gQ[e];
      if (lane < m) {
        sZ[lane] = gZ[lane];
        sa[lane] = A.a0[b * (long)m + lane];
        sc[lane] = A.c ? A.c[b * (long)m + lane] : 0.0;
      }
    }
    __syncthreads();
    auto P0s = [&](int i, int j) { return (j >= i) ? sP[i * m + j] : sP[j * m + i]; };
    for (int e = lane; e < 16 * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      if (i < m) {
        v = sR[e];
      } else if (i == JA) {
        for (int j = 0; j < m; ++j) v += sZ[j] * sR[j * r_ + l];
      }
      sRe[e] = v;
    }
    if (lane < m) {
      double tz = 0.0, pz = 0.0;
      for (int l = 0; l < m; ++l) {
        tz += sZ[l] * sT[l * m + lane];
        pz += P0s(lane, l) * sZ[l];
      }
      sZT[lane] = tz;
      sPZ[lane] = pz;
    }
    __syncthreads();
    for (int e = lane; e < 16 * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      for (int kk = 0; kk < r_; ++kk) v += sRe[i * r_ + kk] * sQ[kk * r_ + l];
      sRQ[e] = v;
    }
    if (lane < 3) {
      double v = 0.0;
      for (int l = 0; l < m; ++l) v += sZ[l] * (lane == 0 ? sPZ[l] : lane == 1 ? sa[l] : sc[l]);
      sS[lane] = v;
    }
    __syncthreads();
    const int cc = bq;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g;
      // T_ext'(row-block register I): element [rr][cc] of T_ext' = T_ext[cc][rr]
      double tv = 0.0;
      if (rr < m) tv = (cc < m) ? sT[cc * m + rr] : (cc == JA ? sZT[rr] : 0.0);
      else if (rr == JC && cc == JC) tv = 1.0;
      Tt[I] = tv;
      Th[I] = 0.5 * tv;
      // P0_ext[rr][cc]
      double pv = 0.0;
      if (rr < m) pv = (cc < m) ? P0s(rr, cc) : (cc == JA ? sPZ[rr] : (cc == JC ? 2.0 * sa[rr] : 0.0));
      else if (rr == JA) pv = (cc < m) ? sPZ[cc] : (cc == JA ? sS[0] : (cc == JC ? 2.0 * sS[1] : 0.0));
      else if (rr == JC) pv = (cc < m) ? 2.0 * sa[cc] : (cc == JA ? 2.0 * sS[1] : 0.0);
      P[I] = pv;
      // C_ext[rr][cc] / 2: R_ext Q R_ext' on states + JA, the intercept 2 c on the carrier row / column
      double cv = 0.0;
      const bool r_cov = rr < m || rr == JA, c_cov = cc < m || cc == JA;
      if (r_cov && c_cov) {
        for (int l = 0; l < r_; ++l) cv += sRQ[rr * r_ + l] * sRe[cc * r_ + l];
      } else if (rr == JC && cc != JC && c_cov) {
        cv = 2.0 * (cc < m ? sc[cc] : sS[2]);
      } else if (cc == JC && rr != JC && r_cov) {
        cv = 2.0 * (rr < m ? sc[rr] : sS[2]);
      }
      Ch[I] = 0.5 * cv;
      stg_off[I] = (unsigned)(((rr < m && cc < m) ? rr * m + cc : 256 + lane) * 8);
    }
    __syncthreads();  // scratch is dead: the loop buffers may be written
  }
  // C_ext must be bitwise symmetric (both mirrored entries are the same sum in a different order otherwise):
  // transpose once through the tile and average -- also zeroes the tile rows no register ever writes
  for (int e = lane; e < 16 * TRS; e += KF_WAVE) TR[e] = 0.0;
  __syncthreads();
#pragma unroll
  for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = Ch[I];
  __syncthreads();
#pragma unroll
  for (int I = 0; I < NB; ++I) Ch[I] = 0.5 * (Ch[I] + TR[bq * TRS + 4 * g + I]);
  __syncthreads();

  const double H1 = A.H[b];
  const double d1 = A.d ? A.d[b] : 0.0;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;

  const double mask0 = (g == 0 && bq < N4) ? 1.0 : 0.0;  // u lives in DPP row 0 (k = 0 of the rank-one MFMA)
  const double adjm = (lane == JC) ? -2.0 : 0.0;
  const bool vec_on = g == 3 && bq < m;                  // row JC of the tile = 2 a'
  const unsigned voff = (unsigned)(bq * 8);
  double* o_fa = A.filt_a ? A.filt_a + b * (long)n * m : nullptr;
  double* o_pa = A.pred_a ? A.pred_a + b * (long)(n + 1) * m : nullptr;
  double* o_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* o_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;
  auto st_off = [](double* base, unsigned byte_off, double v) {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off) = v;
  };

  // predicted tape slot 0 = (a0, P0 as given)
  if (o_pa) {
    if (vec_on) st_off(o_pa, voff, 0.5 * P[NB - 1]);
    o_pa += m;
  }
  if (o_pP) {
    for (int e = lane; e < mm; e += KF_WAVE) o_pP[e] = gP[e];
    o_pP += mm;
  }

  // covariance tapes: registers -> compact m x m image in LDS -> 16 bytes per lane, full lines
  const int nb128 = mm >> 1;
  auto stage_cov = [&](const double (&Preg)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
      *reinterpret_cast<double*>(reinterpret_cast<char*>(STG) + stg_off[I]) = Preg[I];
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

  double logp = 0.0;
  int info = 0;
  // ll of the steps [tb, tb+cnt) held in the ring, one step per lane: (F, v^2/F); F = -1 missing, -2 skipped
  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
    const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * lane);
    const bool act = lane < cnt;
    const double Fv = act ? fq.x : -1.0;
    double llv;
    if (Fv == -1.0) llv = 0.0;
    else if (Fv == -2.0) llv = exact_const ? -0.5 * KF_LOG_2PI : 0.0;
    else if (univariate && !exact_const && Fv == 1.0 && fq.y == 0.0) llv = 0.0;  // :497-500
    else llv = -0.5 * (KF_LOG_2PI + log(Fv) + fq.y);
    if (act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
    __syncthreads();
  };

  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);

  for (int t = 0; t < n; ++t) {
    if ((t & 63) == 0) {
      const int cnt = (n - t) < 64 ? (n - t) : 64;
      if (lane < cnt) YB[lane] = ybase[(long)t + lane];
      wave_lds_fence();
    }
    const double yq = YB[t & 63];
    const bool obs = !(yq != yq);
    const double yd = obs ? (yq + dsign * d1) : 0.0;
    // ---- update: u = row JA of the tile (DPP row 2 of the last register) brought to DPP row 0 ---------------
    const double raw = high_half_everywhere(P[NB - 1]);
    const double u = fma(raw, mask0, adjm * yd);  // u_JC = 2 Z a - 2 (y -+ d) = -2 v
    const double F = H1 + readlane_f64(u, JA);
    const bool degenerate = !obs || (univariate && F == 0.0);  // missing -> skipped; Univariate skips F == 0 (:468-471)
    const bool notpd = !degenerate && !(F > 0.0);
    if (notpd && info == 0) info = ((t + 1) << 8) | 1;
    const double rs = degenerate ? 0.0 : fast_rsqrt(F);
    const double uh = u * rs, un = u * (-rs);
    if (lane == JC) {
      double2_t fq = {degenerate ? (obs ? -2.0 : -1.0) : F, 0.25 * uh * uh};
      *reinterpret_cast<double2_t*>(RG + 2 * (t & 63)) = fq;
    }
    sfor<0, NB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      P[I] = mfma4<I>(un, uh, P[I]);
    });
    if (pred_pending) {  // predicted covariance of the previous step: tile image -> HBM
      flush_cov(o_pP);
      o_pP += mm;
      pred_pending = false;
    }
    // ---- filtered tapes -----------------------------------------------------------------------------------------------
    if (o_fa) {
      if (vec_on) st_off(o_fa, voff, 0.5 * P[NB - 1]);
      o_fa += m;
    }
    if (o_fP) stage_cov(P);  // flushed below, behind the MFMAs
    // ---- predict: tile <- sym(T_ext tile T_ext' + C_ext) ---------------------------------------------------------------------
    double W[NB], S[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      W[I] = 0.0;
      S[I] = Ch[I];
    }
#pragma unroll
    for (int K = 0; K < NB; ++K)
      sfor<0, NB>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        W[I] = mfma4<I>(P[K], Tt[K], W[I]);
      });
#pragma unroll
    for (int K = 0; K < NB; ++K)
      sfor<0, NB>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        S[I] = mfma4<I>(Th[K], W[K], S[I]);
      });
    if (o_fP) {
      flush_cov(o_fP);
      o_fP += mm;
    }
    // transpose S/2 through LDS: element (i, j) at TR[i * TRS + 4 (j & 3) + (j >> 2)]
#pragma unroll
    for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = S[I];
    wave_lds_fence();
    {
      const double* src = TR + bq * TRS + 4 * g;
      const double2_t t01 = *reinterpret_cast<const double2_t*>(src);
      P[0] = S[0] + t01.x;
      if (NB > 1) P[NB > 1 ? 1 : 0] = S[NB > 1 ? 1 : 0] + t01.y;
      if (NB > 2) {
        const double2_t t23 = *reinterpret_cast<const double2_t*>(src + 2);
        P[NB > 2 ? 2 : 0] = S[NB > 2 ? 2 : 0] + t23.x;
        if (NB > 3) P[NB > 3 ? 3 : 0] = S[NB > 3 ? 3 : 0] + t23.y;
      }
    }
    wave_lds_fence();
    if (o_pa) {
      if (vec_on) st_off(o_pa, voff, 0.5 * P[NB - 1]);
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

bool kf_forward_b4_ok(int m, int p, int r) { return p == 1 && m >= 1 && m + 2 <= 16 && r >= 1 && r <= 64; }

int kf_launch_forward_b4(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  if (!kf_forward_b4_ok(a.m, a.p, a.r) || (a.flags & KF_FLAG_TV_ANY)) {
    KF_SET_ERR(h, "forward(b4): needs p == 1, m <= 14, static matrices (got p=%d m=%d r=%d)", a.p, a.m, a.r);
    return KF_ERR_UNSUPPORTED;
  }
  const int lds_setup = b4_setup_doubles(a.m, a.r);
  const size_t lds = sizeof(double) * (size_t)(B4_LOOP_DOUBLES > lds_setup ? B4_LOOP_DOUBLES : lds_setup);
  const int nb = (a.m + 2 + 3) / 4;
#define KF_B4_LAUNCH(NB_) \
  hipLaunchKernelGGL((kf_fwd_b4_kernel<NB_>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a)
  switch (nb) {
    case 1: KF_B4_LAUNCH(1); break;
    case 2: KF_B4_LAUNCH(2); break;
    case 3: KF_B4_LAUNCH(3); break;
    default: KF_B4_LAUNCH(4); break;
  }
#undef KF_B4_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
