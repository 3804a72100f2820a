// kf_smoother_b4_steady.hip -- the RTS smoother behind the point where the FILTER has converged, time-parallel (round 4).
//
// Reference step (filters/kalman_smoother.py:87-97), static T, R, Q:
//     P_hat = T P_f T' + R Q R'      G = P_f T' P_hat^-1      a_s[t] = a_f[t] + G (a_s[t+1] - T a_f[t])
//     P_s[t] = P_f + G (P_s[t+1] - P_hat) G'
// kf_smooth_b4_kernel (kf_smoother_b4.hip) walks every step: six block products and an iterated inverse, 2600 shader cycles per
// draw-step at four wavefronts per SIMD -- 4.6 ms at the headline shape, 0.195 of the HBM peak, throughput-bound on the SUM of its LDS,
// MFMA and VALU work.  But the filter's covariance converges (kalman_filter.py:215-252 repeats one update from then on;
// median step 335 of 1000 on the headline's structural draws): from that step to the end of the series the tape holds ONE
// P_f -- bit for bit behind a steady-state forward (C3, C4), within the forward kernels' own convergence tolerance behind a recursion at
// every step -- and with it P_hat, G are constants of the draw.  Then
//   * the smoothed MEAN is an affine recursion backwards in time, a_s[t] = (I - G T) a_f[t] + G a_s[t+1]: a prefix scan, one step per lane,
//     64 steps per pass, six Hillis-Steele levels with G^(2^k) read from LDS as broadcasts (kf_bwd_steady_scan_kernel's scan);
//   * the smoothed COVARIANCE contracts onto its own fixpoint, P_s[t] - P_inf = G (P_s[t+1] - P_inf) G': it is recursed (two small
//     products per step, in LDS) until a step moves no entry by more than 1 % of the tape tolerance below, and is a CONSTANT below that
//     step -- written by a fill kernel at the HBM rate (fill_periodic, kf_b4_split.h).
// kf_smooth_b4_steady_kernel does this from the last step down to start[b], the first step whose filtered covariance is still P_f[n-1]
// within tolerance -- it READS every filtered covariance of that range to verify it (a missing observation in the middle of the series
// ends the range there) -- and kf_smooth_b4_kernel takes the steps below start[b] from (a_s, P_s)[start[b]] as before.  P_hat^-1 once
// per draw by Cholesky (a pivot below 1e-2 of the largest diagonal entry: the draw is left to kf_smooth_b4_kernel and its fallbacks).
//
// Tolerance: |P_f[t] - P_f[n-1]|_ij <= 1e-14 max|P_f| AND <= 1e-12 sqrt(P_ii P_jj) (KF_SMOOTHER_SPLIT_TOL, measurement builds only -- kf_env_experiment --: the
// first factor).  1e-15, the forward kernels' one-step test, is below the rounding jitter of a converged recursion relative to its
// LAST value and accepted a median of four steps.  Against every step in full the smoothed moments move by the tolerance times the
// conditioning of P_hat -- measured (tests/test_gpu_r04.py, scripts/r04/smooth_split_check.py): <= 6e-13 of a draw's largest entry
// on the BASELINE workloads at 1e-14, 1.3e-12 at 1e-12.  KF_FLAG_EXACT_RECURSION / KF_SMOOTHER_SPLIT=0: every step in
// kf_smooth_b4_kernel.  Used by default for series of >= 1536 steps (kf_smoother_b4_split_ok below).
#include <stdlib.h>

#include "kf_b4.h"
#include "kf_b4_split.h"

namespace {

// ---- which steps qualify: a guess by bisection, then every covariance of the guessed range compared -------------------------------------
// (entry e of the last filtered covariance, its tolerance) for this lane's entries e = lane, lane + 64, ...: the rule of the header
template <int NQ>
__device__ __forceinline__ void ref_and_tol(const double* __restrict__ pl, int m, int mm, int lane, double tol_rel, double (&ref)[NQ],
                                            double (&tol)[NQ]) {
  unsigned pm = 0u;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = lane + q * KF_WAVE;
    ref[q] = e < mm ? pl[e] : 0.0;
    pm = max(pm, abs_hi(ref[q]));
  }
  const double pmax = wave_absmax_hi(pm);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = lane + q * KF_WAVE;
    const int i = e < mm ? e / m : 0, j = e < mm ? e - i * m : 0;
    tol[q] = fmin(tol_rel * pmax, 1e-12 * sqrt(fabs(pl[i * m + i] * pl[j * m + j])));   // (NaN: nothing ever matches)
  }
}

// guess[b] = the first step from which the filtered covariance matches the last one, by bisection (a converged filter stays converged
// unless an observation is missing later: kf_smooth_b4_verify_kernel below finds such steps); tbad[b] = -1
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_b4_guess_kernel(const KfSmoothArgs A, int32_t* __restrict__ guess,
                                                                      int32_t* __restrict__ tbad, const double tol_rel) {
  constexpr int NQ = 2;   // m <= 10
  const long b = blockIdx.x;
  const int lane = threadIdx.x, m = A.m, mm = m * m, n = A.T;
  const double* gfP = A.filt_P + b * (long)n * mm;
  double ref[NQ], tol[NQ];
  ref_and_tol<NQ>(gfP + (long)(n - 1) * mm, m, mm, lane, tol_rel, ref, tol);
  int lo = 0, hi = n - 1;   // invariant: step lo - 1 does not match (or lo == 0), step hi does
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    bool off = false;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = lane + q * KF_WAVE;
      if (e < mm) off = off || !(fabs(gfP[(long)mid * mm + e] - ref[q]) <= tol[q]);
    }
    if (__ballot(off) != 0ull) lo = mid + 1;
    else hi = mid;
  }
  if (lane == 0) {
    guess[b] = lo;
    tbad[b] = -1;
  }
}

// one wavefront per (64-step block, draw): the block's filtered covariances against the last one; tbad[b] = the latest step that differs.
// (reference, tolerance) pairs in LDS, one ds_read_b128 per element; M a template parameter: the element -> entry index is a
// multiply-shift (the first version, runtime m and three cached loads per element, ran at 2 TB/s)
template <int M>
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_b4_verify_kernel(const KfSmoothArgs A, const int32_t* __restrict__ guess,
                                                                       int32_t* __restrict__ tbad, const double tol_rel) {
  constexpr int MM = M * M, NQ = (MM + KF_WAVE - 1) / KF_WAVE;
  __shared__ __attribute__((aligned(16))) double sRef[2 * MM];
  const long b = blockIdx.y;
  const int lane = threadIdx.x, n = A.T;
  const int te = n - 1 - KF_WAVE * (int)blockIdx.x;   // steps [tb, te), as kf_smooth_b4_steady_kernel's blocks
  const int gs = guess[b];
  if (te <= gs) return;
  const int tb = te > KF_WAVE ? te - KF_WAVE : 0;
  const int lo = tb > gs ? tb : gs;
  const double* gfP = A.filt_P + b * (long)n * MM;
  double ref[NQ], tol[NQ];
  ref_and_tol<NQ>(gfP + (long)(n - 1) * MM, M, MM, lane, tol_rel, ref, tol);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (lane + q * KF_WAVE < MM) {
      sRef[2 * (lane + q * KF_WAVE)] = ref[q];
      sRef[2 * (lane + q * KF_WAVE) + 1] = tol[q];
    }
  wave_lds_fence();
  const double* blk = gfP + (long)lo * MM;
  const int cnt = (te - lo) * MM;
  int bad = -1;
  constexpr int UNR = 10;
  for (int e0 = lane; e0 < cnt; e0 += UNR * KF_WAVE) {
    double v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * KF_WAVE;
      v[u] = e < cnt ? blk[e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * KF_WAVE;
      const int tt = e / MM, idx = e - tt * MM;
      const double2_t rt = *reinterpret_cast<const double2_t*>(sRef + 2 * idx);
      if (e < cnt && !(fabs(v[u] - rt.x) <= rt.y)) bad = max(bad, lo + tt);
    }
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) bad = max(bad, __shfl_xor(bad, off));
  if (lane == 0 && bad >= 0) atomicMax(tbad + b, bad);
}

// Cholesky pivots below this fraction of P_hat's largest diagonal entry: the draw stays with kf_smooth_b4_kernel.  The split treats
// filtered covariances within 1e-14 of the last one as equal, an error the inverse amplifies by cond(P_hat).  Measured over the fuzz
// sweep (scripts/r04/fuzz_r04b.py, largest deviation from every step in full, relative to a draw's largest entry): pivot rule 1e-9
// (the elimination kernels') 8e-4 on rank-one state noise, 1e-4 4e-9, 1e-3 2e-11, 1e-2 3e-13 -- with 33 of the 39 cases that took the
// split at 1e-4 still taking it, and C3 (structural draws: well conditioned) unchanged at 13.0 ms.
#ifndef KF_SPLIT_PIVOT
#define KF_SPLIT_PIVOT 1e-2
#endif
template <int M>
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_b4_steady_kernel(const KfSmoothArgs A, int32_t* __restrict__ start,
                                                                       int32_t* __restrict__ fill_rng, double* __restrict__ Pinf,
                                                                       const double tol_rel, const int32_t* __restrict__ guess,
                                                                       const int32_t* __restrict__ tbad_all) {
  constexpr int MM = M * M;
  __shared__ __attribute__((aligned(16))) double sG[6 * MM + 2];   // G^(2^k), row-major
  __shared__ __attribute__((aligned(16))) double sRef[2 * MM];     // (P_f[n-1], tolerance) entry by entry
  __shared__ __attribute__((aligned(16))) double sT[MM], sP[MM], sU[MM], sPh[MM], sL[MM], sX[MM], sW[MM], sPs[MM], sE[MM];
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r;
  const double* gT = A.Tm + b * (long)MM;
  const double* gR = A.R + b * (long)M * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  const double* gfa = A.filt_a + b * (long)n * M;
  const double* gfP = A.filt_P + b * (long)n * MM;
  double* gsa = A.smooth_a + b * (long)n * M;
  double* gsP = A.smooth_P + b * (long)n * MM;
  auto leave = [&]() {  // nothing done: kf_smooth_b4_kernel takes the whole series
    if (lane == 0) {
      start[b] = n;
      fill_rng[2 * b] = 0;
      fill_rng[2 * b + 1] = 0;
    }
  };
  // ---- constants of the draw from the LAST filtered covariance ---------------------------------------------------------------------
  unsigned pm = 0u;
  for (int e = lane; e < MM; e += KF_WAVE) {
    const double v = gfP[(long)(n - 1) * MM + e];
    sP[e] = v;
    sT[e] = gT[e];
    pm = max(pm, abs_hi(v));
    const int i = e / M, j = e - i * M;
    double cv = 0.0;
    for (int l = 0; l < r_; ++l) {
      double rq = 0.0;
      for (int k = 0; k < r_; ++k) rq += gR[i * r_ + k] * gQ[k * r_ + l];
      cv += rq * gR[j * r_ + l];
    }
    sE[e] = cv;   // R Q R' (sE is free until the covariance recursion)
  }
  const double pmax = wave_absmax_hi(pm);   // NaN / Inf come out as such
  wave_lds_fence();
  if (!(pmax < 1e300)) {
    leave();
    return;
  }
  for (int e = lane; e < MM; e += KF_WAVE) {
    const int i = e / M, j = e - i * M;
    double u = 0.0;
#pragma unroll
    for (int l = 0; l < M; ++l) u = fma(sT[i * M + l], sP[l * M + j], u);
    sU[e] = u;
    const double sc = 1e-12 * sqrt(fabs(sP[i * M + i] * sP[j * M + j]));
    sRef[2 * e] = sP[e];
    sRef[2 * e + 1] = fmin(tol_rel * pmax, sc);
  }
  wave_lds_fence();
  for (int e = lane; e < MM; e += KF_WAVE) {
    const int i = e / M, j = e - i * M;
    double s1 = sE[e], s2 = sE[j * M + i];
#pragma unroll
    for (int l = 0; l < M; ++l) {
      s1 = fma(sU[i * M + l], sT[j * M + l], s1);
      s2 = fma(sU[j * M + l], sT[i * M + l], s2);
    }
    sPh[e] = 0.5 * (s1 + s2);
  }
  wave_lds_fence();
  // ---- Cholesky P_hat = L L' (lane i owns row i), then P_hat X = U column by column (lane c owns column c) -------------------------------
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) dmax = fmax(dmax, sPh[i * M + i]);
  bool spd = dmax > 0.0;
#pragma unroll 1
  for (int j = 0; j < M; ++j) {
    const int i = lane < M ? lane : 0;
    double s = sPh[i * M + j];
    for (int k = 0; k < j; ++k) s = fma(-sL[i * M + k], sL[j * M + k], s);
    const double d = __shfl(s, j);
    if (!(d > KF_SPLIT_PIVOT * dmax)) spd = false;
    const double inv = 1.0 / sqrt(spd ? d : 1.0);
    wave_lds_fence();
    if (lane < M && lane >= j) sL[lane * M + j] = s * inv;   // (row j, column j: d / sqrt(d))
    if (lane < j) sL[lane * M + j] = 0.0;
    wave_lds_fence();
  }
  if (!spd) {
    leave();
    return;
  }
  {
    const int c = lane < M ? lane : 0;
    double yv[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = sU[i * M + c];
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(-sL[i * M + k], yv[k], s);
      yv[i] = s / sL[i * M + i];
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      double s = yv[i];
#pragma unroll
      for (int k = i + 1; k < M; ++k) s = fma(-sL[k * M + i], yv[k], s);
      yv[i] = s / sL[i * M + i];
    }
    wave_lds_fence();
    if (lane < M) {
#pragma unroll
      for (int i = 0; i < M; ++i) sX[i * M + c] = yv[i];   // X_e = P_hat^-1 T P_f
    }
  }
  wave_lds_fence();
  for (int e = lane; e < MM; e += KF_WAVE) {
    const int i = e / M, j = e - i * M;
    sG[e] = sX[j * M + i];   // G = X_e'
  }
  wave_lds_fence();
  for (int e = lane; e < MM; e += KF_WAVE) {   // W = I - G T
    const int i = e / M, j = e - i * M;
    double s = i == j ? 1.0 : 0.0;
#pragma unroll
    for (int l = 0; l < M; ++l) s = fma(-sG[i * M + l], sT[l * M + j], s);
    sW[e] = s;
  }
  wave_lds_fence();
#pragma unroll 1
  for (int k = 1; k < 6; ++k) {
    const double* src = sG + (k - 1) * MM;
    for (int x = lane; x < MM; x += KF_WAVE) {
      const int i = x / M, j = x - i * M;
      double v = 0.0;
#pragma unroll
      for (int l = 0; l < M; ++l) v = fma(src[i * M + l], src[l * M + j], v);
      sG[k * MM + x] = v;
    }
    wave_lds_fence();
  }
  // ---- the last step: smoothed = filtered ----------------------------------------------------------------------------------------
  for (int e = lane; e < MM; e += KF_WAVE) gsP[(long)(n - 1) * MM + e] = sP[e];
  double carry[M];
#pragma unroll
  for (int i = 0; i < M; ++i) carry[i] = gfa[(long)(n - 1) * M + i];
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < M; ++i) gsa[(long)(n - 1) * M + i] = carry[i];
  }
  // ---- blocks of 64 steps, backwards: lane j <-> step te - 1 - j ---------------------------------------------------------------------
  // covariance recursion: this lane's entries
  constexpr int NQ = (MM + KF_WAVE - 1) / KF_WAVE, NT = M * (M + 1) / 2;
  static_assert(NT <= KF_WAVE, "one upper-triangle entry per lane");
  double gE[NQ][M], gP[M];
  int ej[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = (lane + q * KF_WAVE) < MM ? lane + q * KF_WAVE : 0;
    const int i = e / M;
    ej[q] = e - i * M;
#pragma unroll
    for (int l = 0; l < M; ++l) gE[q][l] = sG[i * M + l];
  }
  const bool tri = lane < NT;
  int ti = 0, tj = 0;
  {
    int u = tri ? lane : 0;
    while (u >= M - ti) {
      u -= M - ti;
      ++ti;
    }
    tj = ti + u;
  }
#pragma unroll
  for (int l = 0; l < M; ++l) gP[l] = sG[tj * M + l];
  const double pf_own = sP[ti * M + tj], ph_own = sPh[ti * M + tj], tol_own = 0.01 * sRef[2 * (ti * M + tj) + 1];
  double ps_own = pf_own;   // P_s[n-1] = P_f[n-1]
  double* sD = sPs;         // D = P_s - P_hat
  for (int e = lane; e < MM; e += KF_WAVE) sD[e] = sP[e] - sPh[e];
  wave_lds_fence();
  const int t_lim = max(guess[b], tbad_all[b] + 1);
  bool conv = false;   // the covariance recursion has stopped moving: P_s[t] = its last value for every t <= t_c
  int t_c = n - 1, st = n - 1;
  for (int te = n - 1; te > 0; te -= KF_WAVE) {
    const int tb = te > KF_WAVE ? te - KF_WAVE : 0;
    const int cnt = te - tb;
    // (1) the steps of the block whose filtered covariance is P_f[n-1] within tolerance: those from t_lim on (kf_smooth_b4_guess_kernel
    // / _verify_kernel: every covariance from the guess on has been compared)
    const int tbad = t_lim > tb ? t_lim - 1 : -1;
    const int nvalid = tbad < 0 ? cnt : te - 1 - tbad;   // lanes 0 .. nvalid-1 hold steps above the mismatch
    if (nvalid <= 0) break;
    // (2) means
    {
      const bool act = lane < cnt;
      const int t = act ? te - 1 - lane : tb;
      double a[M], x[M], xs[M];
#pragma unroll
      for (int i = 0; i < M; ++i) a[i] = gfa[(long)t * M + i];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double w = 0.0;
#pragma unroll
        for (int l = 0; l < M; ++l) w = fma(sW[i * M + l], a[l], w);
        if (lane == 0) {
#pragma unroll
          for (int l = 0; l < M; ++l) w = fma(sG[i * M + l], carry[l], w);
        }
        x[i] = act ? w : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int s = 1 << k;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          const double up = __shfl_up(x[i], s);
          xs[i] = lane >= s ? up : 0.0;
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
          double vv = x[i];
#pragma unroll
          for (int l = 0; l < M; ++l) vv = fma(sG[k * MM + i * M + l], xs[l], vv);
          x[i] = vv;
        }
      }
      if (lane < nvalid) {
#pragma unroll
        for (int i = 0; i < M; ++i) gsa[(long)t * M + i] = x[i];
      }
#pragma unroll
      for (int i = 0; i < M; ++i) carry[i] = __shfl(x[i], nvalid - 1);
    }
    // (3) covariances: P_s[t] = P_f + G D G', D = P_s[t+1] - P_hat (kept symmetric in LDS), until a step moves no entry by more than 1 %
    // of the tape tolerance.  E = G D: two entries per lane, row i of G in registers, row j of D (= its column) contiguous in LDS;
    // then the upper triangle, one entry per lane: row j of G in registers, row i of E contiguous; both halves written from there
    if (!conv) {
      for (int t = te - 1; t >= te - nvalid; --t) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (lane + q * KF_WAVE < MM) {
            double sacc = 0.0;
#pragma unroll
            for (int l = 0; l < M; ++l) sacc = fma(gE[q][l], sD[ej[q] * M + l], sacc);
            sE[lane + q * KF_WAVE] = sacc;
          }
        }
        wave_lds_fence();
        double pn = 0.0;
        if (tri) {
#pragma unroll
          for (int l = 0; l < M; ++l) pn = fma(sE[ti * M + l], gP[l], pn);
          pn += pf_own;
        }
        const bool moved = tri && !(fabs(pn - ps_own) <= tol_own);
        ps_own = pn;
        wave_lds_fence();
        if (tri) {
          const double dn = pn - ph_own;
          sD[ti * M + tj] = dn;
          sD[tj * M + ti] = dn;
          gsP[(long)t * MM + ti * M + tj] = pn;
          gsP[(long)t * MM + tj * M + ti] = pn;
        }
        wave_lds_fence();
        t_c = t;
        if (__ballot(moved) == 0ull) {
          conv = true;
          break;
        }
      }
    }
    st = te - nvalid;
    if (tbad >= 0) break;
  }
  // steps st .. n-1 are done, except the covariances of [st, t_c) (the constant sPs: kf_smooth_b4_fill_kernel).  A recursion that has
  // not reproduced itself by st leaves nothing to fill (t_c == st)
  const bool fill = conv && st < t_c;
  if (tri) {   // (the last recursed value itself, from the registers that hold it)
    Pinf[b * (long)MM + ti * M + tj] = ps_own;
    Pinf[b * (long)MM + tj * M + ti] = ps_own;
    if (fill) {   // the hand-over step itself: kf_smooth_b4_kernel starts from it
      gsP[(long)st * MM + ti * M + tj] = ps_own;
      gsP[(long)st * MM + tj * M + ti] = ps_own;
    }
  }
  if (lane == 0) {
    start[b] = st;
    fill_rng[2 * b] = fill ? st + 1 : 0;
    fill_rng[2 * b + 1] = fill ? t_c : 0;
  }
}

// smooth_P[b][lo .. hi) = Pinf[b]: (draw, slice) work items, whole 128-byte lines, 16 bytes per lane
__global__ __launch_bounds__(KF_WAVE) void kf_smooth_b4_fill_kernel(const KfSmoothArgs A, const int32_t* __restrict__ fill_rng,
                                                                     const double* __restrict__ Pinf, int nslice) {
  extern __shared__ __attribute__((aligned(16))) double img[];
  const int lane = threadIdx.x, mm = A.m * A.m;
  const long b = blockIdx.x / nslice;
  const int sl = blockIdx.x % nslice;
  const int lo = fill_rng[2 * b], hi = fill_rng[2 * b + 1];
  if (hi <= lo) return;
  for (int x = lane; x <= mm; x += KF_WAVE) img[x] = Pinf[b * (long)mm + (x == mm ? 0 : x)];
  wave_lds_fence();
  double* dst = A.smooth_P + (b * (long)A.T + lo) * mm;
  const long cnt = (long)(hi - lo) * mm;
  const long a0 = (long)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15ull);
  const long lead = ((16 - a0) & 15) < cnt ? ((16 - a0) & 15) : cnt;
  const long nw = (cnt - lead + 127) / 128;
  fill_periodic(dst, cnt, img, mm, 0, lane, nw * sl / nslice, nw * (sl + 1) / nslice);
}

}  // namespace

// AUTO: series of >= 1536 steps (the forward block kernel's own rule for its steady-state phase, kf_forward_b4.hip).  At the headline's
// 1000 steps the launch is as long as the chain of its slowest draws either way -- the 4 % of the structural draws whose filter never
// converges take every step in kf_smooth_b4_kernel, 2.7 us each with nothing else on their SIMD -- and these kernels are links in
// front of that chain (measured, scripts/r04/smooth_split_check.py: 4096 x 1000 4.8 ms against 4.5, 8192 x 2000 (C3) 12.6 against
// 14.5).  KF_SMOOTHER_SPLIT=1 / 0 forces it on / off.
bool kf_smoother_b4_split_ok(const KfSmoothArgs& a) {
  if (!(a.m >= 5 && a.m <= 10 && a.flags == 0 && !a.exact && a.T >= 128 && a.B <= 0x7fffffffL)) return false;
  const char* e = kf_env_switch("KF_SMOOTHER_SPLIT");
  if (e && e[0]) return e[0] != '0';
  return a.T >= 1536;
}

// [start B | fill range 2 B | launch order B | guess B | latest mismatch B | P_inf B m^2]
size_t kf_smoother_b4_split_workspace_bytes(long B, int m) {
  const size_t seg = (sizeof(int32_t) * (size_t)B + 63) & ~(size_t)63;
  return 6 * seg + sizeof(double) * (size_t)B * m * m + 256;
}

int kf_launch_smoother_b4_steady(kf_handle* h, const KfSmoothArgs& a, void* ws, int32_t** start, int32_t** list, hipStream_t stream) {
  const size_t seg = (sizeof(int32_t) * (size_t)a.B + 63) & ~(size_t)63;
  char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 63) & ~(uintptr_t)63);
  int32_t* st = reinterpret_cast<int32_t*>(w); w += seg;
  int32_t* rng = reinterpret_cast<int32_t*>(w); w += 2 * seg;
  int32_t* ord = reinterpret_cast<int32_t*>(w); w += seg;
  int32_t* gss = reinterpret_cast<int32_t*>(w); w += seg;
  int32_t* tbd = reinterpret_cast<int32_t*>(w); w += seg;
  double* pinf = reinterpret_cast<double*>(w);
  double tol = 1e-14;
  {
    const char* e = kf_env_experiment("KF_SMOOTHER_SPLIT_TOL");  // (a documented tolerance: measurement builds only, kf_internal.h)
    if (e && atof(e) > 0.0) tol = atof(e);
  }
  hipLaunchKernelGGL(kf_smooth_b4_guess_kernel, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, gss, tbd, tol);
  KF_HIP_CHECK(h, hipGetLastError());
  const int nblk = (a.T - 1 + KF_WAVE - 1) / KF_WAVE;
  for (long b0 = 0; b0 < a.B; b0 += 65535) {   // (grid.y <= 65535)
    KfSmoothArgs av = a;
    const long nb = a.B - b0 < 65535 ? a.B - b0 : 65535;
    av.filt_P = a.filt_P + b0 * (long)a.T * a.m * a.m;
#define KF_SV(M_) \
  case M_: hipLaunchKernelGGL((kf_smooth_b4_verify_kernel<M_>), dim3((unsigned)nblk, (unsigned)nb), dim3(KF_WAVE), 0, stream, av, gss + b0, tbd + b0, tol); break;
    switch (a.m) {
      KF_SV(5) KF_SV(6) KF_SV(7) KF_SV(8) KF_SV(9) KF_SV(10)
      default: return KF_ERR_UNSUPPORTED;
    }
#undef KF_SV
    KF_HIP_CHECK(h, hipGetLastError());
  }
#define KF_SS(M_) \
  case M_: hipLaunchKernelGGL((kf_smooth_b4_steady_kernel<M_>), dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, st, rng, pinf, tol, gss, tbd); break;
  switch (a.m) {
    KF_SS(5) KF_SS(6) KF_SS(7) KF_SS(8) KF_SS(9) KF_SS(10)
    default: return KF_ERR_UNSUPPORTED;
  }
#undef KF_SS
  KF_HIP_CHECK(h, hipGetLastError());
  // the constant covariances: 8 slices per draw (a draw's range is ~0.5 MB at the headline shape)
  const int nslice = 8;
  hipLaunchKernelGGL(kf_smooth_b4_fill_kernel, dim3((unsigned)(a.B * nslice)), dim3(KF_WAVE), sizeof(double) * (size_t)(a.m * a.m + 2), stream, a,
                     rng, pinf, nslice);
  KF_HIP_CHECK(h, hipGetLastError());
  // placement of the sequential sweep: its draws need start[b] steps each
  const int rc = kf_launch_order(h, st, a.T, a.B, ord, stream);
  if (rc) return rc;
  *start = st;
  *list = ord;
  return KF_OK;
}
