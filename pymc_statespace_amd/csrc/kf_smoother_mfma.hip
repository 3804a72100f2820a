// kf_smoother_mfma.hip -- RTS smoother with every m x m matrix of a draw in MFMA accumulator
// layout (one wavefront per draw, k_states <= 16, static T, R, Q).  Companion of
// kf_forward_mfma.hip (layout and helpers: kf_tile.h).
//
// Reference pymc_statespace/filters/kalman_smoother.py:56-104, backwards over t = n-2 .. 0:
//     P_hat = T P_f T' + R Q R'  (no symmetrisation, :99-104)       X = pinv(P_hat) T P_f
//     a_s[t] = a_f[t] + X' (a_s[t+1] - T a_f[t])    P_s[t] = P_f[t] + X' (P_s[t+1] - P_hat) X
// Products on register tiles (a tile's registers are the matrix as B operand, its transpose as
// A operand):  W2 = P_f T' ;  P_hat = T W2 + RQR' ;  E = D X ;  P_s = X' E + P_f  -- 4 KS MFMAs.
// X: Gauss-Jordan on the pair (P_hat | W), W = W2' (one LDS transpose), m pivots without
// exchanges (P_hat is symmetric positive definite whenever a Cholesky-based pinv is valid);
// per pivot the pivot ROW comes from its lane group by v_permlane swaps, the pivot COLUMN through
// a 128-byte LDS exchange.  A pivot below 1e-9 x the largest diagonal entry flags the draw (the
// LDS kernel decides between Cholesky and pinv at 1e-12: borderline draws are left to it):
// its results are recomputed afterwards by kf_smooth_lds_kernel (Cholesky / Jacobi pinv), which
// exits immediately for unflagged draws -- no host synchronisation.
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int TILE = 16;

// TV (round 5): time-varying T / R / Q ([B][n][..], KF_FLAG_TV_*): the step that smooths t predicts from t with slice t (+ 1 with
// KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING) -- kf_smoother_lds.hip's pairing, which this replaces for 5 <= m <= 16: T' tile reloaded
// (prefetched a step ahead), R Q R' rebuilt per step when R or Q is flagged.
template <int KS, bool TV = false>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu((TV && KS == 4) ? 3 : 4))) void kf_smooth_mfma_kernel(
    const KfSmoothArgs A, int32_t* flags) {
  __shared__ __attribute__((aligned(16))) double TR[TILE * TILE];  // transposes / tape staging
  __shared__ __attribute__((aligned(16))) double EX[TILE];         // column -> row re-indexing
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const int permc = (c & 3) * 4 + (c >> 2);
  const long b = blockIdx.x;
  // (behind kf_smooth_b4_kernel: only the draws it flagged -- this kernel's own verdict replaces the flag)
  if (A.only_flagged && A.only_flagged[b] == 0) return;
  const int n = A.T, r_ = A.r, m = A.m, mm = m * m;
  const bool c_ok = c < m;

  const bool tvT = TV && (A.flags & KF_FLAG_TV_T), tvR = TV && (A.flags & KF_FLAG_TV_R), tvQ = TV && (A.flags & KF_FLAG_TV_Q);
  const int tsh = (TV && (A.flags & KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING)) ? 1 : 0;
  const long bN = b * (long)n;   // slice (b, t) of a time-varying array is (bN + t)
  double TA[KS], RQ[KS];
  bool on[KS];
  int off[KS], offT[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 4 * s + g;
    on[s] = c_ok && k < m;
    off[s] = k * m + c;
    offT[s] = c * m + k;
  }
  auto load_RQ = [&](int t) {   // R Q R' of slice t (static: t ignored)
    const double* gR = A.R + (tvR ? bN + t : b) * (long)m * r_;
    const double* gQ = A.Q + (tvQ ? bN + t : b) * (long)r_ * r_;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + g;
      double sacc = 0.0;
      if (on[s]) {
        for (int l = 0; l < r_; ++l) {
          double rq = 0.0;
          for (int kk = 0; kk < r_; ++kk) rq += gR[k * r_ + kk] * gQ[kk * r_ + l];
          sacc += rq * gR[c * r_ + l];
        }
      }
      RQ[s] = sacc;
    }
  };
  {
    const double* gT = A.Tm + (tvT ? bN : b) * (long)mm;
#pragma unroll
    for (int s = 0; s < KS; ++s) TA[s] = on[s] ? gT[offT[s]] : 0.0;  // T' tile: A operand = T, B operand = T'
    load_RQ(0);
  }
  const double* gfa = A.filt_a + b * (long)n * m;
  const double* gfP = A.filt_P + b * (long)n * mm;
  double* gsa = A.smooth_a + b * (long)n * m;
  double* gsP = A.smooth_P + b * (long)n * mm;
  const int nb128 = mm >> 1;

  // registers -> compact m x m image in LDS -> 16 B per lane, contiguous
  auto store_cov = [&](double* dst, const double (&X)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
      if (on[s]) TR[off[s]] = X[s];
    __syncthreads();
    const double2_t v2 = *reinterpret_cast<const double2_t*>(TR + 2 * lane);
    if (lane < nb128) st_stream(reinterpret_cast<double2_t*>(dst) + lane, v2);
    else if ((mm & 1) && lane == nb128) st_stream(dst + mm - 1, v2.x);
    if (mm > 2 * KF_WAVE) {
      const double2_t w2 = *reinterpret_cast<const double2_t*>(TR + 2 * (lane + KF_WAVE));
      if (lane + KF_WAVE < nb128) st_stream(reinterpret_cast<double2_t*>(dst) + lane + KF_WAVE, w2);
      else if ((mm & 1) && lane + KF_WAVE == nb128) st_stream(dst + mm - 1, w2.x);
    }
    __syncthreads();
  };
  // column-indexed vector (replicated over lane groups) -> its entries 4s+g in every lane
  auto to_rows = [&](double vc, double (&vr)[4]) {
    if (g == 0) EX[permc] = vc;
    __syncthreads();
    const double2_t a01 = *reinterpret_cast<const double2_t*>(EX + 4 * g);
    const double2_t a23 = *reinterpret_cast<const double2_t*>(EX + 4 * g + 2);
    vr[0] = a01.x; vr[1] = a01.y; vr[2] = a23.x; vr[3] = a23.y;
    __syncthreads();
  };

  double PS[KS], PN[KS];
  double as_c, an;
  double TN[KS];   // TV: the T' tile of the step being prefetched
  auto prefetch = [&](int t) {
    const double* rp = gfP + (long)t * mm;
#pragma unroll
    for (int s = 0; s < KS; ++s) PN[s] = on[s] ? rp[off[s]] : 0.0;
    an = c_ok ? gfa[(long)t * m + c] : 0.0;
    if (tvT && t + tsh < n) {   // the first call, prefetch(n - 1), with the reference's pairing (tsh = 1) would read slice n: past the
                                // end of the last draw's [n][m][m] block; the loop's first step reloads TN anyway
      const double* tp = A.Tm + (bN + t + tsh) * (long)mm;
#pragma unroll
      for (int s = 0; s < KS; ++s) TN[s] = on[s] ? tp[offT[s]] : 0.0;
    }
  };
  prefetch(n - 1);
#pragma unroll
  for (int s = 0; s < KS; ++s) PS[s] = PN[s];
  as_c = an;
  store_cov(gsP + (long)(n - 1) * mm, PS);  // last smoothed moments = last filtered ones
  if (g == 0 && c_ok) gsa[(long)(n - 1) * m + c] = as_c;
  bool bad = false;
  if (n > 1) prefetch(n - 2);

  for (int t = n - 2; t >= 0; --t) {
    double PF[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) PF[s] = PN[s];
    const double af_c = an;
    if (tvT) {
#pragma unroll
      for (int s = 0; s < KS; ++s) TA[s] = TN[s];
    }
    if (tvR || tvQ) load_RQ(t + tsh);
    if (t > 0) prefetch(t - 1);

    // ---- P_hat = T P_f T' + RQR' ; W = T P_f ----------------------------------------------------------
    double4_t W2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) W2 = __builtin_amdgcn_mfma_f64_16x16x4f64(PF[s], TA[s], W2, 0, 0, 0);
    double4_t PHv = {RQ[0], KS > 1 ? RQ[KS > 1 ? 1 : 0] : 0.0, KS > 2 ? RQ[KS > 2 ? 2 : 0] : 0.0,
                     KS > 3 ? RQ[KS > 3 ? 3 : 0] : 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) PHv = __builtin_amdgcn_mfma_f64_16x16x4f64(TA[s], W2[s], PHv, 0, 0, 0);
    // W = (P_f T')' through LDS: element (i, j) of W2 at TR[i*16 + perm(j)]
#pragma unroll
    for (int s = 0; s < KS; ++s) TR[(4 * s + g) * TILE + permc] = W2[s];
    __syncthreads();
    double PH[KS], Wt[KS], D[KS];
    {
      const double2_t t01 = *reinterpret_cast<const double2_t*>(TR + c * TILE + 4 * g);
      const double2_t t23 = *reinterpret_cast<const double2_t*>(TR + c * TILE + 4 * g + 2);
      const double tw[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        Wt[s] = tw[s];
        PH[s] = PHv[s];
        D[s] = PS[s] - PHv[s];  // P_s[t+1] - P_hat
      }
    }
    __syncthreads();
    // pivot threshold: 1e-9 x the largest diagonal entry of P_hat
    double dg = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) dg = (c == 4 * s + g) ? PH[s] : dg;
    const double tol = 1e-9 * wave_allmax(dg);

    // ---- Gauss-Jordan: (P_hat | W) -> (I | X) ------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 4 * KS; ++k) {
      if (k < m) {
        const int ks = k >> 2, kg = k & 3;
        // pivot row (column-indexed, in every lane group)
        double rowA, rowW;
        switch (kg) {
          case 0: rowA = group_bcast<0>(PH[ks]); rowW = group_bcast<0>(Wt[ks]); break;
          case 1: rowA = group_bcast<1>(PH[ks]); rowW = group_bcast<1>(Wt[ks]); break;
          case 2: rowA = group_bcast<2>(PH[ks]); rowW = group_bcast<2>(Wt[ks]); break;
          default: rowA = group_bcast<3>(PH[ks]); rowW = group_bcast<3>(Wt[ks]); break;
        }
        const double d = readlane_f64(rowA, k);
        bad = bad || !(d > tol);
        const double invd = fast_rcp(d);
        // pivot column (row-indexed): lanes c == k publish their registers
        if (c == k) {
#pragma unroll
          for (int s = 0; s < KS; ++s) EX[4 * g + s] = PH[s];
        }
        __syncthreads();
        const double2_t q01 = *reinterpret_cast<const double2_t*>(EX + 4 * g);
        const double2_t q23 = *reinterpret_cast<const double2_t*>(EX + 4 * g + 2);
        const double colk[4] = {q01.x, q01.y, q23.x, q23.y};
        __syncthreads();
        const double ra = rowA * invd, rw = rowW * invd;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const double f = colk[s];
          const double nA = fma(-f, ra, PH[s]);
          const double nW = fma(-f, rw, Wt[s]);
          const bool piv = (s == ks) && (g == kg);  // the pivot row itself becomes row / d
          PH[s] = piv ? ra : nA;
          Wt[s] = piv ? rw : nW;
        }
      }
    }
    // Wt = X = P_hat^-1 T P_f

    // ---- P_s[t] = P_f + X' (D X) -------------------------------------------------------------------------------------
    double4_t E = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) E = __builtin_amdgcn_mfma_f64_16x16x4f64(D[s], Wt[s], E, 0, 0, 0);
    double4_t PSv = {PF[0], KS > 1 ? PF[KS > 1 ? 1 : 0] : 0.0, KS > 2 ? PF[KS > 2 ? 2 : 0] : 0.0,
                     KS > 3 ? PF[KS > 3 ? 3 : 0] : 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) PSv = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[s], E[s], PSv, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) PS[s] = PSv[s];

    // ---- a_s[t] = a_f + X' (a_s[t+1] - T a_f) ----------------------------------------------------------------------------
    double afr[4], dfr[4];
    to_rows(af_c, afr);
    double part = TA[0] * afr[0];
#pragma unroll
    for (int s = 1; s < KS; ++s) part = fma(TA[s], afr[s], part);
    // (T a_f)_c = sum_k T[c][k] a_k : TA[s] = T[c][4s+g]
    const double diff_c = as_c - col_allsum(part);
    to_rows(diff_c, dfr);
    double px = Wt[0] * dfr[0];
#pragma unroll
    for (int s = 1; s < KS; ++s) px = fma(Wt[s], dfr[s], px);
    as_c = af_c + col_allsum(px);  // (X' diff)_c = sum_i X[i][c] diff_i

    store_cov(gsP + (long)t * mm, PS);
    if (g == 0 && c_ok) gsa[(long)t * m + c] = as_c;
  }
  if (lane == 0) flags[b] = bad ? 1 : 0;
}

}  // namespace

size_t kf_smoother_mfma_workspace_bytes(long B) { return sizeof(int32_t) * (size_t)B + 64; }

int kf_launch_smoother_mfma(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                            hipStream_t stream) {
  if (a.m > TILE || !workspace || workspace_bytes < kf_smoother_mfma_workspace_bytes(a.B)) {
    KF_SET_ERR(h, "smoother(mfma): needs m <= %d and %zu bytes of workspace", TILE,
               kf_smoother_mfma_workspace_bytes(a.B));
    return KF_ERR_UNSUPPORTED;
  }
  int32_t* flags = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  const int ks = (a.m + 3) / 4;
  const bool tv = (a.flags & (KF_FLAG_TV_T | KF_FLAG_TV_R | KF_FLAG_TV_Q)) != 0;
#define KF_SM_MFMA(KS_)                                                                                                  \
  if (tv) hipLaunchKernelGGL((kf_smooth_mfma_kernel<KS_, true>), dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, flags); \
  else hipLaunchKernelGGL((kf_smooth_mfma_kernel<KS_>), dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, flags)
  switch (ks) {
    case 1: KF_SM_MFMA(1); break;
    case 2: KF_SM_MFMA(2); break;
    case 3: KF_SM_MFMA(3); break;
    default: KF_SM_MFMA(4); break;
  }
#undef KF_SM_MFMA
  KF_HIP_CHECK(h, hipGetLastError());
  // draws whose P_hat was not safely positive definite: pinv path, other draws exit at once
  KfSmoothArgs fb = a;
  fb.only_flagged = flags;
  return kf_launch_smoother_lds(h, fb, stream);
}
