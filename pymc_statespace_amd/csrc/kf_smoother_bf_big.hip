// kf_smoother_bf_big.hip -- kf_smoother_from_filter for state dimensions beyond the on-chip kernels (m > 15, p == 1, static
// matrices): the recursion of kf_smoother_bf.hip (de Jong / Durbin-Koopman; what statsmodels computes) on the HBM workspace with
// kf_forward_big.hip's fp64-MFMA GEMMs.  Reference being replaced: KalmanSmoother.build_graph (filters/kalman_smoother.py:56-104)
// behind this library's filter on the same model -- BASELINE config 5 (daily seasonality, m = 367).
//
// Why.  The reference's recursion needs pinv(P_hat) per step; kf_smoother_big.hip does it by blocked Cholesky + two triangular solves
// (2.3 m^3 of mostly sequential panel work in ~90 launches per step) between four m^3 products: 2.9 ms per step at C5.  With the filter's
// predicted tapes no inverse is needed, and two of the four products have the model's T as a factor, whose all-zero 64 x 32 blocks
// are skipped (big_tmask_kernel, kf_forward_big.hip):
//     kappa = P_t Z' / F        P_f = P_t - F kappa kappa'        a_f = a_t + kappa v                         (prep, one pass over P_t)
//     smooth_a[t] = a_f + P_f r~_t                                                                            (the same pass)
//     Y = M_t P_f               smooth_P[t] = P_f - P_f Y                                                     two dense products
//     w = M_t kappa             X2 = T' M_t - (Z T)' w'                = G' M_t,  G = T - kappa (Z T)         block-sparse product + rank one
//     M_{t-1} = X2 T - (X2 kappa)(Z T) + (Z T)'(Z T) / F               = G' M_t G + ...                       block-sparse product + rank two
//     r~_{t-1} = T' r~_t - (Z T)' (kappa . r~_t) + (Z T)' v / F
// with M_t = T' N_t T, r~_t = T' r_t (M_{n-1} = 0, r~_{n-1} = 0) -- the state block and the vector of kf_smoother_bf.hip's augmented
// tile kept apart.  Eleven launches per step, four of them GEMMs.  A missing observation: kappa = 0, 1 / F := 0.
// When T' is sparse without dense rows (the seasonal / trigonometric models; companion columns up to 32 entries) the last two products are
// one CSR gather, M_{t-1} = T' M_t T - (Z T)'(T' w)' - (T' w)(Z T) + (w . kappa + 1 / F)(Z T)'(Z T)  (kf_bsp_sandwich_launch,
// kf_forward_big.hip): ten launches, two GEMMs; C5 0.99 -> 0.80 ms per step.  The verdict on T' is read back once per call.
#include "kf_internal.h"

namespace {

constexpr int BB_T = 256;

__device__ __forceinline__ double bb_block_sum(double v, double* red) {  // sum over the 256 threads of a workgroup, in every thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// once per call: TT = T' per draw, Z T per draw
__global__ __launch_bounds__(BB_T) void bbb_setup_kernel(const double* __restrict__ T, const double* __restrict__ Z, int m,
                                                         double* __restrict__ TT, double* __restrict__ ZT) {
  const long b = blockIdx.y;
  const double* Tb = T + b * (long)m * m;
  double* TTb = TT + b * (long)m * m;
  for (long e = (long)blockIdx.x * BB_T + threadIdx.x; e < (long)m * m; e += (long)gridDim.x * BB_T) {
    const int i = (int)(e / m), j = (int)(e - (long)i * m);
    TTb[(long)j * m + i] = Tb[e];
  }
  if (blockIdx.x == 0) {
    const double* Zb = Z + b * (long)m;
    for (int j = threadIdx.x; j < m; j += BB_T) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) {
        const double z = Zb[k];
        if (z != 0.0) s = fma(z, Tb[(long)k * m + j], s);
      }
      ZT[b * (long)m + j] = s;
    }
  }
}

struct BbbVec {   // per-draw vectors [B][m] and scalars [B] in the workspace
  double *kap, *w, *x2k, *negZT, *u2, *r, *rn, *af, *twv, *trv;
  double *F, *v, *rF;
};

// per draw: (Z P_t | Z a_t), F, v, kappa, a_f, u2 = (Z T) / F
__global__ __launch_bounds__(BB_T) void bbb_prep_kernel(const KfSmoothBfArgs A, int t, const double* __restrict__ ZT, const BbbVec V) {
  __shared__ double red[4];
  const long b = blockIdx.x;
  const int m = A.m, n = A.T;
  const double* P = A.pred_P + (b * (long)(n + 1) + t) * (long)m * m;
  const double* a = A.pred_a + (b * (long)(n + 1) + t) * (long)m;
  const double* Zb = A.Z + b * (long)m;
  const double yv = A.y[((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L) + t];
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  double fz = 0.0, za = 0.0;
  // zP_j for this thread's columns (kept in registers: m <= 4 * 256)
  double zp[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < m; ++k) {
    const double z = Zb[k];
    if (z == 0.0) continue;   // (wave-uniform: the observation vector of a structural model has a handful of entries)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = threadIdx.x + u * BB_T;
      if (j < m) zp[u] = fma(z, P[(long)k * m + j], zp[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = threadIdx.x + u * BB_T;
    if (j < m) {
      fz = fma(zp[u], Zb[j], fz);
      za = fma(Zb[j], a[j], za);
    }
  }
  const double F = A.H[b] + bb_block_sum(fz, red);
  const double Za = bb_block_sum(za, red);
  const bool upd = !(yv != yv) && F > 0.0;
  const double rF = upd ? 1.0 / F : 0.0;
  const double v = upd ? (yv + dsign * (A.d ? A.d[b] : 0.0)) - Za : 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = threadIdx.x + u * BB_T;
    if (j < m) {
      const double kj = zp[u] * rF;
      V.kap[b * (long)m + j] = kj;
      V.af[b * (long)m + j] = fma(kj, v, a[j]);
      V.u2[b * (long)m + j] = ZT[b * (long)m + j] * rF;
      V.negZT[b * (long)m + j] = -ZT[b * (long)m + j];
    }
  }
  if (threadIdx.x == 0) {
    V.F[b] = upd ? F : 0.0;
    V.v[b] = v;
    V.rF[b] = rF;
  }
}

// one pass over P_t: P_f = P_t - F kappa kappa' (written), smooth_a[t] = a_f + P_f r~_t; 8 rows per workgroup
__global__ __launch_bounds__(BB_T) void bbb_pf_kernel(const KfSmoothBfArgs A, int t, const BbbVec V, double* __restrict__ Pf) {
  const long b = blockIdx.y;
  const int m = A.m, n = A.T;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* P = A.pred_P + (b * (long)(n + 1) + t) * (long)m * m;
  const double* kap = V.kap + b * (long)m;
  const double* r = V.r + b * (long)m;
  const double F = V.F[b];
  double* Pfb = Pf + b * (long)m * m;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int i = blockIdx.x * 8 + wave * 2 + rr;
    if (i >= m) break;
    const double fk = F * kap[i];
    double s = 0.0;
    for (int j = lane; j < m; j += 64) {
      const double pf = fma(-fk, kap[j], P[(long)i * m + j]);
      Pfb[(long)i * m + j] = pf;
      s = fma(pf, r[j], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) A.smooth_a[(b * (long)n + t) * m + i] = V.af[b * (long)m + i] + s;
  }
}

// out = Mat x (dense, row-major [B][m][m]); with `mask` (the block map of Mat: big_tmask_kernel) the 64-column pieces whose two 32-column
// chunks are all zero in the row's 64-row block are not read
__global__ __launch_bounds__(BB_T) void bbb_gemv_kernel(const double* __restrict__ Mat, const uint32_t* __restrict__ mask, int mask_rb,
                                                        const double* __restrict__ x, double* __restrict__ out, int m) {
  const long b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* Mb = Mat + b * (long)m * m;
  const double* xb = x + b * (long)m;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int i = blockIdx.x * 8 + wave * 2 + rr;
    if (i >= m) break;
    const unsigned bits = mask ? mask[b * mask_rb + i / 64] : 0xffffffffu;
    double s = 0.0;
    for (int j0 = 0; j0 < m; j0 += 64) {
      if (!((bits >> (j0 / 32)) & 3u)) continue;
      const int j = j0 + lane;
      if (j < m) s = fma(Mb[(long)i * m + j], xb[j], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[b * (long)m + i] = s;
  }
}

// per draw, behind w = M kappa, twv = T' w, trv = T' r~:  x2k = X2 kappa = T' w - (Z T)' (w . kappa)  (negated: the rank-one factor of the
// last product),  r~_{t-1} = T' r~ - (Z T)' (kappa . r~) + (Z T)' v / F
__global__ __launch_bounds__(BB_T) void bbb_vec_kernel(int m, const double* __restrict__ ZT, const BbbVec V, int fold_u2) {
  __shared__ double red[4];
  const long b = blockIdx.x;
  double wk = 0.0, kr = 0.0;
  for (int j = threadIdx.x; j < m; j += BB_T) {
    const double k = V.kap[b * (long)m + j];
    wk = fma(V.w[b * (long)m + j], k, wk);
    kr = fma(V.r[b * (long)m + j], k, kr);
  }
  wk = bb_block_sum(wk, red);
  kr = bb_block_sum(kr, red);
  const double vr = V.v[b] * V.rF[b];
  for (int j = threadIdx.x; j < m; j += BB_T) {
    const double zt = ZT[b * (long)m + j];
    // (fold_u2: the CSR form adds both rank-one terms with (Z T)' as their right factor at once)
    V.x2k[b * (long)m + j] = -(V.twv[b * (long)m + j] - zt * wk) + (fold_u2 ? V.u2[b * (long)m + j] : 0.0);
    V.rn[b * (long)m + j] = V.trv[b * (long)m + j] + zt * (vr - kr);
  }
}

size_t bb_round(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

bool kf_smoother_bf_big_ok(int m, int p, unsigned flags) { return p == 1 && m > 15 && m <= 1024 && !(flags & KF_FLAG_TV_ANY); }

size_t kf_smoother_bf_big_workspace_bytes(long B, int m) {
  const size_t mm = (size_t)m * m;
  return 5 * bb_round(8 * (size_t)B * mm) + 12 * bb_round(8 * (size_t)B * m) + 3 * bb_round(8 * (size_t)B) +
         bb_round(4 * (size_t)B * ((m + 63) / 64)) + bb_round(kf_bsp_csr_bytes(B, m)) + 512;
}

int kf_launch_smoother_bf_big(kf_handle* h, const KfSmoothBfArgs& a, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const long B = a.B;
  const int m = a.m, n = a.T;
  const size_t need = kf_smoother_bf_big_workspace_bytes(B, m);
  if (!kf_smoother_bf_big_ok(m, 1, a.flags) || !workspace || workspace_bytes < need) {
    KF_SET_ERR(h, "smoother(from filter, workspace path): needs p == 1, 15 < m <= 1024, static matrices and %zu bytes of workspace (got %zu)",
               need, workspace_bytes);
    return workspace && workspace_bytes >= need ? KF_ERR_UNSUPPORTED : KF_ERR_WORKSPACE;
  }
  if (B > 65535) {
    KF_SET_ERR(h, "smoother(from filter, workspace path): B=%ld > 65535 draws per call", B);
    return KF_ERR_UNSUPPORTED;
  }
  const long mm = (long)m * m;
  char* q = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  auto take = [&](size_t bytes) { char* r = q; q += bb_round(bytes); return r; };
  double* M = reinterpret_cast<double*>(take(8 * (size_t)B * mm));
  double* X2 = reinterpret_cast<double*>(take(8 * (size_t)B * mm));
  double* Pf = reinterpret_cast<double*>(take(8 * (size_t)B * mm));
  double* Y = reinterpret_cast<double*>(take(8 * (size_t)B * mm));
  double* TT = reinterpret_cast<double*>(take(8 * (size_t)B * mm));
  BbbVec V;
  double** vecs[] = {&V.kap, &V.w, &V.x2k, &V.negZT, &V.u2, &V.r, &V.rn, &V.af, &V.twv, &V.trv};
  for (auto pv : vecs) *pv = reinterpret_cast<double*>(take(8 * (size_t)B * m));
  double* ZT = reinterpret_cast<double*>(take(8 * (size_t)B * m));
  (void)take(8 * (size_t)B * m);
  V.F = reinterpret_cast<double*>(take(8 * (size_t)B));
  V.v = reinterpret_cast<double*>(take(8 * (size_t)B));
  V.rF = reinterpret_cast<double*>(take(8 * (size_t)B));
  uint32_t* maskT = reinterpret_cast<uint32_t*>(take(4 * (size_t)B * ((m + 63) / 64)));
  const int mask_rb = (m + 63) / 64;

  KF_HIP_CHECK(h, hipMemsetAsync(M, 0, 8 * (size_t)B * mm, st));
  KF_HIP_CHECK(h, hipMemsetAsync(V.r, 0, 8 * (size_t)B * m, st));
  const unsigned sgrid = (unsigned)((mm + BB_T - 1) / BB_T < 256 ? (mm + BB_T - 1) / BB_T : 256);
  hipLaunchKernelGGL(bbb_setup_kernel, dim3(sgrid, (unsigned)B), dim3(BB_T), 0, st, a.Tm, a.Z, m, TT, ZT);
  int rc = kf_big_tmask_launch(st, TT, m, B, maskT);
  if (rc) return rc;
  const uint32_t* mask = maskT;
  {
    const char* te = kf_env_switch("KF_BIG_TMASK");
    if (te && te[0] == '0') mask = nullptr;
  }
  // T' with a few non-zeros per row and no dense row (the seasonal / trigonometric / companion-by-rows models): M_{t-1} = T' M_t T + rank-one
  // terms as a CSR gather (kf_bsp_sandwich_launch) instead of the two block-sparse products.  As in kf_launch_forward_big the verdict is
  // read back once per call (4 + 4 bytes, one stream synchronisation; not while the stream is being captured; KF_BIG_SPARSE=0: never)
  KfBspCsr csr;
  bool sparse = false;
  {
    void* cmem = take(kf_bsp_csr_bytes(B, m));
    const char* se = kf_env_switch("KF_BIG_SPARSE");
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap_status);
    if (!(se && se[0] == '0') && cap_status == hipStreamCaptureStatusNone && kf_bsp_csr_build(st, TT, m, B, cmem, &csr) == KF_OK) {
      int verdict[2] = {0, 1};
      KF_HIP_CHECK(h, hipMemcpyAsync(verdict, csr.ok, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
      KF_HIP_CHECK(h, hipStreamSynchronize(st));
      sparse = verdict[0] != 0 && verdict[1] == 0;
    }
  }
  const unsigned rgrid = (unsigned)((m + 7) / 8);
  const long sTapeS = (long)n * mm;   // stride of the smoothed covariance tape per draw
  for (int t = n - 1; t >= 0; --t) {
    hipLaunchKernelGGL(bbb_prep_kernel, dim3((unsigned)B), dim3(BB_T), 0, st, a, t, ZT, V);
    hipLaunchKernelGGL(bbb_pf_kernel, dim3(rgrid, (unsigned)B), dim3(BB_T), 0, st, a, t, V, Pf);
    // Y = M_t P_f ;  smooth_P[t] = P_f - P_f Y
    rc = kf_big_gemm_launch_x(st, false, false, M, mm, Pf, mm, nullptr, 0, Y, mm, m, B, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, false);
    if (rc) return rc;
    rc = kf_big_gemm_launch_x(st, false, true, Pf, mm, Y, mm, Pf, mm, a.smooth_P + (long)t * mm, sTapeS, m, B, nullptr, 0, /*neg=*/1,
                              nullptr, nullptr, nullptr, nullptr, /*upper_mirror=*/true);
    if (rc) return rc;
    // w = M_t kappa ;  X2 = T' M_t - (Z T)' w'
    hipLaunchKernelGGL(bbb_gemv_kernel, dim3(rgrid, (unsigned)B), dim3(BB_T), 0, st, M, (const uint32_t*)nullptr, mask_rb, V.kap, V.w, m);
    if (!sparse) {
      rc = kf_big_gemm_launch_x(st, false, false, TT, mm, M, mm, nullptr, 0, X2, mm, m, B, mask, /*on_b=*/0, 0, V.negZT, V.w, nullptr, nullptr, false);
      if (rc) return rc;
    }
    // T' w, T' r~ ;  x2k, r~_{t-1}
    hipLaunchKernelGGL(bbb_gemv_kernel, dim3(rgrid, (unsigned)B), dim3(BB_T), 0, st, TT, mask, mask_rb, V.w, V.twv, m);
    hipLaunchKernelGGL(bbb_gemv_kernel, dim3(rgrid, (unsigned)B), dim3(BB_T), 0, st, TT, mask, mask_rb, V.r, V.trv, m);
    hipLaunchKernelGGL(bbb_vec_kernel, dim3((unsigned)B), dim3(BB_T), 0, st, m, ZT, V, sparse ? 1 : 0);
    if (sparse) {
      // M_{t-1} = T' M_t T - (Z T)' (T' w)' + (x2k + (Z T)' / F)(Z T),  x2k = -(T' w - (Z T)' (w . kappa)):  into the other buffer
      rc = kf_bsp_sandwich_launch(st, csr, M, X2, m, B, V.negZT, V.twv, V.x2k, ZT);
      if (rc) return rc;
      double* tm = M;
      M = X2;
      X2 = tm;
    } else {
      // M_{t-1} = X2 T - (X2 kappa)(Z T) + ((Z T)' / F)(Z T):  op(B) = TT' = T, the tile's 64 columns are rows of TT
      rc = kf_big_gemm_launch_x(st, true, false, X2, mm, TT, mm, nullptr, 0, M, mm, m, B, mask, /*on_b=*/1, 0, V.x2k, ZT, V.u2, ZT, /*upper_mirror=*/true);
      if (rc) return rc;
    }
    double* tmp = V.r;
    V.r = V.rn;
    V.rn = tmp;
  }
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
