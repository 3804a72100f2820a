// kf_smoother_mfma_wg.hip -- RTS smoother for 48 < k_states <= 64: one workgroup of four wavefronts
// per draw, wavefront J owning column block J of every m x m matrix in MFMA accumulator layout
// (register X[s] = X[4 s + g][16 J + c], s < KSR = ceil(m / 4)); static T, R, Q.  Companion of
// kf_forward_mfma_wg.hip and the workgroup generalisation of kf_smoother_mfma_nb.hip.
//
// Reference pymc_statespace/filters/kalman_smoother.py:56-104, backwards over t = n-2 .. 0:
//     P_hat = T P_f T' + R Q R'                                    X = pinv(P_hat) T P_f
//     a_s[t] = a_f[t] + X' (a_s[t+1] - T a_f[t])    P_s[t] = P_f[t] + X' (P_s[t+1] - P_hat) X
// Per wavefront (column block J): W2(:, J) = P_f T'(:, J), P_hat(:, J) = T W2(:, J) + RQR'(:, J),
// E(:, J) = D X(:, J), P_s(:, J) = X' E(:, J) + P_f(:, J): 4 x 4 x KSR MFMAs whose A operands (P_f, T,
// D, X' -- all column blocks) come from LDS images every wavefront writes its block of, B operands
// and accumulators are its own registers; W = W2' through the 64 x 64 LDS transpose tile.
// X: Gauss-Jordan on (P_hat | W) without exchanges, one barrier per pivot: the wavefront that holds
// pivot column k publishes it (double-buffered), the pivot row is in every wavefront's own
// registers (lane group k & 3, slice k >> 2).
// A pivot below 1e-9 x the largest diagonal entry flags the draw: with the per-draw flag workspace
// (4 B + 64 bytes) its results are recomputed afterwards by kf_smooth_pinv_wide_kernel
// (kf_smoother_lds.hip: Jacobi pinv at every step, as the reference), which exits at once for
// unflagged draws; without the workspace the draw's outputs become NaN.  (The HBM-workspace smoother
// this kernel replaces in the range, 1.2 - 1.6 s per 1000 steps at 4096 draws, is Cholesky only.)
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int NBW = 4;
constexpr int NT = NBW * KF_WAVE;

template <int KSR>
struct WgSmLayout {
  static constexpr int KSRP = KSR + (KSR & 1);
  static constexpr int LD = 4 * KSRP + 2;
  static constexpr int IMG = NBW * KSR * KF_WAVE;
  // one buffer, four uses in turn (a barrier between each): image of P_f -> transpose tile of W2 -> image
  // of D -> image of X.  With the static image of T: ~70 KB, two draws per CU.
  static constexpr int OFF_B1 = IMG;
  static constexpr int OFF_B2 = OFF_B1;
  static constexpr int B2 = (4 * KSR * LD > IMG) ? 4 * KSR * LD : IMG;
  static constexpr int OFF_EXK = OFF_B2 + B2;          // pivot column, double buffered
  static constexpr int OFF_EXD = OFF_EXK + 2 * 4 * KSRP;
  static constexpr int OFF_RED = OFF_EXD + 4 * KSRP;
  static constexpr int LOOP = OFF_RED + NBW;
  static int doubles(int m, int r) { return LOOP > 2 * m * r ? LOOP : 2 * m * r; }  // setup scratch aliases everything
  static bool setup_fits(int m, int r) { return sizeof(double) * (size_t)doubles(m, r) <= 80 * 1024; }
};

template <int KSR>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void kf_smooth_mfma_wg_kernel(const KfSmoothArgs A,
                                                                                                     int32_t* flags) {
  using L = WgSmLayout<KSR>;
  constexpr int KSRP = L::KSRP, LD = L::LD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* TL = smem;
  double* B1 = smem + L::OFF_B1;
  double* B2 = smem + L::OFF_B2;
  double* EXK = smem + L::OFF_EXK;
  double* EXD = smem + L::OFF_EXD;
  double* RED = smem + L::OFF_RED;
  const int tid = threadIdx.x, lane = tid & 63, J = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, mm = m * m;
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

  double RQ[KSR];  // RQ[s] = (R Q R')[4 s + g][j]
  // TA(s) = T[j][4 s + g], own column block: read back from the image (registers are scarce at two
  // wavefronts per SIMD)
  auto TA = [&](int s) { return TL[(J * KSR + s) * KF_WAVE + lane]; };
  {
    double* sR = smem;             // R     m x r   (Q is read from global memory)
    double* sRQ = sR + m * r_;     // R Q   m x r
    const double* gR = A.R + b * (long)m * r_;
    const double* gQ = A.Q + b * (long)r_ * r_;
    const double* gT = A.Tm + b * (long)mm;
    for (int e = tid; e < m * r_; e += NT) sR[e] = gR[e];
    __syncthreads();
    for (int e = tid; e < m * r_; e += NT) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      for (int kk = 0; kk < r_; ++kk) v += sR[i * r_ + kk] * gQ[kk * r_ + l];
      sRQ[e] = v;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      const int k = 4 * s + g;
      const bool on = cok && k < m;
      double acc = 0.0;
      if (on)
        for (int l = 0; l < r_; ++l) acc += sRQ[k * r_ + l] * sR[j * r_ + l];
      RQ[s] = acc;
    }
    __syncthreads();  // scratch is dead
#pragma unroll
    for (int s = 0; s < KSR; ++s)
      TL[(J * KSR + s) * KF_WAVE + lane] = (cok && rok(s)) ? gT[j * m + 4 * s + g] : 0.0;
    if (tid < 2 * 4 * KSRP) EXK[tid] = 0.0;
    if (tid < 4 * KSRP) EXD[tid] = 0.0;
    __syncthreads();
  }
  const double* gfa = A.filt_a + b * (long)n * m;
  const double* gfP = A.filt_P + b * (long)n * mm;
  double* gsa = A.smooth_a + b * (long)n * m;
  double* gsP = A.smooth_P + b * (long)n * mm;

  double PS[KSR], as_c;
#pragma unroll
  for (int s = 0; s < KSR; ++s) {
    PS[s] = (cok && rok(s)) ? gfP[(long)(n - 1) * mm + obase + 4 * s * m] : 0.0;
    if (cok && rok(s)) gsP[(long)(n - 1) * mm + obase + 4 * s * m] = PS[s];  // last smoothed = last filtered
  }
  as_c = cok ? gfa[(long)(n - 1) * m + j] : 0.0;
  if (g == 0 && cok) gsa[(long)(n - 1) * m + j] = as_c;
  bool bad = false;
  int par = 0;

  for (int t = n - 2; t >= 0; --t) {
    // ---- a_f, T a_f ; image of P_f ---------------------------------------------------------------------------
    const double af_c = cok ? gfa[(long)t * m + j] : 0.0;
    {
      double part = 0.0;
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        B1[(J * KSR + s) * KF_WAVE + lane] = (cok && rok(s)) ? gfP[(long)t * mm + obase + 4 * s * m] : 0.0;
        const double afr = rok(s) ? gfa[(long)t * m + 4 * s + g] : 0.0;
        part = fma(TA(s), afr, part);
      }
      const double diff_c = as_c - col_allsum(part);  // a_s[t+1] - T a_f,  (T a_f)_j = sum_k T[j][k] a_k
      if (g == 0 && j < 4 * KSR) EXD[pos(j)] = diff_c;
    }
    __syncthreads();
    // ---- W2(:, J) = P_f T'(:, J) ; P_hat(:, J) = T W2(:, J) + RQR'(:, J) ------------------------------------------
    double4_t W2[NBW], PHv[NBW];
#pragma unroll
    for (int I = 0; I < NBW; ++I) {
      W2[I] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 4; ++u) PHv[I][u] = (4 * I + u < KSR) ? RQ[(4 * I + u < KSR) ? 4 * I + u : 0] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR)
          W2[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(B1[(I * KSR + ks) * KF_WAVE + lane], TA(ks), W2[I], 0, 0, 0);
    __syncthreads();  // image of P_f read by every wavefront: the buffer becomes the transpose tile
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR)
          PHv[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(TL[(I * KSR + ks) * KF_WAVE + lane], W2[ks >> 2][ks & 3], PHv[I], 0, 0, 0);
    // W = W2' through the transpose tile: element (i, j) of W2 at B2[i * LD + pos(j)]
#pragma unroll
    for (int s = 0; s < KSR; ++s)
      if (j < 4 * KSR) B2[(4 * s + g) * LD + pos(j)] = W2[s >> 2][s & 3];
    __syncthreads();
    double Wt[KSR], PH[KSR], D[KSR], dfr[KSRP];
    {
      double tw[KSRP];
      read_rows(B2 + (j < 4 * KSR ? j : 4 * KSR - 1) * LD, tw);
      double dg = 0.0;
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        Wt[s] = cok ? tw[s] : 0.0;
        PH[s] = PHv[s >> 2][s & 3];
        D[s] = PS[s] - PH[s];  // P_s[t+1] - P_hat
        dg = (j == 4 * s + g) ? PH[s] : dg;
      }
      read_rows(EXD, dfr);
      // pivot threshold: 1e-9 x the largest diagonal entry of P_hat (over the four wavefronts)
      dg = wave_allmax(dg);
      if (lane == 0) RED[J] = dg;
    }
    __syncthreads();
    const double tol = 1e-9 * fmax(fmax(RED[0], RED[1]), fmax(RED[2], RED[3]));

    // ---- Gauss-Jordan: (P_hat | W) -> (I | X), one barrier per pivot ---------------------------------------------
#pragma unroll
    for (int k = 0; k < 4 * KSR; ++k) {
      if (k < m) {
        const int ks = k >> 2, kg = k & 3, Jk = k >> 4;
        double* exk = EXK + par * 4 * KSRP;
        par ^= 1;
        if (J == Jk && c == (k & 15)) {  // the lanes that hold column k publish it (row-indexed)
#pragma unroll
          for (int s = 0; s < KSR; ++s) exk[g * KSRP + s] = PH[s];
        }
        const double rowW = group_bcast_sel(Wt[ks], kg);
        const double rowA = group_bcast_sel(PH[ks], kg);
        __syncthreads();
        double colk[KSRP];
        read_rows(exk, colk);
        const double d = exk[kg * KSRP + ks];
        bad = bad || !(d > tol);
        const double invd = fast_rcp(d);
        const double rw = rowW * invd, ra = rowA * invd;
#pragma unroll
        for (int s = 0; s < KSR; ++s) {
          const bool piv = (s == ks) && (g == kg);  // the pivot row itself becomes row / d
          const double nW = fma(-colk[s], rw, Wt[s]);
          Wt[s] = piv ? rw : nW;
          if (J >= Jk) {  // column blocks left of the pivot are identity already
            const double nA = fma(-colk[s], ra, PH[s]);
            PH[s] = piv ? ra : nA;
          }
        }
      }
    }
    // Wt = X(:, J) = (P_hat^-1 T P_f)(:, J)

    // ---- image of D ; E(:, J) = D X(:, J) ; image of X ; P_s(:, J) = P_f(:, J) + X' E(:, J) -------------------
#pragma unroll
    for (int s = 0; s < KSR; ++s) B1[(J * KSR + s) * KF_WAVE + lane] = D[s];
    __syncthreads();
    double4_t E[NBW], PSv[NBW];
#pragma unroll
    for (int I = 0; I < NBW; ++I) E[I] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR)
          E[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(B1[(I * KSR + ks) * KF_WAVE + lane], Wt[ks], E[I], 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KSR; ++s) B2[(J * KSR + s) * KF_WAVE + lane] = Wt[s];
#pragma unroll
    for (int I = 0; I < NBW; ++I)
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // seed: P_f, re-read (L2) rather than held through the elimination
        const int s = (4 * I + u < KSR) ? 4 * I + u : 0;
        PSv[I][u] = (4 * I + u < KSR && cok && rok(s)) ? gfP[(long)t * mm + obase + 4 * s * m] : 0.0;
      }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NBW; ++I)
        if (4 * I < KSR)
          PSv[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(B2[(I * KSR + ks) * KF_WAVE + lane], E[ks >> 2][ks & 3], PSv[I], 0, 0, 0);
    // ---- a_s[t] = a_f + X' (a_s[t+1] - T a_f) ; tapes ---------------------------------------------------------------
    double px = Wt[0] * dfr[0];
#pragma unroll
    for (int s = 1; s < KSR; ++s) px = fma(Wt[s], dfr[s], px);
    as_c = af_c + col_allsum(px);  // (X' diff)_j = sum_i X[i][j] diff_i
    if (g == 0 && cok) gsa[(long)t * m + j] = as_c;
#pragma unroll
    for (int s = 0; s < KSR; ++s) {
      PS[s] = (cok && rok(s)) ? PSv[s >> 2][s & 3] : 0.0;
      if (cok && rok(s)) st_stream(gsP + (long)t * mm + obase + 4 * s * m, PS[s]);
    }
    __syncthreads();  // the images are rewritten by the next step
  }
  if (flags) {  // the pinv kernel launched behind recomputes flagged draws (kf_smooth_pinv_wide_kernel)
    if (tid == 0) flags[b] = bad ? 1 : 0;
  } else if (bad) {  // no flag workspace: not safely positive definite -> NaN, loudly
    __threadfence();
    __syncthreads();
    const double qnan = __builtin_nan("");
    for (long e = tid; e < (long)n * m; e += NT) gsa[e] = qnan;
    for (long e = tid; e < (long)n * mm; e += NT) gsP[e] = qnan;
  }
}

template <int KSR>
int launch_sm_wg(kf_handle* h, const KfSmoothArgs& a, int32_t* flags, hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)WgSmLayout<KSR>::doubles(a.m, a.r);
  static KfAttrOnce attr_set;  // the attribute is per device
  if (attr_set.need(h->device)) {
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&kf_smooth_mfma_wg_kernel<KSR>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, KF_LDS_BYTES_MAX));
    attr_set.mark(h->device);
  }
  hipLaunchKernelGGL((kf_smooth_mfma_wg_kernel<KSR>), dim3((unsigned)a.B), dim3(NT), lds, stream, a, flags);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

bool kf_smoother_mfma_wg_ok(int m, int r) {
  if (!(m > KF_NB_M_MAX - 1 && m <= KF_WG_M_MAX) || kf_smoother_lds_fits(m)) return false;
  switch ((m + 3) / 4) {
    case 12: return WgSmLayout<12>::setup_fits(m, r);
    case 13: return WgSmLayout<13>::setup_fits(m, r);
    case 14: return WgSmLayout<14>::setup_fits(m, r);
    case 15: return WgSmLayout<15>::setup_fits(m, r);
    default: return WgSmLayout<16>::setup_fits(m, r);
  }
}

int kf_launch_smoother_mfma_wg(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
  if (!kf_smoother_mfma_wg_ok(a.m, a.r)) {
    KF_SET_ERR(h, "smoother(mfma_wg): m=%d r=%d out of range (47 < m <= %d)", a.m, a.r, KF_WG_M_MAX);
    return KF_ERR_UNSUPPORTED;
  }
  // with the per-draw flag workspace (kf_workspace_bytes) flagged draws are redone with the pinv; without
  // it they come back as NaN
  int32_t* flags = nullptr;
  if (workspace && workspace_bytes >= kf_smoother_mfma_workspace_bytes(a.B))
    flags = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  int rc;
  switch ((a.m + 3) / 4) {
    case 12: rc = launch_sm_wg<12>(h, a, flags, stream); break;
    case 13: rc = launch_sm_wg<13>(h, a, flags, stream); break;
    case 14: rc = launch_sm_wg<14>(h, a, flags, stream); break;
    case 15: rc = launch_sm_wg<15>(h, a, flags, stream); break;
    default: rc = launch_sm_wg<16>(h, a, flags, stream); break;
  }
  if (rc != KF_OK || !flags) return rc;
  return kf_launch_smoother_pinv_wide(h, a, flags, stream);
}
