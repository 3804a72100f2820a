// kf_smoother_bf_wg.hip -- smoothed moments from the FILTER's predicted tapes on 16 x 16 fp64 MFMA tiles, one workgroup of NB
// wavefronts per draw (scalar observation p == 1, 16 <= k_states <= 79, static T, Z, H, d): the on-chip form of kf_smoother_from_filter
// between the block-tile kernel (kf_smoother_bf.hip, m <= 15) and the HBM-workspace path (kf_smoother_bf_big.hip).
//
// What it replaces: KalmanSmoother.build_graph (pymc_statespace/filters/kalman_smoother.py:56-104) on the outputs of the same model's
// filter (core/statespace.py:176-197).  For this range the reference's recursion -- pinv(P_hat) at every step -- is what the RTS tile
// kernels run (kf_smoother_mfma_nb / _wg.hip: Gauss-Jordan, flagged draws redone with a Jacobi pseudo-inverse), and with state noise of
// few columns (every ARMA / SARIMA model: k_posdef = 1, models/SARIMAX.py:23-24) P_hat sits on pinv's cut-off and most draws take the
// fallback: 1024 draws x 300 steps, r = 1: m = 32 / 48 / 64: 2.0 / 7.3 / 37.7 s.  Round 5 sent those models to the recursion of de Jong /
// Durbin & Koopman on the HBM workspace (no inverse; 43 / 58 / 75 ms, launch bound: profiles/r05/smoother_mid_m_rank.txt).  This is that
// recursion on chip -- the algebra of kf_smoother_bf.hip line by line, on 16 x 16 blocks:
//     v = y - Z a_t - d      F = Z P_t Z' + H      kappa = (P_t Z' / F, -v / F)        (a_t, P_t: PREDICTED; tile index JC = N - 1 carries vectors)
//     Pf_ext = P_ext - F kappa kappa'              G = T_aug - K_aug (Z T)_aug
//     Y = M_t Pf_ext (row JC left out)             S = Pf_ext - P_f Y:   smooth_P[t] = P_f - P_f M_t P_f,  smooth_a[t]' = a_f' + row JC of Y
//     V = M_t G                                    M_{t-1} = G' V + zeta zeta' / F,   zeta = ((Z T)', v),   M_t = T_aug' N_aug,t T_aug
// Every product is of the form U' V with U symmetric (M, Pf_ext) or used transposed on purpose (G'): in the accumulator layout of
// v_mfma_f64_16x16x4 a register block used as the A operand IS its transpose (kf_tile.h), so no matrix is ever transposed.
//
// Layout.  N = 16 NB >= m + 1.  Wavefront J owns column block J of every matrix: register X[K][s] = X[16 K + 4 s + g][16 J + c]
// (lane = 16 g + c).  The left factor of a product comes from an LDS IMAGE every wavefront writes its column block of (block (K, I) of
// the image, read as A operand, is wavefront I's registers): IM1 holds M_t for the whole step, IM2 first Pf_ext, then G.  Four barriers
// per step (the vector exchange; Pf image written; Pf image consumed; G image written); the rank-one parts are element-wise VALU work on
// the registers (kappa by row index through the exchange vector, by column index from the wavefront's own column sums).
// Per step and draw: 4 NB^3 x 4 MFMAs (NB = 2: 128; NB = 4: 1024), 8 (m + m^2) bytes in and out.
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

template <int NB>
struct BfWgLds {
  static constexpr int N = 16 * NB;
  static constexpr int IMG = NB * NB * 4 * KF_WAVE;   // one image: NB x NB blocks of 4 registers x 64 lanes
  static constexpr int OFF_IM2 = IMG, OFF_ZP = 2 * IMG, OFF_VZ = OFF_ZP + N, OFF_VT = OFF_VZ + N, TOTAL = OFF_VT + N;
};

template <int NB>
__global__ __launch_bounds__(KF_WAVE * NB) void kf_smooth_bf_wg_kernel(const KfSmoothBfArgs A) {
  using L = BfWgLds<NB>;
  constexpr int N = L::N, JC = N - 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* IM1 = smem;
  double* IM2 = smem + L::OFF_IM2;
  double* ZP = smem + L::OFF_ZP;
  double* VZ = smem + L::OFF_VZ;
  double* VT = smem + L::OFF_VT;
  const int tid = threadIdx.x, J = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const long b = blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m;
  const int col = 16 * J + c;
  const double* gT = A.Tm + b * (long)mm;
  const double* gZ = A.Z + b * (long)m;
  // ---- per-draw constants ---------------------------------------------------------------------------------------------------------
  for (int e = tid; e < 2 * L::IMG; e += KF_WAVE * NB) smem[e] = 0.0;   // M_{n-1} = 0
  for (int j = tid; j < N; j += KF_WAVE * NB) {
    double zt = 0.0;
    if (j < m)
      for (int l = 0; l < m; ++l) zt = fma(gZ[l], gT[l * m + j], zt);
    VZ[j] = j < m ? gZ[j] : 0.0;
    VT[j] = zt;
    ZP[j] = 0.0;
  }
  // NB = 5 (64 <= m <= 79): five wavefronts share four SIMDs, so each has 256 registers, not 512 -- T_aug is then re-read from L2 where
  // it is used and the next step's tile is not prefetched (first version: 408 spilled registers, 94 ms at m = 64 against 13 at m = 63)
  constexpr bool LOW = NB >= 5;
  double Tr[LOW ? 1 : NB][4];   // T_aug, own column block
  int rowi[NB][4];
  auto t_aug = [&](int r) { return (r < m && col < m) ? gT[r * m + col] : ((r == JC && col == JC) ? 1.0 : 0.0); };
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = 16 * K + 4 * s + g;
      rowi[K][s] = r;
      if (!LOW) Tr[LOW ? 0 : K][s] = t_aug(r);
    }
  __syncthreads();
  const double Zc = VZ[col], ZTc = VT[col];
  const double H1 = A.H[b];
  const double d1 = A.d ? A.d[b] : 0.0;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // kalman_filter.py:335 quirk: y_hat = Z a - d
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
  const double* gpa = A.pred_a + b * (long)(n + 1) * m;
  const double* gpP = A.pred_P + b * (long)(n + 1) * mm;
  double* gsa = A.smooth_a + b * (long)n * m;
  double* gsP = A.smooth_P + b * (long)n * mm;
  // tape image of step t: the state block P_t, a_t in column JC and in row JC, zero elsewhere
  bool onm[NB][4];   // inside the m x m state block
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int s = 0; s < 4; ++s) onm[K][s] = rowi[K][s] < m && col < m;
  double PN[NB][4];
  auto prefetch = [&](int t) {
    const double* tP = gpP + (long)t * mm;
    const double* ta = gpa + (long)t * m;
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = rowi[K][s];
        double x = 0.0;
        if (onm[K][s]) x = tP[r * m + col];
        else if (r < m && col == JC) x = ta[r];
        else if (r == JC && col < m) x = ta[col];
        PN[K][s] = x;
      }
  };
  // (U' V)(I, own column block) = sum_K U(K, I)' V(K, .): U from an LDS image, V(K, s) from the registers (through `vget`)
  auto product = [&](const double* IMU, auto&& vget, double4_t (&R)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const double* blk = IMU + ((K * NB + I) * 4) * KF_WAVE + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(blk[s * KF_WAVE], vget(K, s), acc, 0, 0, 0);
      }
      R[I] = acc;
    }
  };
  double* own1 = IM1 + (J * 4) * KF_WAVE + lane;   // this wavefront's column block of an image: element (K, s) at [(K NB 4 + s) 64]
  double* own2 = IM2 + (J * 4) * KF_WAVE + lane;
  constexpr int KSTR = NB * 4 * KF_WAVE;

  if (!LOW) prefetch(n - 1);
  double ynext = ybase[n - 1];
  for (int t = n - 1; t >= 0; --t) {
    if (LOW) prefetch(t);
    double Sf[NB][4];
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int s = 0; s < 4; ++s) Sf[K][s] = PN[K][s];
    const double yv = ynext;
    if (t > 0) {
      if (!LOW) prefetch(t - 1);
      ynext = ybase[t - 1];
    }
    // ---- (Z P_t | Z a_t) by column, exchanged; F, v, kappa -----------------------------------------------------------------------------
    double zpart = 0.0;
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int s = 0; s < 4; ++s) zpart = fma(VZ[rowi[K][s]], Sf[K][s], zpart);
    const double zPc = col_allsum(zpart);
    if (g == 0) ZP[col] = zPc;
    __syncthreads();                                                                                   // B1
    double fpart = 0.0;
#pragma unroll
    for (int K = 0; K < NB; ++K) fpart = fma(ZP[16 * K + c], VZ[16 * K + c], fpart);
    const double F = H1 + row_allsum(fpart);
    const bool upd = !(yv != yv) && F > 0.0;   // (F <= 0: the filter flagged the draw; UnivariateFilter skips F == 0, :468-471)
    const double rF = upd ? fast_rcp(F) : 0.0;
    const double v = upd ? (yv + dsign * d1) - ZP[JC] : 0.0;
    const double kc = (col == JC ? -v : zPc) * rF;   // kappa by column index
    auto kappa_row = [&](int r) { return (r == JC ? -v : ZP[r]) * rF; };   // ... by row index
    // Pf_ext = P_ext - F kappa kappa' -> its image; as right factor of Y its row JC (a_f') is left out
    double Pb[LOW ? 1 : NB][4];
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = rowi[K][s];
        const double pf = fma(-F * kappa_row(r), kc, Sf[K][s]);
        own2[K * KSTR + s * KF_WAVE] = pf;
        if (!LOW) Pb[LOW ? 0 : K][s] = (r == JC) ? 0.0 : pf;
      }
    double4_t Y[NB], V[NB];
    // Y = M_t Pf_ext (row JC left out).  (LOW: the right factor re-read from this wavefront's own part of the image)
    product(IM1, [&](int K, int s) { return LOW ? ((rowi[K][s] == JC) ? 0.0 : own2[K * KSTR + s * KF_WAVE]) : Pb[LOW ? 0 : K][s]; }, Y);
    auto g_elem = [&](int K, int s) {   // G = T_aug - K_aug (Z T)_aug, K_aug[JC] = 0
      const int r = rowi[K][s];
      const double tv = LOW ? t_aug(r) : Tr[LOW ? 0 : K][s];
      return (r == JC) ? tv : fma(-kappa_row(r), ZTc, tv);
    };
    if (!LOW) product(IM1, g_elem, V);                           // V = M_t G  (LOW: behind S, when Y's registers are free)
    __syncthreads();                                                                                   // B2: Pf image complete
    // ---- S = Pf_ext - P_f Y; row JC: a_s' = a_f' + (row JC of Y) ------------------------------------------------------------------------
    const double yjc = Y[NB - 1][3];   // (meaningful in the lanes of row JC: g == 3)
    double4_t PY[NB];
    product(IM2, [&](int K, int s) { return (rowi[K][s] == JC) ? 0.0 : Y[K][s]; }, PY);
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = rowi[K][s];
        const double pf = own2[K * KSTR + s * KF_WAVE];
        const double sv = (r == JC) ? pf + yjc : pf - PY[K][s];
        if (onm[K][s]) gsP[(long)t * mm + (long)r * m + col] = sv;
        else if (r == JC && col < m) gsa[(long)t * m + col] = sv;
      }
    if (LOW) product(IM1, g_elem, V);                            // (M_t's image is untouched until the end of the step)
    __syncthreads();                                                                                   // B3: Pf image consumed
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int s = 0; s < 4; ++s) own2[K * KSTR + s * KF_WAVE] = g_elem(K, s);
    __syncthreads();                                                                                   // B4: G image complete
    // ---- M_{t-1} = G' V + zeta zeta' / F ------------------------------------------------------------------------------------------------
    double4_t Mn[NB];
    product(IM2, [&](int K, int s) { return V[K][s]; }, Mn);
    const double zc = (col == JC ? v : ZTc) * rF;
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = rowi[K][s];
        // (every wavefront finished reading the old image before B2; the next step's B1 publishes the new one)
        own1[K * KSTR + s * KF_WAVE] = fma(r == JC ? v : VT[r], zc, Mn[K][s]);
      }
  }
}

}  // namespace

bool kf_smoother_bf_wg_ok(int m, int p, unsigned flags) { return p == 1 && m >= 16 && m + 1 <= 80 && !(flags & KF_FLAG_TV_ANY); }

int kf_launch_smoother_bf_wg(kf_handle* h, const KfSmoothBfArgs& a, hipStream_t stream) {
  if (!kf_smoother_bf_wg_ok(a.m, 1, a.flags)) {
    KF_SET_ERR(h, "smoother(from filter, tiles): needs p == 1, 16 <= m <= 79 and static matrices");
    return KF_ERR_UNSUPPORTED;
  }
  const int nb = (a.m + 1 + 15) / 16;
#define KF_SBW(NB_)                                                                                                              \
  do {                                                                                                                           \
    const size_t lds = sizeof(double) * (size_t)BfWgLds<NB_>::TOTAL;                                                             \
    static KfAttrOnce once;                                                                                                      \
    if (lds > 64 * 1024 && once.need(h->device)) {                                                                               \
      KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_smooth_bf_wg_kernel<NB_>),                            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
      once.mark(h->device);                                                                                                      \
    }                                                                                                                            \
    hipLaunchKernelGGL((kf_smooth_bf_wg_kernel<NB_>), dim3((unsigned)a.B), dim3(KF_WAVE * NB_), lds, stream, a);                 \
  } while (0)
  switch (nb) {
    case 2: KF_SBW(2); break;
    case 3: KF_SBW(3); break;
    case 4: KF_SBW(4); break;
    default: KF_SBW(5); break;
  }
#undef KF_SBW
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
