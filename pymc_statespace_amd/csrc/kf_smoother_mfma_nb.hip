// kf_smoother_mfma_nb.hip -- RTS smoother for 16 < k_states <= 47: every m x m matrix of a draw as
// NB x NB blocks of 16 x 16 in MFMA accumulator layout (one wavefront per draw, static T, R, Q).
// Companion of kf_forward_mfma_nb.hip (register layout X[J][s] = X[4 s + g][16 J + c]) and the
// block generalisation of kf_smoother_mfma.hip.
//
// Reference pymc_statespace/filters/kalman_smoother.py:56-104, backwards over t = n-2 .. 0:
//     P_hat = T P_f T' + R Q R'                                    X = pinv(P_hat) T P_f
//     a_s[t] = a_f[t] + X' (a_s[t+1] - T a_f[t])    P_s[t] = P_f[t] + X' (P_s[t+1] - P_hat) X
// Products on register blocks:
//     W2 = P_f T'            NB^2 KSR MFMAs          P_hat = T W2 + RQR'   blocks I <= J, mirrored
//     E  = D X               NB^2 KSR MFMAs          P_s   = X' E + P_f    blocks I <= J, mirrored
// (P_hat and P_s are symmetric up to rounding; here they are made exactly symmetric -- mirrored
// off-diagonal blocks, averaged diagonal blocks -- which differs from the reference's unsymmetrised
// values at rounding level only).
// X: Gauss-Jordan on the pair (P_hat | W), W = W2', m pivots without exchanges; per pivot the
// pivot ROW comes from its lane group by v_permlane swaps, the pivot COLUMN through an LDS exchange;
// column blocks of P_hat left of the pivot are already identity and are skipped.  A pivot below
// 1e-9 x the largest diagonal entry flags the draw: its results are recomputed afterwards by
// kf_smooth_lds_kernel (Cholesky / Jacobi pinv), which exits at once for unflagged draws.
// Why: before this kernel the range was served by the LDS-resident smoother: 260 ms (m = 20) to
// 4.9 s (m = 40) per 1000 steps at 4096 draws.
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

template <int NB, int KSR>
struct NbSmLayout {
  static constexpr int KSRP = KSR + (KSR & 1);
  static constexpr int LD = 4 * KSRP + 2;
  static constexpr int OFF_RQ = 4 * KSR * LD;             // TR: transpose tile / compact m x m tape image
  static constexpr int OFF_EX = OFF_RQ + NB * KSR * KF_WAVE;  // R Q R' in register layout: [(J KSR + s) 64 + lane]
  static constexpr int LOOP = OFF_EX + 4 * KSRP;
  static int doubles(int m, int r) {
    const int setup = 2 * m * r + r * r;
    return LOOP > setup ? LOOP : setup;
  }
};

template <int NB, int KSR>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(NB == 2 ? 2 : 1))) void kf_smooth_mfma_nb_kernel(
    const KfSmoothArgs A, int32_t* flags) {
  using L = NbSmLayout<NB, KSR>;
  constexpr int KSRP = L::KSRP, LD = L::LD;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* TR = smem;
  double* RQL = smem + L::OFF_RQ;
  double* EX = smem + L::OFF_EX;
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, mm = m * m;

  bool cok[NB];
#pragma unroll
  for (int J = 0; J < NB; ++J) cok[J] = 16 * J + c < m;
  auto rok = [&](int s) { return 4 * s + g < m; };
  auto pos = [&](int j) { return (j & 3) * KSRP + (j >> 2); };
  const int obase = g * m + c;  // offset of element (4 s + g, 16 J + c): obase + 4 s m + 16 J

  const double* gT = A.Tm + b * (long)mm;
  {
    double* sR = smem;
    double* sQ = sR + m * r_;
    double* sRQ = sQ + r_ * r_;
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
    double rq[NB][KSR];
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
        rq[J][s] = acc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s) RQL[(J * KSR + s) * KF_WAVE + lane] = rq[J][s];
    wave_lds_fence();
  }
  const double* gfa = A.filt_a + b * (long)n * m;
  const double* gfP = A.filt_P + b * (long)n * mm;
  double* gsa = A.smooth_a + b * (long)n * m;
  double* gsP = A.smooth_P + b * (long)n * mm;
  const int nb128 = mm >> 1;

  auto load_cov = [&](const double* src, double (&X)[NB][KSR]) {
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s) X[J][s] = (cok[J] && rok(s)) ? src[obase + 4 * s * m + 16 * J] : 0.0;
  };
  auto flush_image = [&](double* dst) {  // compact m x m image in TR -> HBM, 16 B per lane, contiguous
    wave_lds_fence();
    for (int e = lane; e < nb128; e += KF_WAVE)
      st_stream(reinterpret_cast<double2_t*>(dst) + e, *reinterpret_cast<const double2_t*>(TR + 2 * e));
    if ((mm & 1) && lane == 0) st_stream(dst + mm - 1, TR[mm - 1]);
    wave_lds_fence();
  };
  auto read_rows = [&](const double* src, double (&x)[KSRP]) {
#pragma unroll
    for (int s = 0; s < KSRP; s += 2) {
      const double2_t v2 = *reinterpret_cast<const double2_t*>(src + g * KSRP + s);
      x[s] = v2.x;
      x[s + 1] = v2.y;
    }
  };
  auto to_rows = [&](const double (&vc)[NB], double (&vr)[KSRP]) {
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
      if (g == 0 && j < 4 * KSR) EX[pos(j)] = vc[J];
    }
    wave_lds_fence();
    read_rows(EX, vr);
    wave_lds_fence();
  };

  double PS[NB][KSR], as_c[NB];
  load_cov(gfP + (long)(n - 1) * mm, PS);
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    as_c[J] = cok[J] ? gfa[(long)(n - 1) * m + 16 * J + c] : 0.0;
    if (g == 0 && cok[J]) gsa[(long)(n - 1) * m + 16 * J + c] = as_c[J];
  }
  // last smoothed moments = last filtered ones
  for (int e = lane; e < mm; e += KF_WAVE) gsP[(long)(n - 1) * mm + e] = gfP[(long)(n - 1) * mm + e];
  bool bad = false;

  for (int t = n - 2; t >= 0; --t) {
    // T and P_f are re-read per step (L2 hits) rather than held across the Gauss-Jordan: with them
    // resident the kernel spills (3 x 3 blocks: ~200 VGPRs to scratch)
    double TA[NB][KSR];  // TA[J][s] = T[16 J + c][4 s + g]: A operand "T", B operand "T'"
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s) TA[J][s] = (cok[J] && rok(s)) ? gT[(16 * J + c) * m + 4 * s + g] : 0.0;
    double af_c[NB], diff_c[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) af_c[J] = cok[J] ? gfa[(long)t * m + 16 * J + c] : 0.0;
    {
      double afr[KSRP];
      to_rows(af_c, afr);
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        double part = TA[J][0] * afr[0];
#pragma unroll
        for (int s = 1; s < KSR; ++s) part = fma(TA[J][s], afr[s], part);
        diff_c[J] = as_c[J] - col_allsum(part);  // a_s[t+1] - T a_f,  (T a_f)_j = sum_k T[j][k] a_k
      }
    }
    double4_t W2[NB][NB];
    {
      double PF[NB][KSR];
      load_cov(gfP + (long)t * mm, PF);
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
            W2[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(PF[I][ks], TA[J][ks], W2[I][J], 0, 0, 0);
    }

    // ---- W2 = P_f T' (above) ; P_hat = T W2 + RQR' (blocks I <= J) ------------------------------------
    double4_t PHv[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          PHv[I][J][u] = (4 * I + u < KSR) ? RQL[(J * KSR + ((4 * I + u < KSR) ? 4 * I + u : 0)) * KF_WAVE + lane] : 0.0;
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J)
          PHv[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(TA[I][ks], W2[ks >> 2][J][ks & 3], PHv[I][J], 0, 0, 0);
    // W = W2' through LDS: element (i, j) of W2 at TR[i * LD + pos(j)]
    double Wt[NB][KSR], PH[NB][KSR], D[NB][KSR];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
#pragma unroll
      for (int s = 0; s < KSR; ++s)
        if (j < 4 * KSR) TR[(4 * s + g) * LD + pos(j)] = W2[s >> 2][J][s & 3];
    }
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
      double tw[KSRP];
      read_rows(TR + (j < 4 * KSR ? j : 4 * KSR - 1) * LD, tw);
#pragma unroll
      for (int s = 0; s < KSR; ++s) Wt[J][s] = cok[J] ? tw[s] : 0.0;
    }
    wave_lds_fence();
    // P_hat: blocks above the diagonal as computed, below mirrored, DIAGONAL blocks (S + S') / 2 -- the
    // tile must be exactly symmetric: every symmetric matrix here doubles as its own transpose (A
    // operand), and with mirroring an antisymmetric residue in the diagonal blocks is fed back as a
    // symmetric error each step (measured: errors growing 1.45x per step backwards in time)
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
#pragma unroll
      for (int s = 0; s < 4 * (J + 1) && s < KSR; ++s)
        if (j < 4 * KSR) TR[(4 * s + g) * LD + pos(j)] = PHv[s >> 2][J][s & 3];
    }
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int j = 16 * J + c;
      double tw[KSRP];
      read_rows(TR + (j < 4 * KSR ? j : 4 * KSR - 1) * LD, tw);
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        const int I = s >> 2;
        double ph;
        if (I < J) ph = PHv[I < J ? I : 0][J][s & 3];
        else if (I == J) ph = cok[J] ? 0.5 * (PHv[I == J ? I : 0][J][s & 3] + tw[s]) : 0.0;
        else ph = cok[J] ? tw[s] : 0.0;
        PH[J][s] = ph;
        D[J][s] = PS[J][s] - ph;  // P_s[t+1] - P_hat
      }
    }
    wave_lds_fence();
    // pivot threshold: 1e-9 x the largest diagonal entry of P_hat
    double dg = 0.0;
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s) dg = (16 * J + c == 4 * s + g) ? fmax(dg, PH[J][s]) : dg;
    const double tol = 1e-9 * wave_allmax(dg);

    // ---- Gauss-Jordan: (P_hat | W) -> (I | X) ------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 4 * KSR; ++k) {
      if (k < m) {
        const int ks = k >> 2, kg = k & 3, Jk = k >> 4;
        double rowA[NB], rowW[NB];
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          if (J >= Jk) rowA[J] = group_bcast_sel(PH[J][ks], kg);
          rowW[J] = group_bcast_sel(Wt[J][ks], kg);
        }
        const double d = readlane_f64(rowA[Jk], k & 15);
        bad = bad || !(d > tol);
        const double invd = fast_rcp(d);
        // pivot column (row-indexed): the lanes that hold column k publish their registers
        if (c == (k & 15)) {
#pragma unroll
          for (int s = 0; s < KSR; ++s) EX[g * KSRP + s] = PH[Jk][s];
        }
        wave_lds_fence();
        double colk[KSRP];
        read_rows(EX, colk);
        wave_lds_fence();
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          const double rw = rowW[J] * invd;
          const double ra = (J >= Jk) ? rowA[J >= Jk ? J : Jk] * invd : 0.0;
#pragma unroll
          for (int s = 0; s < KSR; ++s) {
            const bool piv = (s == ks) && (g == kg);  // the pivot row itself becomes row / d
            const double nW = fma(-colk[s], rw, Wt[J][s]);
            Wt[J][s] = piv ? rw : nW;
            if (J >= Jk) {  // column blocks left of the pivot are identity already
              const double nA = fma(-colk[s], ra, PH[J][s]);
              PH[J][s] = piv ? ra : nA;
            }
          }
        }
      }
    }
    // Wt = X = P_hat^-1 T P_f

    // ---- P_s[t] = P_f + X' (D X) -------------------------------------------------------------------------------------
    double4_t E[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = 0; J < NB; ++J) E[I][J] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J)
          E[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(D[I][ks], Wt[J][ks], E[I][J], 0, 0, 0);
    double4_t PSv[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J)
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // seed: P_f, re-read (see above)
          const int s = (4 * I + u < KSR) ? 4 * I + u : 0;
          PSv[I][J][u] = (4 * I + u < KSR && cok[J] && rok(s)) ? gfP[(long)t * mm + obase + 4 * s * m + 16 * J] : 0.0;
        }
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J)
          PSv[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[I][ks], E[ks >> 2][J][ks & 3], PSv[I][J], 0, 0, 0);

    // ---- a_s[t] = a_f + X' (a_s[t+1] - T a_f) ----------------------------------------------------------------------------
    double dfr[KSRP];
    to_rows(diff_c, dfr);
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      double px = Wt[J][0] * dfr[0];
#pragma unroll
      for (int s = 1; s < KSR; ++s) px = fma(Wt[J][s], dfr[s], px);
      as_c[J] = af_c[J] + col_allsum(px);  // (X' diff)_j = sum_i X[i][j] diff_i
      if (g == 0 && cok[J]) gsa[(long)t * m + 16 * J + c] = as_c[J];
    }

    // ---- smoothed covariance: blocks above the diagonal as computed, below mirrored, diagonal blocks
    //      symmetrised (see P_hat above); compact image -> HBM ------------------------------------------------
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        const int I = s >> 2;
        if (I <= J && cok[J] && rok(s)) {
          const double v = PSv[I <= J ? I : 0][J][s & 3];
          TR[(4 * s + g) * m + 16 * J + c] = v;
          if (I < J) TR[(16 * J + c) * m + 4 * s + g] = v;
        }
      }
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 0; s < KSR; ++s) {
        const int I = s >> 2;
        const bool on = cok[J] && rok(s);
        if (I < J) PS[J][s] = on ? PSv[I < J ? I : 0][J][s & 3] : 0.0;
        else if (I == J) PS[J][s] = on ? 0.5 * (PSv[I == J ? I : 0][J][s & 3] + TR[(16 * J + c) * m + 4 * s + g]) : 0.0;
        else PS[J][s] = on ? TR[obase + 4 * s * m + 16 * J] : 0.0;
      }
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int s = 4 * J; s < 4 * (J + 1) && s < KSR; ++s)
        if (cok[J] && rok(s)) TR[(4 * s + g) * m + 16 * J + c] = PS[J][s];
    flush_image(gsP + (long)t * mm);
  }
  if (lane == 0) flags[b] = bad ? 1 : 0;
}

template <int NB, int KSR>
void launch_sm_nb(const KfSmoothArgs& a, int32_t* flags, hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)NbSmLayout<NB, KSR>::doubles(a.m, a.r);
  hipLaunchKernelGGL((kf_smooth_mfma_nb_kernel<NB, KSR>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a, flags);
}

}  // namespace

bool kf_smoother_mfma_nb_ok(int m, int r) {
  // the fallback for flagged draws is the LDS-resident smoother: it has to fit as well
  return m > 16 && m <= KF_NB_M_MAX && kf_smoother_lds_fits(m) &&
         sizeof(double) * (size_t)(2 * m * r + r * r) <= 64 * 1024;
}

int kf_launch_smoother_mfma_nb(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
  if (!kf_smoother_mfma_nb_ok(a.m, a.r) || !workspace || workspace_bytes < kf_smoother_mfma_workspace_bytes(a.B)) {
    KF_SET_ERR(h, "smoother(mfma_nb): needs 16 < m <= %d and %zu bytes of workspace", KF_NB_M_MAX,
               kf_smoother_mfma_workspace_bytes(a.B));
    return KF_ERR_UNSUPPORTED;
  }
  int32_t* flags = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  switch ((a.m + 3) / 4) {
    case 5: launch_sm_nb<2, 5>(a, flags, stream); break;
    case 6: launch_sm_nb<2, 6>(a, flags, stream); break;
    case 7: launch_sm_nb<2, 7>(a, flags, stream); break;
    case 8: launch_sm_nb<2, 8>(a, flags, stream); break;
    case 9: launch_sm_nb<3, 9>(a, flags, stream); break;
    case 10: launch_sm_nb<3, 10>(a, flags, stream); break;
    case 11: launch_sm_nb<3, 11>(a, flags, stream); break;
    default: launch_sm_nb<3, 12>(a, flags, stream); break;
  }
  KF_HIP_CHECK(h, hipGetLastError());
  // draws whose P_hat was not safely positive definite: pinv path, other draws exit at once
  KfSmoothArgs fb = a;
  fb.only_flagged = flags;
  return kf_launch_smoother_lds(h, fb, stream);
}
