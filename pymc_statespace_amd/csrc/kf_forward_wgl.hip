// kf_forward_wgl.hip -- forward filter for 64 < k_states <= 96 with the covariance RESIDENT IN LDS: one workgroup (four wavefronts,
// one per SIMD) per draw, running all T steps; scalar observation (p == 1), static matrices, no covariance tapes (log-likelihood,
// per-step terms, filtered / predicted STATES only -- the log-likelihood-only call a sampler makes).
//
// Reference recursion (kalman_filter.py:255-284 update, :231-252 predict), as kf_forward_big.hip runs it for large m:
//     m = P z     F = z' m + H     v = y - d - z' a     k = m / F     a_f = a + k v     logp_t = -(log 2 pi + v^2 / F) / 2 - log sqrt F
//     P_f = P - k m' - m k' + (k F) k'        a <- T a_f + c        P <- T P_f T' + R Q R'
// Why: the HBM-workspace path moves P, P_f / W between kernels through HBM / L2 -- ~300 KB per draw and step at m = 80, 1.2 TB for
// 4096 draws x 1000 steps: bound by that traffic and by three launches per step at 0.83 s (scripts/r03/mid_m_bench.py).  Here P
// (m = 96: 74.5 KB) and W = T P_f (another 74.5 KB) stay in the CU's 160 KB of LDS for the whole series, T (one draw per CU: 256
// draws x 72 KB in the L2s) is the only operand read from outside, and nothing but the log-likelihood goes back.
//
// Products on v_mfma_f64_16x16x4_f64.  The m x m output is NT x NT tiles of 16 x 16 (NT = 5: m <= 80, NT = 6: m <= 96); wavefront
// w owns tile row w and a share of tile rows 4 (and 5) -- 7 / 6 / 6 / 6 tiles at NT = 5, 9 each at NT = 6 (2 x 2 blocks of 3 x 3
// would leave one wavefront 9 of NT = 5's 25).  Both products have T as their LEFT factor (W = T P_f, S = T W': P_f is symmetric), and
// a wavefront needs only the T fragments of its tile rows (its own, 4, 5): 40 - 72 doubles per lane, loaded ONCE and kept in registers
// for the whole series together with the R Q R' tiles of its outputs -- one wavefront per SIMD has 512 registers per lane to itself.
// Inside the loop over time only LDS is read.  (First version: T from L2 every step, S = W T' with T as the right factor: 33 us per
// step at m = 80, 13.6 of them the second product with its 20 loads and 64 selects per chunk; scripts/r03/wgl_time.py.)  The k index
// inside a chunk of 16 is permuted -- lane group g = lane / 16 takes k = 4 g + s at sub-step s, the LDS operand follows -- so that
// a lane's four fragment values of a chunk are four CONSECUTIVE doubles of a row of T.  P_f is formed on the way from LDS into the
// right-factor fragments; a_hat = T a_f rides along with W.
#include <type_traits>

#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int WL_NT_THREADS = 256;

template <int NT>
struct WlLds {
  static constexpr int MP = 16 * NT, LD = MP + 1;
  static constexpr int OFF_P = 0, OFF_W = MP * LD, OFF_V = 2 * MP * LD;
  // vectors: a, a_f, k, m, k F, z, c; then scalars
  static constexpr int V_A = 0, V_AF = 1, V_K = 2, V_M = 3, V_KF = 4, V_Z = 5, V_C = 6, NVEC = 7;
  static constexpr int OFF_S = OFF_V + NVEC * MP;
  static constexpr int TOTAL = OFF_S + 16;
};

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wl_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NT>
__global__ __launch_bounds__(WL_NT_THREADS) void kf_fwd_wgl_kernel(const KfFwdArgs A, double* __restrict__ RQR_ws) {
  using L = WlLds<NT>;
  constexpr int MP = L::MP, LD = L::LD;
  constexpr int NE = NT == 5 ? 2 : 3;   // extra tiles per wavefront beyond its own tile row (NT = 5: wavefront 0 has two, the rest one)
  constexpr int NR = NT == 5 ? 2 : 3;   // tile rows a wavefront touches: its own, 4 (and 5)
  extern __shared__ double sm[];
  double* Pm = sm + L::OFF_P;
  double* Wm = sm + L::OFF_W;
  double* va = sm + L::OFF_V + L::V_A * MP;
  double* vaf = sm + L::OFF_V + L::V_AF * MP;
  double* vk = sm + L::OFF_V + L::V_K * MP;
  double* vm = sm + L::OFF_V + L::V_M * MP;
  double* vkf = sm + L::OFF_V + L::V_KF * MP;
  double* vz = sm + L::OFF_V + L::V_Z * MP;
  double* vc = sm + L::OFF_V + L::V_C * MP;
  double* sc = sm + L::OFF_S;  // [0] v, [1] F, [2] sqrt F
  const long b = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int m = A.m, n = A.T, r = A.r;
  const long mm = (long)m * m;
  const double* Tb = A.Tm + b * mm;
  double* RQRb = RQR_ws + b * mm;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double dq = A.d ? A.d[b] : 0.0;
  const double Hq = A.H[b];
  const double* yb = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);

  // ---- the tiles of this wavefront: slot c < NT = (w, c); extras e: tile number w + 4 e of rows 4.. in row-major order ----------------
  int er[NE], ec[NE];
  bool eon[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = w + 4 * e;
    eon[e] = idx < (NT - 4) * NT;
    er[e] = eon[e] ? 4 + idx / NT : 4;
    ec[e] = eon[e] ? idx % NT : 0;
  }
  // tile rows whose fragments this wavefront loads: its own, 4, (5); a_hat of tile row R is written by wavefront R (R < 4), 0 (R = 4), 1 (R = 5)
  int rrow[NR];
  rrow[0] = w;
  rrow[1] = 4;
  if (NR > 2) rrow[2] = 5;

  // ---- setup: P0 (zero-padded), vectors, R Q R' into the workspace, slot 0 of the predicted-state tape -------------------------------
  for (int e = tid; e < MP * MP; e += WL_NT_THREADS) {
    const int i = e / MP, j = e - i * MP;
    Pm[i * LD + j] = (i < m && j < m) ? A.P0[b * mm + (long)i * m + j] : 0.0;
  }
  for (int e = tid; e < MP; e += WL_NT_THREADS) {
    const bool on = e < m;
    const double a0 = on ? A.a0[b * m + e] : 0.0;
    va[e] = a0;
    vz[e] = on ? A.Z[b * m + e] : 0.0;
    vc[e] = (on && A.c) ? A.c[b * m + e] : 0.0;
    vaf[e] = vk[e] = vm[e] = vkf[e] = 0.0;
    if (on && A.pred_a) A.pred_a[b * (long)(n + 1) * m + e] = a0;
  }
  {
    const double* gR = A.R + b * (long)m * r;
    const double* gQ = A.Q + b * (long)r * r;
    for (long e = tid; e < mm; e += WL_NT_THREADS) {
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double s = 0.0;
      for (int l = 0; l < r; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
        s += rq * gR[j * r + l];
      }
      RQRb[e] = s;
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---- resident in registers for the whole series: the T fragments of the tile rows this wavefront multiplies with (lane (li, lk)
  // holds T[16 R + li][16 kc + 4 lk + 0..3]) and the R Q R' tiles of its output tiles.  Both products have T as their LEFT factor:
  // W = T P_f and S = T W' (P_f is symmetric), so nothing but LDS is read inside the loop over time.
  double tfr[NR][NT][4];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int row = 16 * rrow[q] + li;
    const double* tr = Tb + (long)min(row, m - 1) * m;
#pragma unroll
    for (int kc = 0; kc < NT; ++kc)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = 16 * kc + 4 * lk + s;
        const double tv = tr[min(kk, m - 1)];
        tfr[q][kc][s] = (row < m && kk < m) ? tv : 0.0;
      }
  }
  auto rqr_tile = [&](int R, int C) {
    double4_t v;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int row = 16 * R + lk + 4 * rg, col = 16 * C + li;
      v[rg] = (row < m && col < m) ? RQRb[(long)row * m + col] : 0.0;
    }
    return v;
  };
  // The second product S = T W' + R Q R' is symmetric: 15 of the 25 tiles are computed -- wavefront 0: (0,0..3), 1: (1,1..4),
  // 2: (2,2..4) and (4,0), 3: (3,3), (3,4), (4,4): at most four per wavefront instead of seven -- and written to both triangles.
  static_assert(NT == 5, "the tile tables of the symmetric product are those of 5 x 5 tiles (NT = 6: kf_fwd_wgl_range_kernel)");
  auto sym_tile = [&](int i, int& R, int& C) {  // slot i of this wavefront (run-time form of the tables in the product below)
    if (w == 0) { R = 0; C = i; }
    else if (w == 1) { R = 1; C = 1 + i; }
    else if (w == 2) { R = i < 3 ? 2 : 4; C = i < 3 ? 2 + i : 0; }
    else { R = i < 2 ? 3 : 4; C = i < 2 ? 3 + i : 4; }
  };
  double4_t rq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int R, C;
    sym_tile(i, R, C);
    rq[i] = rqr_tile(R, C);
  }

  double logp = 0.0;
  int info = 0;

  for (int t = 0; t < n; ++t) {
    // ---- (a) m = P z: thread (row, half) sums 8 NT columns; the halves meet in the scratch (W is free here) ----------------------------
    {
      const int row = tid % 128, half = tid / 128;
      if (row < MP) {
        const double* pr = Pm + row * LD + half * (MP / 2);
        const double* zz = vz + half * (MP / 2);
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
        for (int c = 0; c < MP / 2; c += 2) {
          s0 = fma(pr[c], zz[c], s0);
          s1 = fma(pr[c + 1], zz[c + 1], s1);
        }
        Wm[half * MP + row] = s0 + s1;
      }
    }
    __syncthreads();
    // ---- (b) wavefront 0: F, v, the log-likelihood term ----------------------------------------------------------------------------------
    if (w == 0) {
      const double yv = yb[t];
      const bool obs = !(yv != yv);
      double pf = 0.0, pv = 0.0;
      for (int e = lane; e < MP; e += 64) {
        const double mv = obs ? Wm[e] + Wm[MP + e] : 0.0;
        vm[e] = mv;
        pf = fma(vz[e], mv, pf);
        pv = fma(vz[e], va[e], pv);
      }
      pf = wl_wave_sum(pf);
      pv = wl_wave_sum(pv);
      if (lane == 0) {
        const double F = obs ? pf + Hq : 1.0;
        const double v = obs ? (yv + dsign * dq) - pv : 0.0;
        const double dj = sqrt(F);
        const double wv = v / dj;
        const double ll = obs ? (-0.5 * (KF_LOG_2PI + wv * wv) - log(dj)) : 0.0;
        logp += ll;
        if (A.ll_obs) A.ll_obs[b * (long)n + t] = ll;
        if (!(F > 0.0) && info == 0) info = ((t + 1) << 8) | 1;
        sc[0] = v;
        sc[1] = F;
        sc[2] = dj;
      }
    }
    __syncthreads();
    // ---- (c) k = m / F (two divisions by sqrt F, as the Cholesky solve of the general path), k F, a_f ------------------------------------
    {
      const double v = sc[0], F = sc[1], dj = sc[2];
      for (int e = tid; e < MP; e += WL_NT_THREADS) {
        const double kk = (vm[e] / dj) / dj;
        vk[e] = kk;
        vkf[e] = kk * F;
        const double af = va[e] + kk * v;
        vaf[e] = af;
        if (e < m && A.filt_a) A.filt_a[(b * (long)n + t) * m + e] = af;
      }
    }
    __syncthreads();
    // ---- (e) W = T P_f, a_hat = T a_f + c.  P_f = P - k m' - m k' + (k F) k' is formed on the way from LDS into the right-factor
    // fragments (big_pf_kernel's order of operations; a missing observation has k = m = 0) ---------------------------------------------------
    double4_t acc[NT + NE];
#pragma unroll
    for (int sidx = 0; sidx < NT + NE; ++sidx) acc[sidx] = (double4_t){0.0, 0.0, 0.0, 0.0};
    {
      double kcol[NT + NE], mcol[NT + NE];
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        kcol[c] = vk[16 * c + li];
        mcol[c] = vm[16 * c + li];
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        kcol[NT + e] = vk[16 * ec[e] + li];
        mcol[NT + e] = vm[16 * ec[e] + li];
      }
      double ah[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) ah[q] = 0.0;
#pragma unroll
      for (int kc = 0; kc < NT; ++kc) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = 16 * kc + 4 * lk + s;
          const double* prow = Pm + kk * LD + li;
          const double kr = vk[kk], mr = vm[kk], kfr = vkf[kk], afk = vaf[kk];
#pragma unroll
          for (int q = 0; q < NR; ++q) ah[q] = fma(tfr[q][kc][s], afk, ah[q]);
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            double bv = prow[16 * c];
            bv -= kr * mcol[c];
            bv -= mr * kcol[c];
            bv += kfr * kcol[c];
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(tfr[0][kc][s], bv, acc[c], 0, 0, 0);
          }
#pragma unroll
          for (int e = 0; e < NE; ++e) {  // (a wavefront without this extra repeats tile (4, 0) into an accumulator nobody stores)
            double bv = prow[16 * ec[e]];
            bv -= kr * mcol[NT + e];
            bv -= mr * kcol[NT + e];
            bv += kfr * kcol[NT + e];
            const double ae = (NR > 2 && er[e] == 5) ? tfr[NR - 1][kc][s] : tfr[1][kc][s];
            acc[NT + e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, bv, acc[NT + e], 0, 0, 0);
          }
        }
      }
      // W tiles -> LDS: D[row = lk + 4 rg][col = li]
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Wm[(16 * w + lk + 4 * rg) * LD + 16 * c + li] = acc[c][rg];
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (eon[e]) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) Wm[(16 * er[e] + lk + 4 * rg) * LD + 16 * ec[e] + li] = acc[NT + e][rg];
        }
      // a_hat: sum over the four lane groups; written by the owner of the tile row
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const double sv = col_allsum(ah[q]);
        const int R = rrow[q];
        const bool owner = q == 0 ? true : (R == 4 ? w == 0 : w == 1);
        const int row = 16 * R + li;
        if (owner && lk == 0 && R < NT) {
          const double an = sv + vc[row];
          va[row] = row < m ? an : 0.0;
          if (row < m && A.pred_a) A.pred_a[(b * (long)(n + 1) + t + 1) * m + row] = an;
        }
      }
    }
    __syncthreads();
    // ---- (f) P <- T W' + R Q R' (= T P_f T' + R Q R': P_f is symmetric), the upper-triangle tiles only; the accumulators start from
    // R Q R'.  One instantiation per wavefront: which row fragment and which tile column a slot uses are compile-time constants ----
    auto sym_product = [&](auto wtag) {
      constexpr int WV = decltype(wtag)::value;
      constexpr int SQ[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 1}, {0, 0, 1, 1}};   // 0: the own tile row, 1: tile row 4
      constexpr int SC[4][4] = {{0, 1, 2, 3}, {1, 2, 3, 4}, {2, 3, 4, 0}, {3, 4, 4, 4}};
      constexpr int NS = WV == 3 ? 3 : 4;
      double4_t a2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a2[i] = rq[i];
#pragma unroll
      for (int kc = 0; kc < NT; ++kc) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = 16 * kc + 4 * lk + s;
          const double* wrow = Wm + li * LD + kk;   // W'[kk][16 C + li] = W[16 C + li][kk]
#pragma unroll
          for (int i = 0; i < NS; ++i)
            a2[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(tfr[SQ[WV][i]][kc][s], wrow[16 * SC[WV][i] * LD], a2[i], 0, 0, 0);
        }
      }
      // (P is not read by this product: the new tiles go straight in, each into both triangles)
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int R = SQ[WV][i] ? 4 : WV, C = SC[WV][i];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          Pm[(16 * R + lk + 4 * rg) * LD + 16 * C + li] = a2[i][rg];
          if (R != C) Pm[(16 * C + li) * LD + 16 * R + lk + 4 * rg] = a2[i][rg];
        }
      }
    };
    switch (w) {
      case 0: sym_product(std::integral_constant<int, 0>{}); break;
      case 1: sym_product(std::integral_constant<int, 1>{}); break;
      case 2: sym_product(std::integral_constant<int, 2>{}); break;
      default: sym_product(std::integral_constant<int, 3>{}); break;
    }
    __syncthreads();
  }
  if (tid == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

// ---- the same scheme for NT = 6 (80 < m <= 96): wavefront w of NW takes the tiles [NT^2 w / NW, NT^2 (w + 1) / NW) in row-major order --
// nine tiles out of exactly two tile rows at NW = 4, so 48 doubles of T per lane instead of the 72 that the three tile rows of the
// kernel above need (its NT = 6 instance spilled 234 registers; eight wavefronts of 4 - 5 tiles, two per SIMD, spilled 491 of 256).
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void kf_fwd_wgl_range_kernel(const KfFwdArgs A, double* __restrict__ RQR_ws) {
  using L = WlLds<NT>;
  constexpr int MP = L::MP, LD = L::LD, NTH = 64 * NW;
  constexpr int TOT = NT * NT, MAXT = (TOT + NW - 1) / NW;
  extern __shared__ double sm[];
  double* Pm = sm + L::OFF_P;
  double* Wm = sm + L::OFF_W;
  double* va = sm + L::OFF_V + L::V_A * MP;
  double* vaf = sm + L::OFF_V + L::V_AF * MP;
  double* vk = sm + L::OFF_V + L::V_K * MP;
  double* vm = sm + L::OFF_V + L::V_M * MP;
  double* vkf = sm + L::OFF_V + L::V_KF * MP;
  double* vz = sm + L::OFF_V + L::V_Z * MP;
  double* vc = sm + L::OFF_V + L::V_C * MP;
  double* sc = sm + L::OFF_S;
  const long b = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int m = A.m, n = A.T, r = A.r;
  const long mm = (long)m * m;
  const double* Tb = A.Tm + b * mm;
  double* RQRb = RQR_ws + b * mm;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double dq = A.d ? A.d[b] : 0.0;
  const double Hq = A.H[b];
  const double* yb = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);

  static_assert(NT == 6 && NW == 4, "the tile tables below are those of 36 tiles on four wavefronts");
  // wavefront w: tiles 9 w .. 9 w + 8 = two tile rows R0, R0 + 1; even w: (R0, 0..5), (R0+1, 0..2); odd w: (R0, 3..5), (R0+1, 0..5).
  // The two shapes are two instantiations of the products below (everything about a slot is a compile-time constant: with the row
  // of a slot chosen at run time the compiler hoisted 216 selected fragments out of the loop over time and spilled 530 registers).
  const int R0 = (9 * w) / NT;

  for (int e = tid; e < MP * MP; e += NTH) {
    const int i = e / MP, j = e - i * MP;
    Pm[i * LD + j] = (i < m && j < m) ? A.P0[b * mm + (long)i * m + j] : 0.0;
  }
  for (int e = tid; e < MP; e += NTH) {
    const bool on = e < m;
    const double a0 = on ? A.a0[b * m + e] : 0.0;
    va[e] = a0;
    vz[e] = on ? A.Z[b * m + e] : 0.0;
    vc[e] = (on && A.c) ? A.c[b * m + e] : 0.0;
    vaf[e] = vk[e] = vm[e] = vkf[e] = 0.0;
    if (on && A.pred_a) A.pred_a[b * (long)(n + 1) * m + e] = a0;
  }
  {
    const double* gR = A.R + b * (long)m * r;
    const double* gQ = A.Q + b * (long)r * r;
    for (long e = tid; e < mm; e += NTH) {
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double s = 0.0;
      for (int l = 0; l < r; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
        s += rq * gR[j * r + l];
      }
      RQRb[e] = s;
    }
  }
  __threadfence_block();
  __syncthreads();

  double tfr[2][NT][4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 16 * min(R0 + q, NT - 1) + li;
    const double* tr = Tb + (long)min(row, m - 1) * m;
#pragma unroll
    for (int kc = 0; kc < NT; ++kc)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = 16 * kc + 4 * lk + s;
        const double tv = tr[min(kk, m - 1)];
        tfr[q][kc][s] = (row < m && kk < m) ? tv : 0.0;
      }
  }

  double logp = 0.0;
  int info = 0;
  for (int t = 0; t < n; ++t) {
    // (a) m = P z
    {
      const int row = tid % 128, half = (tid / 128) & 1;
      if (row < MP && tid < 256) {
        const double* pr = Pm + row * LD + half * (MP / 2);
        const double* zz = vz + half * (MP / 2);
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
        for (int c = 0; c < MP / 2; c += 2) {
          s0 = fma(pr[c], zz[c], s0);
          s1 = fma(pr[c + 1], zz[c + 1], s1);
        }
        Wm[half * MP + row] = s0 + s1;
      }
    }
    __syncthreads();
    // (b) F, v, the log-likelihood term
    if (w == 0) {
      const double yv = yb[t];
      const bool obs = !(yv != yv);
      double pf = 0.0, pv = 0.0;
      for (int e = lane; e < MP; e += 64) {
        const double mv = obs ? Wm[e] + Wm[MP + e] : 0.0;
        vm[e] = mv;
        pf = fma(vz[e], mv, pf);
        pv = fma(vz[e], va[e], pv);
      }
      pf = wl_wave_sum(pf);
      pv = wl_wave_sum(pv);
      if (lane == 0) {
        const double F = obs ? pf + Hq : 1.0;
        const double v = obs ? (yv + dsign * dq) - pv : 0.0;
        const double dj = sqrt(F);
        const double wv = v / dj;
        const double ll = obs ? (-0.5 * (KF_LOG_2PI + wv * wv) - log(dj)) : 0.0;
        logp += ll;
        if (A.ll_obs) A.ll_obs[b * (long)n + t] = ll;
        if (!(F > 0.0) && info == 0) info = ((t + 1) << 8) | 1;
        sc[0] = v;
        sc[1] = F;
        sc[2] = dj;
      }
    }
    __syncthreads();
    // (c) k, k F, a_f
    {
      const double v = sc[0], F = sc[1], dj = sc[2];
      for (int e = tid; e < MP; e += NTH) {
        const double kk = (vm[e] / dj) / dj;
        vk[e] = kk;
        vkf[e] = kk * F;
        const double af = va[e] + kk * v;
        vaf[e] = af;
        if (e < m && A.filt_a) A.filt_a[(b * (long)n + t) * m + e] = af;
      }
    }
    __syncthreads();
    auto products = [&](auto vtag) {
      constexpr int WV = decltype(vtag)::value;   // the wavefront
      constexpr int V = WV & 1;                   // the shape of its nine tiles in the first product
      constexpr bool OWN0 = V == 0;   // a_hat of tile row R: the wavefront whose range holds tile (R, 0)
#define TQ(i) ((V == 0) ? ((i) >= 6 ? 1 : 0) : ((i) >= 3 ? 1 : 0))
#define TC(i) ((V == 0) ? ((i) >= 6 ? (i) - 6 : (i)) : ((i) >= 3 ? (i) - 3 : (i) + 3))
    // (e) W = T P_f, a_hat = T a_f + c
    double4_t acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    {
      double kcol[MAXT], mcol[MAXT];
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        kcol[i] = vk[16 * TC(i) + li];
        mcol[i] = vm[16 * TC(i) + li];
      }
      double ah0 = 0.0, ah1 = 0.0;
#pragma unroll
      for (int kc = 0; kc < NT; ++kc) {
        wave_lds_fence();  // (bounds how far ahead the LDS reads are hoisted: registers)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = 16 * kc + 4 * lk + s;
          const double* prow = Pm + kk * LD + li;
          const double kr = vk[kk], mr = vm[kk], kfr = vkf[kk], afk = vaf[kk];
          ah0 = fma(tfr[0][kc][s], afk, ah0);
          ah1 = fma(tfr[1][kc][s], afk, ah1);
#pragma unroll
          for (int i = 0; i < MAXT; ++i) {
            double bv = prow[16 * TC(i)];
            bv -= kr * mcol[i];
            bv -= mr * kcol[i];
            bv += kfr * kcol[i];
            const double ae = TQ(i) ? tfr[1][kc][s] : tfr[0][kc][s];
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, bv, acc[i], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < MAXT; ++i)
        {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) Wm[(16 * (R0 + TQ(i)) + lk + 4 * rg) * LD + 16 * TC(i) + li] = acc[i][rg];
        }
      const double s0 = col_allsum(ah0), s1 = col_allsum(ah1);
      if (lk == 0) {
        if (OWN0) {
          const int row = 16 * R0 + li;
          const double an = s0 + vc[row];
          va[row] = row < m ? an : 0.0;
          if (row < m && A.pred_a) A.pred_a[(b * (long)(n + 1) + t + 1) * m + row] = an;
        }
        if (true) {
          const int row = 16 * (R0 + 1) + li;
          const double an = s1 + vc[row];
          va[row] = row < m ? an : 0.0;
          if (row < m && A.pred_a) A.pred_a[(b * (long)(n + 1) + t + 1) * m + row] = an;
        }
      }
    }
    __syncthreads();
    // (f) P <- T W' + R Q R', symmetric: 21 of the 36 tiles, at most six per wavefront instead of nine, written to both triangles.
    //     wavefront 0 (rows 0, 1): (0,0..5); 1 (rows 1, 2): (1,1..4), (2,2); 2 (rows 3, 4): (3,3), (3,4), (3,2), (4,2), (4,4);
    //     3 (rows 4, 5): (5,1), (5,2), (5,3), (4,5), (5,5).  R Q R' from L2 at every step.
    {
      constexpr int SQ[4][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 1, 1}, {0, 0, 0, 1, 1, 1}, {1, 1, 1, 0, 1, 1}};
      constexpr int SC[4][6] = {{0, 1, 2, 3, 4, 5}, {1, 2, 3, 4, 2, 2}, {3, 4, 2, 2, 4, 4}, {1, 2, 3, 5, 5, 5}};
      constexpr int NS = WV == 0 ? 6 : 5;
      double4_t a2[6];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = 16 * (R0 + SQ[WV][i]) + lk + 4 * rg, col = 16 * SC[WV][i] + li;
          a2[i][rg] = (row < m && col < m) ? RQRb[(long)row * m + col] : 0.0;
        }
      }
#pragma unroll
      for (int kc = 0; kc < NT; ++kc) {
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = 16 * kc + 4 * lk + s;
          const double* wrow = Wm + li * LD + kk;
#pragma unroll
          for (int i = 0; i < NS; ++i)
            a2[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(tfr[SQ[WV][i]][kc][s], wrow[16 * SC[WV][i] * LD], a2[i], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int R = R0 + SQ[WV][i], C = SC[WV][i];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          Pm[(16 * R + lk + 4 * rg) * LD + 16 * C + li] = a2[i][rg];
          if (R != C) Pm[(16 * C + li) * LD + 16 * R + lk + 4 * rg] = a2[i][rg];
        }
      }
    }
#undef TQ
#undef TC
    };
    switch (w) {
      case 0: products(std::integral_constant<int, 0>{}); break;
      case 1: products(std::integral_constant<int, 1>{}); break;
      case 2: products(std::integral_constant<int, 2>{}); break;
      default: products(std::integral_constant<int, 3>{}); break;
    }
    __syncthreads();
  }
  if (tid == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

bool kf_forward_wgl_ok(int m, int p, unsigned flags, bool cov_tapes) {
  return m > KF_WG_M_MAX && m <= 96 && p == 1 && !(flags & KF_FLAG_TV_ANY) && !cov_tapes;
}

size_t kf_forward_wgl_workspace_bytes(long B, int m) { return sizeof(double) * (size_t)B * m * m + 64; }

int kf_launch_forward_wgl(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!kf_forward_wgl_ok(a.m, a.p, a.flags, a.filt_P || a.pred_P) || !workspace ||
      workspace_bytes < kf_forward_wgl_workspace_bytes(a.B, a.m)) {
    KF_SET_ERR(h, "forward(wgl): needs 64 < m <= 96, p == 1, static matrices, no covariance tapes and %zu bytes of workspace",
               kf_forward_wgl_workspace_bytes(a.B, a.m));
    return KF_ERR_UNSUPPORTED;
  }
  double* rqr = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  if (a.m <= 80) {
    const int lds = (int)(sizeof(double) * WlLds<5>::TOTAL);
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_fwd_wgl_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((kf_fwd_wgl_kernel<5>), dim3((unsigned)a.B), dim3(WL_NT_THREADS), lds, stream, a, rqr);
  } else {
    const int lds = (int)(sizeof(double) * WlLds<6>::TOTAL);
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_fwd_wgl_range_kernel<6, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((kf_fwd_wgl_range_kernel<6, 4>), dim3((unsigned)a.B), dim3(256), lds, stream, a, rqr);
  }
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
