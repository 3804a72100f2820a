// kf_forward_b4c.hip -- the time-chunked forward pipeline for SMALL batches (p == 1, m <= 10, static matrices, no missing observation):
// the two kernels either side of the block kernel's chunk instantiation (kf_forward_b4.hip, SS == 3).
//
// What it is for.  The reference evaluates one draw at a time (core/statespace.py:230-236 around the scan of
// filters/kalman_filter.py:126-163); this library's batch axis fills the chip from ~4096 draws on.  One GPU's share of the north-star
// batch is 512 draws -- one wavefront per TWO SIMDs -- and a launch then lasts as long as one wavefront's chain of T dependent steps
// (1000 x 0.58 us = 0.58 ms whatever the batch: the >= 6x strong-scaling clause of BASELINE.json stood at 2.7x, profiles/r05/strong_proxy.md).
// The chain can be cut: for a static model without missing observations the covariance sequence (predict :215-223, update :255-284) does
// not depend on y, and the map of the FILTERED covariance over s steps is ONE element (A, C, J) of the associative operator of
// Sarkka & Garcia-Fernandez (2021, "Temporal parallelization of Bayesian smoothers", Lemma 8),
//     Phi(P) = A (I + P J)^-1 P A' + C,
// raised to the s-th power.  Because every step's element is the SAME, Phi_s costs log2(s) SQUARINGS per draw -- the structure-
// preserving doubling iteration of the Riccati literature -- not one combine per time step (what DESIGN section 6 costed and rejected):
//     one step:   S0 = Z Qt Z' + H,  K0 = Qt Z' / S0,  A1 = (I - K0 Z) T,  C1 = (I - K0 Z) Qt,  J1 = (Z T)' (Z T) / S0      (Qt = R Q R')
//     squaring:   D = (I + C J)^-1,  A <- A D A,  C <- A D C A' + C,  J <- A' J D A + J
//     applying:   D = (I + P_f J)^-1,  Psi = A D,  P_f <- Psi P_f A' + C      and the MEAN of the same steps:  a_f <- Psi a_f + (zero-start response)
// (needs S0 > 0 -- H > 0 or state noise the observation sees: every model of the reference; else the flag below).  Prototype and error
// budget: scripts/r06/chunk_proto.py (chunk-start covariances within 1e-15 of the sequential recursion on the BASELINE workloads, a
// P0 = 1e6 I start included).
//
// Pipeline (kf_launch_forward_b4, kf_forward_b4.hip):
//   1. kf_b4c_starts_kernel   one wavefront per draw: the doubling (plain fp64 FMAs on LDS-resident m x m matrices, Gauss-Jordan with
//                             partial pivoting for D), then P_f(k s) for k = 1 .. nc-1 one after the other; leaves the predicted
//                             covariance T P_f(k s) T' + Qt every chunk starts from and Psi_k
//   2. kf_fwd_b4_kernel<.., SS = 3>   B x nc wavefronts: chunk k of a draw runs steps [k s + 1, (k + 1) s + 1) (chunk 0 from step 0)
//                             -- the unchanged arithmetic of the sequential kernel -- with its mean started at c ("zero start"):
//                             covariance tapes final; gains K_t, F_t and the chunk's last filtered mean go to the workspace
//   3. kf_b4c_mean_kernel     one workgroup per draw, four chunks per wavefront: a_f(k s) = Psi_{k-1} a_f((k-1) s) + z_{k-1} chains the
//                             chunks (nc matrix-vector products), then every chunk runs the mean recursion from its own start with the
//                             stored gains (one m x m matrix-vector product per step), writes the state tapes and the likelihood
// Self-check instead of trust: every chunk compares the covariance it ENDS with against the one its successor STARTED from (1e-12 of the
// largest entry); a mismatch, a missing observation, S0 <= 0 or an F <= 0 raises a device flag and the sequential kernel, launched
// behind with run_if = flag, redoes the batch -- no host synchronisation, the call stays asynchronous.
// KF_FLAG_EXACT_RECURSION (or KF_FORWARD_CHUNKED=0 in the environment) keeps the sequential kernel: the chunked covariances are the
// recursion's own up to rounding (<= 1e-13 relative), but not bit for bit.
#include <type_traits>

#include "kf_b4.h"
#include "kf_b4_split.h"

namespace {

constexpr int B4C_MAXM = 10;   // (the block kernel's ring-streamed range, NB <= 3)
constexpr int B4C_NT = 256;    // threads of the doubling kernel's workgroup: an element of a 10 x 20 elimination tableau each

// ---- small dense helpers on LDS-resident M x M matrices (row-major, leading dimension M), one workgroup of B4C_NT threads per draw; M is a
// compile-time constant: every loop unrolls, the LDS reads of a product are issued back to back, an element per thread (first version:
// run-time m on one wavefront, 183 us for 512 draws; unrolled 125; four wavefronts: see profiles/r06)
// out = op(X) op(Y) [+ E], element per lane and round.  `out` may alias an operand: every lane reads all it needs before any lane writes
template <int M, bool TX, bool TY>
__device__ __forceinline__ void b4c_mul(double* out, const double* X, const double* Y, const double* E, int tid) {
  constexpr int MM = M * M, RD = (MM + B4C_NT - 1) / B4C_NT;
  double acc[RD];
#pragma unroll
  for (int rd = 0; rd < RD; ++rd) {
    int e = tid + B4C_NT * rd;
    const bool on = MM % B4C_NT == 0 || e < MM;
    e = on ? e : 0;
    const int i = e / M, j = e - i * M;
    double a0 = E ? E[e] : 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
      a0 = fma(TX ? X[k * M + i] : X[i * M + k], TY ? Y[j * M + k] : Y[k * M + j], a0);
      if (k + 1 < M) a1 = fma(TX ? X[(k + 1) * M + i] : X[i * M + k + 1], TY ? Y[j * M + k + 1] : Y[(k + 1) * M + j], a1);
    }
    acc[rd] = a0 + a1;
  }
  __syncthreads();
#pragma unroll
  for (int rd = 0; rd < RD; ++rd) {
    const int e = tid + B4C_NT * rd;
    if (MM % B4C_NT == 0 || e < MM) out[e] = acc[rd];
  }
  __syncthreads();
}
// X <- (X + X') / 2
template <int M>
__device__ __forceinline__ void b4c_sym(double* X, int tid) {
  constexpr int MM = M * M, RD = (MM + B4C_NT - 1) / B4C_NT;
  double v[RD];
#pragma unroll
  for (int rd = 0; rd < RD; ++rd) {
    int e = tid + B4C_NT * rd;
    e = e < MM ? e : 0;
    const int i = e / M, j = e - i * M;
    v[rd] = 0.5 * (X[e] + X[j * M + i]);
  }
  __syncthreads();
#pragma unroll
  for (int rd = 0; rd < RD; ++rd) {
    const int e = tid + B4C_NT * rd;
    if (e < MM) X[e] = v[rd];
  }
  __syncthreads();
}
// D <- (I + X Y)^-1 by Gauss-Jordan with partial pivoting on the augmented [I + X Y | I] (AUG: M x 2M); false: singular
template <int M>
__device__ __forceinline__ bool b4c_inv_ipxy(double* D, double* AUG, const double* X, const double* Y, int tid) {
  constexpr int MM = M * M, W2 = 2 * M, RD = (MM + B4C_NT - 1) / B4C_NT, RE = (M * W2 + B4C_NT - 1) / B4C_NT;
#pragma unroll
  for (int rd = 0; rd < RD; ++rd) {
    const int e = tid + B4C_NT * rd;
    if (e < MM) {
      const int i = e / M, j = e - i * M;
      double a0 = (i == j) ? 1.0 : 0.0, a1 = 0.0;
#pragma unroll
      for (int k = 0; k < M; k += 2) {
        a0 = fma(X[i * M + k], Y[k * M + j], a0);
        if (k + 1 < M) a1 = fma(X[i * M + k + 1], Y[(k + 1) * M + j], a1);
      }
      AUG[i * W2 + j] = a0 + a1;
      AUG[i * W2 + M + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  bool ok = true;
#pragma unroll 1
  for (int k = 0; k < M; ++k) {
    // pivot: the largest |entry| of column k in rows k .. M-1 -- a max over the sixteen lanes of DPP row 0 (four row rotations), then
    // the first lane that holds it
    const int lane = tid & 63;   // (every wavefront searches for itself: the same column, the same verdict)
    const double mine = (lane >= k && lane < M) ? fabs(AUG[lane * W2 + k]) : -1.0;
    double v = mine;
    v = fmax(v, dpp_mov<0x128>(v));
    v = fmax(v, dpp_mov<0x124>(v));
    v = fmax(v, dpp_mov<0x122>(v));
    v = fmax(v, dpp_mov<0x121>(v));
    const double pv = uniform_f64(v);
    if (!(pv > 0.0)) {
      ok = false;
      break;
    }
    const unsigned long long hit = __ballot(lane < M && mine == pv);
    const int p = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)hit) - 1);
    // One read phase, one write phase (LDS is in order within the wavefront).  With q = (old row p) / pivot:
    //   new row k = q;   new row p (p != k) = (old row k) - (old row k)[k] q;   every other row i: (old row i) - (old row i)[k] q
    const double piv = AUG[p * W2 + k];
    const double rpiv = fast_rcp(fabs(piv)) * (piv < 0.0 ? -1.0 : 1.0);
    double nv[RE];
#pragma unroll
    for (int rd = 0; rd < RE; ++rd) {
      int e = tid + B4C_NT * rd;
      e = e < M * W2 ? e : 0;
      const int i = e / W2, j = e - i * W2;
      const int src = (i == p) ? k : i;              // the row whose OLD values become row i (rows k and p trade places)
      const double q = AUG[p * W2 + j] * rpiv;
      const double x = AUG[src * W2 + j], f = AUG[src * W2 + k];
      nv[rd] = (i == k) ? q : fma(-f, q, x);
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < RE; ++rd) {
      const int e = tid + B4C_NT * rd;
      if (e < M * W2) AUG[e] = nv[rd];
    }
    __syncthreads();
  }
#pragma unroll
  for (int rd = 0; rd < RD; ++rd) {
    const int e = tid + B4C_NT * rd;
    if (e < MM) {
      const int i = e / M, j = e - i * M;
      D[e] = AUG[i * W2 + M + j];
    }
  }
  __syncthreads();
  return ok;
}

// ---- 1. chunk-start covariances by doubling --------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(B4C_NT) void kf_b4c_starts_kernel(const KfFwdArgs A, const B4cWs W, int s, int nc) {
  constexpr int MM = M * M;
  __shared__ double sT[MM], sQ[MM], sA[MM], sC[MM], sJ[MM], sD[MM], sX[MM], sY[MM], sV[MM], sP[MM], sAUG[2 * MM], sv[4 * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r;
  double* sZ = sv;
  double* sK0 = sv + 16;
  double* sZT = sv + 32;
  double* sMv = sv + 48;
  // a missing observation makes the covariance path depend on the data: the sequential kernel's business
  {
    const double* yb = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
    bool nan = false;
    for (int e = lane; e < n; e += KF_WAVE) {   // (every wavefront scans the whole series: the same verdict in each)
      const double yv = yb[e];
      nan = nan || (yv != yv);
    }
    if (__ballot(nan) != 0ull) {
      if (tid == 0) *W.flag = 1;
      return;
    }
  }
  {
    const double* gT = A.Tm + b * (long)MM;
    const double* gR = A.R + b * (long)M * r_;
    const double* gQ = A.Q + b * (long)r_ * r_;
    const double* gP = A.P0 + b * (long)MM;
    for (int e = tid; e < MM; e += B4C_NT) {
      const int i = e / M, j = e - i * M;
      sT[e] = gT[e];
      sP[e] = (j >= i) ? gP[e] : gP[j * M + i];   // the upper triangle, as b4_setup reads P0
      double acc = 0.0;
      for (int l = 0; l < r_; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r_; ++k) rq = fma(gR[i * r_ + k], gQ[k * r_ + l], rq);
        acc = fma(rq, gR[j * r_ + l], acc);
      }
      sQ[e] = acc;
    }
    if (tid < M) sZ[tid] = A.Z[b * (long)M + tid];
  }
  __syncthreads();
  b4c_sym<M>(sQ, tid);
  const double H1 = A.H[b];
  // ---- the one-step element --------------------------------------------------------------------------------------------------------
  if (tid < M) {
    double qz = 0.0, zt = 0.0;
#pragma unroll
    for (int l = 0; l < M; ++l) {
      qz = fma(sQ[tid * M + l], sZ[l], qz);
      zt = fma(sZ[l], sT[l * M + tid], zt);
    }
    sK0[tid] = qz;   // Qt Z'
    sZT[tid] = zt;   // Z T
  }
  __syncthreads();
  double S0 = H1;
#pragma unroll
  for (int l = 0; l < M; ++l) S0 = fma(sZ[l], sK0[l], S0);
  if (!(S0 > 0.0)) {
    if (tid == 0) *W.flag = 1;
    return;
  }
  const double rS0 = 1.0 / S0;
  for (int e = tid; e < MM; e += B4C_NT) {
    const int i = e / M, j = e - i * M;
    const double k0 = sK0[i] * rS0;
    sA[e] = fma(-k0, sZT[j], sT[e]);          // (I - K0 Z) T
    sC[e] = fma(-k0, sK0[j], sQ[e]);          // Qt - Qt Z' Z Qt / S0
    sJ[e] = sZT[i] * sZT[j] * rS0;
  }
  __syncthreads();
  b4c_sym<M>(sC, tid);
  // ---- log2(s) squarings ----------------------------------------------------------------------------------------------------------------
#pragma unroll 1
  for (int k = 1; k < s; k <<= 1) {
    if (!b4c_inv_ipxy<M>(sD, sAUG, sC, sJ, tid)) {
      if (tid == 0) *W.flag = 1;
      return;
    }
    b4c_mul<M, false, false>(sX, sA, sD, nullptr, tid);   // X = A D
    b4c_mul<M, false, false>(sY, sX, sC, nullptr, tid);   // Y = A D C
    b4c_mul<M, false, false>(sV, sJ, sD, nullptr, tid);   // V = J D
    b4c_mul<M, true, false>(sD, sA, sV, nullptr, tid);    // D = A' J D
    b4c_mul<M, false, false>(sJ, sD, sA, sJ, tid);        // J <- A' J D A + J
    b4c_mul<M, false, true>(sC, sY, sA, sC, tid);         // C <- A D C A' + C
    b4c_mul<M, false, false>(sA, sX, sA, nullptr, tid);   // A <- A D A
    b4c_sym<M>(sC, tid);
    b4c_sym<M>(sJ, tid);
  }
  // ---- P_f(0): the update of step 0 ------------------------------------------------------------------------------------------------------
  if (tid < M) {
    double mv = 0.0;
#pragma unroll
    for (int l = 0; l < M; ++l) mv = fma(sP[tid * M + l], sZ[l], mv);
    sMv[tid] = mv;
  }
  __syncthreads();
  double F0 = H1;
#pragma unroll
  for (int l = 0; l < M; ++l) F0 = fma(sZ[l], sMv[l], F0);
  if (!(F0 > 0.0)) {
    if (tid == 0) *W.flag = 1;
    return;
  }
  {
    const double rF = 1.0 / F0;
    constexpr int RD = (MM + B4C_NT - 1) / B4C_NT;
    double v[RD];
#pragma unroll
    for (int rd = 0; rd < RD; ++rd) {
      int e = tid + B4C_NT * rd;
      e = e < MM ? e : 0;
      const int i = e / M, j = e - i * M;
      v[rd] = fma(-sMv[i] * rF, sMv[j], sP[e]);
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < RD; ++rd) {
      const int e = tid + B4C_NT * rd;
      if (e < MM) sP[e] = v[rd];
    }
    __syncthreads();
  }
  // ---- P_f(k s), k = 1 .. nc - 1, and what chunk k starts from ------------------------------------------------------------------------------
#pragma unroll 1
  for (int k = 1; k < nc; ++k) {
    if (!b4c_inv_ipxy<M>(sD, sAUG, sP, sJ, tid)) {
      if (tid == 0) *W.flag = 1;
      return;
    }
    b4c_mul<M, false, false>(sX, sA, sD, nullptr, tid);   // Psi_{k-1} = A D: a_f(k s) = Psi_{k-1} a_f((k-1) s) + ...
    b4c_mul<M, false, false>(sY, sX, sP, nullptr, tid);   // Psi P_f
    b4c_mul<M, false, true>(sP, sY, sA, sC, tid);         // P_f(k s)
    b4c_sym<M>(sP, tid);
    b4c_mul<M, false, false>(sY, sT, sP, nullptr, tid);   // T P_f
    b4c_mul<M, false, true>(sD, sY, sT, sQ, tid);         // T P_f T' + Qt
    b4c_sym<M>(sD, tid);
    double* gs = W.start + ((long)b * nc + k) * MM;
    double* gp = W.psi + ((long)b * nc + k - 1) * MM;
    for (int e = tid; e < MM; e += B4C_NT) {
      gs[e] = sD[e];
      gp[e] = sX[e];
    }
  }
}

// lane L of every DPP row -> all sixteen lanes of its row (row_newbcast: one v_mov_dpp per dword, no trip through the LDS crossbar)
template <int LSRC>
__device__ __forceinline__ double row_bcast(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x150 + LSRC, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x150 + LSRC, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int I = 0, typename F>
__device__ __forceinline__ void sfor10(F&& f) {
  if constexpr (I < B4C_MAXM) {
    f(std::integral_constant<int, I>{});
    sfor10<I + 1>(f);
  }
}
// ---- 1b. the same on 16 x 16 fp64 MFMA tiles, ONE wavefront per draw, nothing in LDS (round 6, second version) --------------------------------
// Every matrix is a tile in the accumulator layout of v_mfma_f64_16x16x4 (kf_tile.h: lane = 16 g + c, register s holds X[4 s + g][c]);
// a tile used as the A operand is its transpose, so  tn(U, V) = U' V  is KS = ceil(m / 4) instructions and no matrix is ever moved:
//     squaring:  D = (I + C J)^-1   Xt = tn(D, At)   A <- tn(Xt, A)   At <- tn(A, Xt)   C <- tn(tn(C, Xt), At) + C   J <- tn(tn(tn(J, D), A), A) + J
//     applying:  D = (I + P J)^-1   Psi' = tn(D, At)   Psi = tn(At, D)   P <- tn(tn(P, Psi'), At) + C   start = tn(tn(P, Tt), Tt) + Qt
// (A and A' are both carried; C, J, P, Qt are symmetric and re-symmetrised with one more product, X' = tn(X, I)).  The inverse is a
// Gauss-Jordan elimination with IMPLICIT partial pivoting in registers: column k by row index is one DPP broadcast per register, the
// pivot search a max over the four lane groups (v_permlane swaps), the pivot row a select + a group broadcast; rows are never
// exchanged -- the left tile ends as a permutation Pi and the inverse is tn(Pi, right tile).  No LDS round trip, no barrier on the
// chain: 110 us (four wavefronts on LDS, a barrier per pivot phase) -> see profiles/r06/.
template <int KS>
__device__ __forceinline__ double4_t b4c_tn(const double4_t& U, const double4_t& V, double4_t acc) {
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(U[s], V[s], acc, 0, 0, 0);
  return acc;
}
// all-reduce max over the four lane groups (lanes c, c + 16, c + 32, c + 48)
__device__ __forceinline__ double b4c_col_allmax(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
  lo = __double2loint(x);
  hi = __double2hiint(x);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return fmax(__hiloint2double(d[0], c[0]), __hiloint2double(d[1], c[1]));
}
__device__ __forceinline__ int b4c_col_allmin_i(int x) {
  auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  x = min((int)a[0], (int)a[1]);
  auto c = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return min((int)c[0], (int)c[1]);
}
// D = M^-1 for the leading m x m block of a tile (rows / columns >= m: identity); false: singular
template <int KS, int K = 0>
__device__ __forceinline__ void b4c_gj_pivots(double4_t& Lt, double4_t& Rt, unsigned& used, bool& ok, int m, int g) {
  if constexpr (K < 4 * KS) {
    if (K < m) {   // (wave-uniform)
      double f[KS];
      double best = -1.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        f[s] = row_bcast<K>(Lt[s]);                        // M[4 s + g][K]
        const bool cand = 4 * s + g < m && !((used >> (4 * s + g)) & 1u);
        best = fmax(best, cand ? fabs(f[s]) : -1.0);
      }
      const double vmax = b4c_col_allmax(best);
      int mine = 64;
#pragma unroll
      for (int s = KS - 1; s >= 0; --s) {
        const bool cand = 4 * s + g < m && !((used >> (4 * s + g)) & 1u);
        if (cand && fabs(f[s]) == vmax) mine = 4 * s + g;
      }
      const int p = __builtin_amdgcn_readfirstlane(b4c_col_allmin_i(mine));
      if (!(vmax > 0.0) || p >= 64) ok = false;
      const int sp = p >> 2, gp = p & 3;
      double rl = Lt[0], rr = Rt[0];
#pragma unroll
      for (int s = 1; s < KS; ++s) {
        rl = (sp == s) ? Lt[s] : rl;
        rr = (sp == s) ? Rt[s] : rr;
      }
      rl = group_bcast_sel(rl, gp);                        // row p of the left / right tile, by column
      rr = group_bcast_sel(rr, gp);
      const double piv = row_bcast<K>(rl);
      const double rp = fast_rcp(fabs(piv)) * (piv < 0.0 ? -1.0 : 1.0);
      const double ql = rl * rp, qr = rr * rp;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bool isp = 4 * s + g == p;
        Lt[s] = isp ? ql : fma(-f[s], ql, Lt[s]);
        Rt[s] = isp ? qr : fma(-f[s], qr, Rt[s]);
      }
      used |= 1u << (p & 31);
    }
    b4c_gj_pivots<KS, K + 1>(Lt, Rt, used, ok, m, g);
  }
}
template <int KS>
__device__ __forceinline__ bool b4c_tile_inverse(const double4_t& M, const double4_t& Id, double4_t& D, int m, int g) {
  double4_t Lt = M, Rt = Id;
  unsigned used = 0u;
  bool ok = true;
  b4c_gj_pivots<KS>(Lt, Rt, used, ok, m, g);
  const double4_t z = {0.0, 0.0, 0.0, 0.0};
  D = b4c_tn<KS>(Lt, Rt, z);   // Pi' (Pi M^-1 ... ) = M^-1
  return ok;
}

template <int KS>
__global__ __launch_bounds__(KF_WAVE) void kf_b4c_starts_mfma_kernel(const KfFwdArgs A, const B4cWs W, int s_len, int nc) {
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const long b = blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m, r_ = A.r;
  const double4_t z4 = {0.0, 0.0, 0.0, 0.0};
  {
    const double* yb = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
    bool nan = false;
    for (int e = lane; e < n; e += KF_WAVE) {
      const double yv = yb[e];
      nan = nan || (yv != yv);
    }
    if (__ballot(nan) != 0ull) {
      if (lane == 0) *W.flag = 1;
      return;
    }
  }
  const double* gT = A.Tm + b * (long)mm;
  const double* gP = A.P0 + b * (long)mm;
  // R Q through LDS for r <= 16 (first version: R and Q read from global memory inside the r x r loops -- a chain of dependent loads;
  // kept for wider state noise, which no model of the reference has)
  __shared__ double sR[12 * 16 + 16 * 16], sRQ[12 * 16];   // R [m][r], then Q [r][r]; R Q [m][r]
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  const bool rq_lds = r_ <= 16;
  if (rq_lds) {
    double* sQ = sR + m * r_;
    for (int e = lane; e < m * r_; e += KF_WAVE) sR[e] = gR[e];
    for (int e = lane; e < r_ * r_; e += KF_WAVE) sQ[e] = gQ[e];
    wave_lds_fence();
    for (int e = lane; e < m * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double rq = 0.0;
      for (int k = 0; k < r_; ++k) rq = fma(sR[i * r_ + k], sQ[k * r_ + l], rq);
      sRQ[e] = rq;
    }
    wave_lds_fence();
  }
  double4_t Tt, Tn, Qt, Pm, Id;
  double Zr[KS];
  const double Zc = c < m ? A.Z[b * (long)m + c] : 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = 4 * s + g;
    const bool on = s < KS && row < m && c < m;
    Tn[s] = on ? gT[row * m + c] : 0.0;
    Tt[s] = on ? gT[c * m + row] : 0.0;
    Pm[s] = on ? (c >= row ? gP[row * m + c] : gP[c * m + row]) : 0.0;   // the upper triangle, as b4_setup reads P0
    Id[s] = (s < KS && row == c) ? 1.0 : 0.0;
    double acc = 0.0;
    if (on && rq_lds) {
      for (int l = 0; l < r_; ++l) acc = fma(sRQ[row * r_ + l], sR[c * r_ + l], acc);
    } else if (on) {
      for (int l = 0; l < r_; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r_; ++k) rq = fma(gR[row * r_ + k], gQ[k * r_ + l], rq);
        acc = fma(rq, gR[c * r_ + l], acc);
      }
    }
    Qt[s] = acc;
    if (s < KS) Zr[s] = (row < m) ? A.Z[b * (long)m + row] : 0.0;
  }
  auto symm = [&](double4_t& X) {
    const double4_t Xt = b4c_tn<KS>(X, Id, z4);
#pragma unroll
    for (int s = 0; s < KS; ++s) X[s] = 0.5 * (X[s] + Xt[s]);
  };
  symm(Qt);
  const double H1 = A.H[b];
  // ---- the one-step element: vectors by row index (row sums) and by column index (column sums) ----------------------------------------
  double qzr[KS], ztr[KS], part_q = 0.0, part_t = 0.0;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    qzr[s] = row_allsum(Qt[s] * Zc);        // (Qt Z')[4 s + g]
    ztr[s] = row_allsum(Tt[s] * Zc);        // (Z T)[4 s + g] = sum_c T[c][4 s + g] Z[c]
    part_q = fma(Zr[s], Qt[s], part_q);
    part_t = fma(Zr[s], Tn[s], part_t);
  }
  const double qzc = col_allsum(part_q);    // (Qt Z')[c]   (Qt symmetric)
  const double ztc = col_allsum(part_t);    // (Z T)[c]
  const double S0 = H1 + row_allsum(qzc * Zc);
  if (!(S0 > 0.0)) {
    if (lane == 0) *W.flag = 1;
    return;
  }
  const double rS0 = 1.0 / S0;
  double4_t Am = z4, At = z4, Cm = z4, Jm = z4;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    Am[s] = fma(-qzr[s] * rS0, ztc, Tn[s]);        // (I - K0 Z) T
    At[s] = fma(-ztr[s] * rS0, qzc, Tt[s]);        // its transpose
    Cm[s] = fma(-qzr[s] * rS0, qzc, Qt[s]);        // Qt - Qt Z' Z Qt / S0
    Jm[s] = ztr[s] * ztc * rS0;
  }
  // ---- log2(s) squarings ----------------------------------------------------------------------------------------------------------------
  bool ok = true;
#pragma unroll 1
  for (int k = 1; k < s_len; k <<= 1) {
    double4_t D;
    ok = b4c_tile_inverse<KS>(b4c_tn<KS>(Cm, Jm, Id), Id, D, m, g) && ok;
    const double4_t Xt = b4c_tn<KS>(D, At, z4);                 // (A D)'
    const double4_t Vv = b4c_tn<KS>(Jm, D, z4);                 // J D
    const double4_t Yt = b4c_tn<KS>(Cm, Xt, z4);                // (A D C)'
    const double4_t Wt = b4c_tn<KS>(Vv, Am, z4);                // (A' J D)'
    Cm = b4c_tn<KS>(Yt, At, Cm);                                // A D C A' + C
    Jm = b4c_tn<KS>(Wt, Am, Jm);                                // A' J D A + J
    const double4_t An = b4c_tn<KS>(Xt, Am, z4);                // A D A
    At = b4c_tn<KS>(Am, Xt, z4);                                // its transpose
    Am = An;
    symm(Cm);
    symm(Jm);
  }
  // ---- P_f(0): the update of step 0 ------------------------------------------------------------------------------------------------------
  {
    double part = 0.0, mvr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      mvr[s] = row_allsum(Pm[s] * Zc);
      part = fma(Zr[s], Pm[s], part);
    }
    const double mvc = col_allsum(part);
    const double F0 = H1 + row_allsum(mvc * Zc);
    if (!(F0 > 0.0)) ok = false;
    const double rF = 1.0 / F0;
#pragma unroll
    for (int s = 0; s < KS; ++s) Pm[s] = fma(-mvr[s] * rF, mvc, Pm[s]);
  }
  // ---- P_f(k s), k = 1 .. nc - 1, and what chunk k starts from ------------------------------------------------------------------------------
#pragma unroll 1
  for (int k = 1; k < nc; ++k) {
    double4_t D;
    ok = b4c_tile_inverse<KS>(b4c_tn<KS>(Pm, Jm, Id), Id, D, m, g) && ok;
    const double4_t Pst = b4c_tn<KS>(D, At, z4);                // Psi' = (A D)'
    const double4_t Psi = b4c_tn<KS>(At, D, z4);                // Psi = A D: a_f(k s) = Psi_{k-1} a_f((k-1) s) + ...
    const double4_t Yt = b4c_tn<KS>(Pm, Pst, z4);               // (Psi P_f)'
    Pm = b4c_tn<KS>(Yt, At, Cm);                                // P_f(k s)
    symm(Pm);
    const double4_t Z1 = b4c_tn<KS>(Pm, Tt, z4);                // (T P_f)'
    double4_t St = b4c_tn<KS>(Z1, Tt, Qt);                      // T P_f T' + Qt
    symm(St);
    double* gs = W.start + ((long)b * nc + k) * mm;
    double* gp = W.psi + ((long)b * nc + k - 1) * mm;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int row = 4 * s + g;
      if (row < m && c < m) {
        gs[row * m + c] = St[s];
        gp[row * m + c] = Psi[s];
      }
    }
  }
  if (!ok && lane == 0) *W.flag = 1;
}

// ---- 3. the mean pass ----------------------------------------------------------------------------------------------------------------------
constexpr int MB = 36;   // steps per LDS block (chunks of 2^k and 2^k + 1 steps: no block with a single step)
struct B4cMeanLds {
  // per chunk: K block [MB][10], the filtered / predicted means of the block [MB][10] each, per-step scalars
  // [MB][4] = {y -+ d, 1 / F, log F, v^2 / F}
  static constexpr int LDK = B4C_MAXM;
  static constexpr int O_K = 0, O_AF = MB * LDK, O_AP = 2 * MB * LDK, O_S = 3 * MB * LDK, PER_CHUNK = O_S + 4 * MB;
};

// One workgroup per draw, FOUR chunks per wavefront -- one per DPP row (first version: one chunk per wavefront, every row a replica:
// 4096 wavefronts x ~70 instructions per step were issue bound, 125 us; the row sums and broadcasts below are per row anyway).
// The mean recursion itself (kalman_filter.py:219, :272 at p = 1) with the gains the chunk kernel left:
//     v_t = (y_t -+ d) - Z a_t,   a_f = a_t + K_t v_t,   a_{t+1} = T a_f + c
// from the chunk's own start  a_{k s + 1} = T a_f(k s) + c,  a_f(k s) chained over the chunks before it (second version: the chunk
// kernel wrote zero-start trajectories to the state tapes and this kernel added the homogeneous part -- a read-modify-write of both
// tapes whose loads it waited for, 106 us; now the state tapes are written once, here).
__global__ __launch_bounds__(128) void kf_b4c_mean_kernel(const KfFwdArgs A, const B4cWs W, int s, int nc) {
  using L = B4cMeanLds;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = (int)(blockDim.x >> 6);
  const long b = blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m;
  double* part = smem + (size_t)(4 * nw) * L::PER_CHUNK;   // [nw] partial log-likelihoods
  const bool run = *W.flag == 0;
  if (run) {
    const int g = lane >> 4, i = lane & 15;   // DPP row = chunk slot, state index
    const int cj = 4 * wave + g;              // this row's chunk (>= nc: idle row)
    const bool act = cj < nc;
    double* blk = smem + (size_t)cj * L::PER_CHUNK;
    const int t0 = !act ? 0 : (cj ? cj * s + 1 : 0);
    const int t1 = !act ? 0 : ((cj + 1) * s + 1 < n ? (cj + 1) * s + 1 : n);
    const bool ion = i < m;
    double Tr[B4C_MAXM];              // row i of T
    const double* gT = A.Tm + b * (long)mm;
#pragma unroll
    for (int l = 0; l < B4C_MAXM; ++l) Tr[l] = (ion && l < m) ? gT[i * m + l] : 0.0;
    const double Zi = ion ? A.Z[b * (long)m + i] : 0.0;
    const double ci = (ion && A.c) ? A.c[b * (long)m + i] : 0.0;
    const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // kalman_filter.py:335 quirk: y_hat = Z a - d
    const double dd = A.d ? dsign * A.d[b] : 0.0;
    auto matvec = [&](const double (&Mr)[B4C_MAXM], double x, double acc0) {   // acc0 + (row i of a matrix) . x, x spread over the row's lanes
      double a0 = acc0, a1 = 0.0;
      sfor10([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        const double xb = row_bcast<l>(x);
        if (l & 1) a1 = fma(Mr[l], xb, a1);
        else a0 = fma(Mr[l], xb, a0);
      });
      return a0 + a1;
    };
    // ---- the predicted mean this chunk starts from: a0, or T a_f(cj s) + c with a_f(cj s) chained over the chunks before ---------------
    double a = ion ? A.a0[b * (long)m + i] : 0.0;
    {
      double af = ion ? W.zend[((long)b * nc + 0) * m + i] : 0.0;   // a_f(s): chunk 0 ran from (a0, P0)
      const int kmax = 4 * wave + 3 < nc - 1 ? 4 * wave + 3 : nc - 1;
      for (int k = 1; k < kmax; ++k) {
        const double* psi = W.psi + ((long)b * nc + k) * mm;         // a_f((k + 1) s) = Psi_k a_f(k s) + z_k
        double pr[B4C_MAXM];
#pragma unroll
        for (int l = 0; l < B4C_MAXM; ++l) pr[l] = (ion && l < m) ? psi[i * m + l] : 0.0;
        const double nx = matvec(pr, af, ion ? W.zend[((long)b * nc + k) * m + i] : 0.0);
        af = (k < cj) ? nx : af;      // (a row whose chunk starts earlier keeps its own)
      }
      const double as = matvec(Tr, af, ci);   // T a_f(cj s) + c
      a = (act && cj > 0) ? as : a;
    }
    const bool w_fa = A.filt_a != nullptr, w_pa = A.pred_a != nullptr;
    const double* gK = W.K + b * (long)n * m;
    const double* gfv = W.fv + b * (long)n * 2;
    const double* gy = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
    double* gfa = w_fa ? A.filt_a + b * (long)n * m : nullptr;
    double* gpa = w_pa ? A.pred_a + b * (long)(n + 1) * m : nullptr;
    if (w_pa && act && cj == 0 && ion) gpa[i] = a;   // predicted slot 0 = a0
    double lsum = 0.0;
    constexpr int NE = (MB * B4C_MAXM + 15) / 16;   // elements of a block per lane of its row
    constexpr int NS = (MB + 15) / 16;              // steps of a block per lane of its row
    int steps_max = t1 - t0;                        // the longest chunk of this wavefront (wave-uniform loop bound)
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) steps_max = max(steps_max, __shfl_xor(steps_max, off, KF_WAVE));
    // the next block's gains, F and y are fetched into registers while this block's steps run
    double pK[NE], pF[NS], pY[NS];
    auto fetch = [&](int tb) {
      const int cnt = min(max(t1 - tb, 0), MB);
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int e = i + 16 * k;
        pK[k] = (e < cnt * m) ? gK[(long)tb * m + e] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const bool on = i + 16 * k < cnt;
        pF[k] = on ? gfv[2 * ((long)tb + i + 16 * k)] : 1.0;
        pY[k] = on ? gy[(long)tb + i + 16 * k] : 0.0;
      }
    };
    fetch(t0);
    for (int o = 0; o < steps_max; o += MB) {
      const int tb = t0 + o;
      const int cnt = min(max(t1 - tb, 0), MB);
      // block in (each row its own chunk's block): gains, per-step scalars
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int e = i + 16 * k;
        if (e < cnt * m) blk[L::O_K + e] = pK[k];       // (rows of m doubles, packed as on the tape)
      }
#pragma unroll
      for (int k = 0; k < NS; ++k)
        if (i + 16 * k < cnt) {
          double4_t sc = {pY[k] + dd, 1.0 / pF[k], log(pF[k]), 0.0};
          *reinterpret_cast<double4_t*>(blk + L::O_S + 4 * (i + 16 * k)) = sc;
        }
      wave_lds_fence();
      if (o + MB < steps_max) fetch(tb + MB);
      // one step ahead: the block's values of step st + 1 are read while step st computes
      double Kt = blk[L::O_K + i];
      double2_t yr = *reinterpret_cast<const double2_t*>(blk + L::O_S);   // {y -+ d, 1 / F}
      for (int st = 0; st < MB; ++st) {
        const int sn = st + 1 < MB ? st + 1 : st;
        const double Kn = blk[L::O_K + sn * m + i];
        const double2_t yrn = *reinterpret_cast<const double2_t*>(blk + L::O_S + 4 * sn);
        const bool on = st < cnt;
        const double za = row_allsum(Zi * a);           // Z a_t
        const double v = yr.x - za;                     // the innovation
        const double af = ion ? fma(Kt, v, a) : 0.0;    // a_t + K_t v_t
        const double an = matvec(Tr, af, ci);           // T a_f + c
        if (ion && on) {
          blk[L::O_AF + st * m + i] = af;
          blk[L::O_AP + st * m + i] = an;
        }
        if (i == 0 && on) blk[L::O_S + 4 * st + 3] = v * v * yr.y;
        a = on ? an : a;
        Kt = Kn;
        yr = yrn;
      }
      wave_lds_fence();
      // block out: the state tapes (stores only), the log-likelihood terms
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const int e = i + 16 * k;
        if (e < cnt * m) {
          if (w_fa) gfa[(long)tb * m + e] = blk[L::O_AF + e];
          if (w_pa) gpa[(long)(tb + 1) * m + e] = blk[L::O_AP + e];
        }
      }
#pragma unroll
      for (int k = 0; k < NS; ++k)
        if (i + 16 * k < cnt) {
          const int st = i + 16 * k;
          const double llv = -0.5 * (KF_LOG_2PI + blk[L::O_S + 4 * st + 2] + blk[L::O_S + 4 * st + 3]);
          if (A.ll_obs) A.ll_obs[b * (long)n + tb + st] = llv;
          lsum += llv;
        }
      wave_lds_fence();
    }
    lsum = col_allsum(row_allsum(lsum));
    if (lane == 0) part[wave] = lsum;
  }
  __syncthreads();
  if (run && threadIdx.x == 0) {
    double lp = 0.0;
    for (int k = 0; k < nw; ++k) lp += part[k];   // (fixed order: a draw's bits do not depend on the schedule)
    A.logp[b] = lp;
    if (A.info) A.info[b] = 0;
  }
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------------------------
// shapes the pipeline takes: the ring-streamed block kernel's (p == 1, m <= 10), static matrices, no internal tapes; from 256 steps on
bool kf_forward_b4c_shape_ok(const kf_handle* h, long B, int T, int p, int m, int r, unsigned flags) {
  (void)h;
  if (p != 1 || m < 1 || m > B4C_MAXM || r < 1 || r > 64 || T < 256) return false;
  if (flags & (KF_FLAG_TV_ANY | KF_FLAG_EXACT_RECURSION)) return false;
  const char* e = kf_env_switch("KF_FORWARD_CHUNKED");
  if (e && *e) return e[0] != '0';
  // worth it while the batch leaves SIMDs idle (measured, profiles/r06/chunk_check.txt, full tapes, chunked against sequential ms:
  // 256 x 1000: 0.31 / 0.54; 512 x 1000: 0.40 / 0.58; 1024 x 1000: 0.62 / 0.60; 1024 x 2000: 1.05 / 1.19): up to three wavefronts
  // per four SIMDs, a full round of SIMDs for long series
  const long simds = (h && h->n_cu > 0 ? h->n_cu : 256) * 4L;
  return B <= (3 * simds) / 4 || (B <= simds && T >= 1536);
}

// chunk length s (a power of two: squarings only) and count: eight chunks -- one workgroup of the mean pass, four wavefronts per SIMD at
// 512 draws -- of at least 64 steps.  A function of the series length ALONE: a draw's bits must not depend on how many other draws
// share the call (sharding a batch over ranks leaves every log-likelihood bit for bit: tests/test_gpu_distributed.py)
void kf_forward_b4c_plan(const kf_handle* h, long B, int T, int* s_out, int* nc_out) {
  (void)h;
  (void)B;
  int s = 64;
  while (8 * s + 1 < T) s <<= 1;                    // chunk k covers steps [k s + 1, (k + 1) s + 1), chunk 0 from step 0
  int nc = 1;
  while (nc * s + 1 < T) ++nc;
  *s_out = s;
  *nc_out = nc;
}

size_t kf_forward_b4c_carve(void* base, long B, int T, int m, int nc, B4cWs* W) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char* p = (char*)base;
  size_t off = 0;
  W->flag = (int32_t*)(p + off); off += up(64);
  W->start = (double*)(p + off); off += up(sizeof(double) * (size_t)B * nc * m * m);
  W->psi = (double*)(p + off); off += up(sizeof(double) * (size_t)B * nc * m * m);
  W->zend = (double*)(p + off); off += up(sizeof(double) * (size_t)B * nc * m);
  W->fv = (double*)(p + off); off += up(sizeof(double) * (size_t)B * T * 2);
  W->K = (double*)(p + off); off += up(sizeof(double) * (size_t)B * T * m);
  return off + 256;
}

int kf_launch_b4c_starts(kf_handle* h, const KfFwdArgs& a, const B4cWs& W, int s, int nc, hipStream_t stream) {
  KF_HIP_CHECK(h, hipMemsetAsync(W.flag, 0, 64, stream));
  // the doubling on MFMA tiles, one wavefront per draw (KF_B4C_STARTS_LDS=1 in the environment: the LDS version on four wavefronts)
  const char* le = kf_env_switch("KF_B4C_STARTS_LDS");
  if (!(le && le[0] == '1')) {
    if (a.m <= 4) hipLaunchKernelGGL(kf_b4c_starts_mfma_kernel<1>, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, W, s, nc);
    else if (a.m <= 8) hipLaunchKernelGGL(kf_b4c_starts_mfma_kernel<2>, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, W, s, nc);
    else hipLaunchKernelGGL(kf_b4c_starts_mfma_kernel<3>, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, W, s, nc);
    KF_HIP_CHECK(h, hipGetLastError());
    return KF_OK;
  }
#define KF_B4C_ST(M_) hipLaunchKernelGGL(kf_b4c_starts_kernel<M_>, dim3((unsigned)a.B), dim3(B4C_NT), 0, stream, a, W, s, nc)
  switch (a.m) {
    case 1: KF_B4C_ST(1); break;
    case 2: KF_B4C_ST(2); break;
    case 3: KF_B4C_ST(3); break;
    case 4: KF_B4C_ST(4); break;
    case 5: KF_B4C_ST(5); break;
    case 6: KF_B4C_ST(6); break;
    case 7: KF_B4C_ST(7); break;
    case 8: KF_B4C_ST(8); break;
    case 9: KF_B4C_ST(9); break;
    default: KF_B4C_ST(10); break;
  }
#undef KF_B4C_ST
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

int kf_launch_b4c_means(kf_handle* h, const KfFwdArgs& a, const B4cWs& W, int s, int nc, hipStream_t stream) {
  const int nw = (nc + 3) / 4;   // four chunks per wavefront (nc <= 8: at most two)
  const size_t lds = sizeof(double) * ((size_t)(4 * nw) * B4cMeanLds::PER_CHUNK + 8);
  static KfAttrOnce once;
  if (once.need(h->device)) {
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_b4c_mean_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(sizeof(double) * (8 * B4cMeanLds::PER_CHUNK + 8))));
    once.mark(h->device);
  }
  hipLaunchKernelGGL(kf_b4c_mean_kernel, dim3((unsigned)a.B), dim3((unsigned)(KF_WAVE * nw)), lds, stream, a, W, s, nc);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
