// kf_smoother_big.hip -- RTS smoother for LARGE state dimension (per-draw tiles do not fit LDS:
// m > ~47, e.g. BASELINE config C5, m = 367, "MFMA m x m path + RTS smoother").
//
// Same contract as kf_smooth_lds_kernel (reference pymc_statespace/filters/kalman_smoother.py:
// 56-104): backwards over t = n-2 .. 0 from (a_f[n-1], P_f[n-1]),
//     a_hat = T a_f[t]                       P_hat = T P_f[t] T' + R Q R'   (no c, not symmetrised)
//     G     = (pinv(P_hat) T P_f[t])'
//     a_s[t] = a_f[t] + G (a_s[t+1] - a_hat)  P_s[t] = P_f[t] + G (P_s[t+1] - P_hat) G'
// While P_hat is safely positive definite pinv is a Cholesky solve:  X = P_hat^-1 (T P_f),  G = X'.
// A Cholesky pivot below 1e-12 x max diag(P_hat) (or NaN) flags the draw; flagged draws are redone after
// the main pass, the whole series, by sb_pinv_redo_kernel: one 1024-thread workgroup per draw, every step
// with the Moore-Penrose inverse from a parallel-ordered Jacobi eigen-decomposition in the draw's workspace
// (numpy's cutoff 1e-15 x max |eigenvalue|, kalman_smoother.py:92) -- tens of ms per step at m = 367, rare.
//
// Everything is batched over draws in a caller-provided HBM workspace (6 B m^2 doubles + the
// inverses of the diagonal Cholesky blocks) and every m x m x m piece is an fp64-MFMA GEMM
// (v_mfma_f64_16x16x4_f64, 64 x 64 output tile per 256-thread workgroup, K staged through LDS with
// register prefetch -- the kernel of kf_forward_big.hip generalised to sub-blocks):
//     W = T P_f ; P_hat = W T' + RQR' ; D = P_s[t+1] - P_hat
//     blocked right-looking Cholesky, block 64:  potrf + inverse of the diagonal block in LDS,
//         panel  L21 = A21 inv(L11)'  (GEMM),  trailing  A22 -= L21 L21'  (GEMM)
//     forward / backward substitution by blocks:  Y_k = inv(L_kk) W_k, W_i -= L_ik Y_k ; then with
//         U = L':  X_k = inv(L_kk)' Y_k, W_i -= U_ik X_k          (all GEMMs)
//     E = D X ; G = X' ; P_s[t] = G E + P_f[t] ; a_s[t] by two matrix-vector products
// ~10.3 m^3 flops per draw-step, >97% of them inside the GEMM kernel.
#include "kf_internal.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GTM = 64, GTN = 64, GK = 32, GNT = 256;
constexpr int GWR = 2, GWC = 2;
constexpr int GTI = GTM / GWR / 16, GTJ = GTN / GWC / 16;
constexpr int GLDA = GTM + 1, GLDB = GTN + 1;
constexpr int GPA = GTM * GK / GNT, GPB = GTN * GK / GNT;
constexpr int NB = 64;  // Cholesky block

enum { EPI_STORE = 0, EPI_ADD = 1, EPI_RSUB = 2 };  // C = acc | E + acc | E - acc

// C[b] (M x N, ldc) = epi( A[b] (M x K, lda) * op(B[b]) ),  op(B) = B (K x N, ldb) or B' (B is N x K)
// grid (ceil(N/64), ceil(M/64), draws).  C may alias E; C may alias A or B only where each
// workgroup reads exactly the region it writes (see the call sites).
template <bool TRANS_B, int EPI>
__global__ __launch_bounds__(GNT) void sb_gemm_kernel(const double* A, long sA, int lda, const double* Bm, long sB,
                                                      int ldb, const double* E, long sE, int lde, double* Cm,
                                                      long sC, int ldc, int M, int N, int K) {
  __shared__ double As[GK][GLDA];
  __shared__ double Bs[GK][GLDB];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / GWC, wc = wave % GWC;
  const long b = blockIdx.z;
  const int r0 = blockIdx.y * GTM, c0 = blockIdx.x * GTN;
  const double* Ab = A + b * sA;
  const double* Bb = Bm + b * sB;

  double4_t acc[GTI][GTJ];
#pragma unroll
  for (int i = 0; i < GTI; ++i)
#pragma unroll
    for (int j = 0; j < GTJ; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int li = lane & 15, lk = lane >> 4;
  double ra[GPA], rb[GPB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < GPA; ++q) {
      const int e = tid + GNT * q;
      const int rr = e / GK, kk = e - rr * GK;
      const int gr = r0 + rr, gk = k0 + kk;
      ra[q] = (gr < M && gk < K) ? Ab[(long)gr * lda + gk] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < GPB; ++q) {
      const int e = tid + GNT * q;
      if (TRANS_B) {
        const int cc = e / GK, kk = e - cc * GK;
        const int gc = c0 + cc, gk = k0 + kk;
        rb[q] = (gc < N && gk < K) ? Bb[(long)gc * ldb + gk] : 0.0;
      } else {
        const int kk = e / GTN, cc = e - kk * GTN;
        const int gc = c0 + cc, gk = k0 + kk;
        rb[q] = (gc < N && gk < K) ? Bb[(long)gk * ldb + gc] : 0.0;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < GPA; ++q) {
      const int e = tid + GNT * q;
      const int rr = e / GK, kk = e - rr * GK;
      As[kk][rr] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < GPB; ++q) {
      const int e = tid + GNT * q;
      if (TRANS_B) {
        const int cc = e / GK, kk = e - cc * GK;
        Bs[kk][cc] = rb[q];
      } else {
        const int kk = e / GTN, cc = e - kk * GTN;
        Bs[kk][cc] = rb[q];
      }
    }
  };

  fetch(0);
  for (int k0 = 0; k0 < K; k0 += GK) {
    stage();
    __syncthreads();
    if (k0 + GK < K) fetch(k0 + GK);
#pragma unroll
    for (int ks = 0; ks < GK; ks += 4) {
      double af[GTI], bf[GTJ];
#pragma unroll
      for (int i = 0; i < GTI; ++i) af[i] = As[ks + lk][wr * (GTM / GWR) + i * 16 + li];
#pragma unroll
      for (int j = 0; j < GTJ; ++j) bf[j] = Bs[ks + lk][wc * (GTN / GWC) + j * 16 + li];
#pragma unroll
      for (int i = 0; i < GTI; ++i)
#pragma unroll
        for (int j = 0; j < GTJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  double* Cb = Cm + b * sC;
  const double* Eb = (EPI != EPI_STORE) ? E + b * sE : nullptr;
#pragma unroll
  for (int i = 0; i < GTI; ++i)
#pragma unroll
    for (int j = 0; j < GTJ; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int gr = r0 + wr * (GTM / GWR) + i * 16 + lk + 4 * rg;
        const int gc = c0 + wc * (GTN / GWC) + j * 16 + li;
        if (gr < M && gc < N) {
          double v = acc[i][j][rg];
          if (EPI == EPI_ADD) v = Eb[(long)gr * lde + gc] + v;
          if (EPI == EPI_RSUB) v = Eb[(long)gr * lde + gc] - v;
          Cb[(long)gr * ldc + gc] = v;
        }
      }
}

// RQR' per draw (m x m), once.
__global__ __launch_bounds__(256) void sb_rqr_kernel(const double* R, const double* Q, double* out, int m, int r) {
  const long b = blockIdx.y;
  const double* gR = R + b * (long)m * r;
  const double* gQ = Q + b * (long)r * r;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)m * m) return;
  const int i = (int)(e / m), j = (int)(e - (long)i * m);
  double s = 0.0;
  for (int l = 0; l < r; ++l) {
    double rq = 0.0;
    for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
    s += rq * gR[j * r + l];
  }
  out[b * (long)m * m + e] = s;
}

// out = A - B, elementwise over one m x m matrix per draw
__global__ __launch_bounds__(256) void sb_sub_kernel(const double* A, long sA, const double* Bm, long sB, double* out,
                                                     long sO, long mm) {
  const long b = blockIdx.y;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < mm) out[b * sO + e] = A[b * sA + e] - Bm[b * sB + e];
}

// dst = src' (m x m per draw), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void sb_transpose_kernel(const double* src, long sS, double* dst, long sD, int m) {
  __shared__ double tile[32][33];
  const long b = blockIdx.z;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = by + k, c = bx + tx;
    tile[k][tx] = (r < m && c < m) ? src[b * sS + (long)r * m + c] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int r = bx + k, c = by + tx;
    if (r < m && c < m) dst[b * sD + (long)r * m + c] = tile[tx][k];
  }
}

// Cholesky of the nb x nb diagonal block at (k1, k1) of A (in place, lower) and the inverse of its
// factor: LINV[b][blk] = inv(L_kk), LINVT = its transpose (64 x 64, zero padded).  One workgroup
// per draw.  A pivot below 1e-12 x scale[b] (or NaN) flags the draw for the pseudo-inverse pass.
__global__ __launch_bounds__(256) void sb_potrf_inv_kernel(double* A, long sA, int m, int k1, int nb, double* LINV,
                                                           double* LINVT, long sL, const double* scale,
                                                           int32_t* flags) {
  __shared__ double L[NB][NB + 1];
  __shared__ double I[NB][NB + 1];
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  double* Ab = A + b * sA + (long)k1 * m + k1;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, j = e - i * NB;
    L[i][j] = (i < nb && j <= i) ? Ab[(long)i * m + j] : 0.0;
    I[i][j] = 0.0;
  }
  __syncthreads();
  const double tol = 1e-12 * scale[b];
  for (int j = 0; j < nb; ++j) {
    const double piv = L[j][j];
    if (tid == 0 && !(piv > tol)) flags[b] = 1;
    const double d = sqrt(piv);  // NaN for a negative pivot (the flagged draw is redone anyway)
    __syncthreads();
    for (int i = j + tid; i < nb; i += 256) L[i][j] = (i == j) ? d : L[i][j] / d;
    __syncthreads();
    // trailing rank-one update, threads as a 16 x 16 grid over (row, column): no integer division
    {
      const int ty = tid >> 4, tx = tid & 15;
      for (int ii = j + 1 + ty; ii < nb; ii += 16) {
        const double lij = L[ii][j];
        for (int kk = j + 1 + tx; kk <= ii; kk += 16) L[ii][kk] -= lij * L[kk][j];
      }
    }
    __syncthreads();
  }
  // inverse of the factor by blocks, 16 -> 32 -> 64: inv([A 0; B C]) = [inv A, 0; -inv(C) B inv(A), inv C].  (One column per thread
  // by forward substitution -- a chain of 2000 dependent LDS reads in thread 0 -- was most of this kernel's 83 us.)
  static_assert(NB == 64, "the block recursion below is written for 64 x 64");
  __shared__ double T2[32][33];
  if (tid < NB && tid >= nb) L[tid][tid] = 1.0;   // identity padding (not written back)
  __syncthreads();
  {  // the four diagonal 16 x 16 blocks: wavefront q, lane c < 16 = column c, the column in registers
    const int q = tid >> 6, c = tid & 63, o = 16 * q;
    if (c < 16) {
      double x[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double sv = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) sv = fma(-L[o + i][o + k], x[k], sv);
        x[i] = (i >= c) ? sv / L[o + i][o + i] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) I[o + i][o + c] = x[i];
    }
  }
  __syncthreads();
  for (int pq = 0; pq < 2; ++pq) {  // the two 32 x 32 blocks: off-diagonal 16 x 16 = -inv(C) (B inv(A)), one element per thread
    const int a0 = 32 * pq, b0 = a0 + 16, r = tid >> 4, c = tid & 15;
    double sv = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) sv = fma(L[b0 + r][a0 + k], I[a0 + k][a0 + c], sv);
    T2[r][c] = sv;
    __syncthreads();
    double uv = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) uv = fma(I[b0 + r][b0 + k], T2[k][c], uv);
    __syncthreads();
    I[b0 + r][a0 + c] = -uv;
  }
  __syncthreads();
  {  // the 64 x 64 block: off-diagonal 32 x 32, four elements per thread
    const int r = tid >> 3, c0 = (tid & 7) * 4;
    double sv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double l = L[32 + r][k];
#pragma unroll
      for (int u = 0; u < 4; ++u) sv[u] = fma(l, I[k][c0 + u], sv[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) T2[r][c0 + u] = sv[u];
    __syncthreads();
    double uv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double ci = I[32 + r][32 + k];
#pragma unroll
      for (int u = 0; u < 4; ++u) uv[u] = fma(ci, T2[k][c0 + u], uv[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) I[32 + r][c0 + u] = -uv[u];
  }
  __syncthreads();
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, j = e - i * NB;
    if (i < nb && j <= i) Ab[(long)i * m + j] = L[i][j];
    const bool in = i < nb && j < nb;
    LINV[b * sL + e] = in ? I[i][j] : 0.0;
    LINVT[b * sL + e] = in ? I[j][i] : 0.0;
  }
}

// a_s[t] = a_f[t] + G (a_s[t+1] - T a_f[t]);  one workgroup per draw, m <= 1024
__global__ __launch_bounds__(256) void sb_avec_kernel(const double* Tm, const double* G, long sG, const double* filt_a,
                                                      double* smooth_a, int n, int m, int t) {
  extern __shared__ double sh[];  // a_f (m), diff (m)
  double* af = sh;
  double* df = sh + m;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const double* gT = Tm + b * (long)m * m;
  const double* fa = filt_a + (b * (long)n + t) * m;
  const double* sn = smooth_a + (b * (long)n + t + 1) * m;
  for (int i = tid; i < m; i += 256) af[i] = fa[i];
  __syncthreads();
  for (int i = tid; i < m; i += 256) {
    double s = 0.0;
    for (int k = 0; k < m; ++k) s = fma(gT[(long)i * m + k], af[k], s);
    df[i] = sn[i] - s;
  }
  __syncthreads();
  const double* Gb = G + b * sG;
  for (int i = tid; i < m; i += 256) {
    double s = af[i];
    for (int k = 0; k < m; ++k) s = fma(Gb[(long)i * m + k], df[k], s);
    smooth_a[(b * (long)n + t) * m + i] = s;
  }
}

// The same in two launches over (row chunks, draws), one wavefront per row (whole lines of T / G; sb_avec_kernel runs one workgroup
// per draw: 155 us at C5's 128 draws):  df = a_s[t+1] - T a_f[t]  ;  a_s[t] = a_f[t] + G df
constexpr int AV_ROWS = 16;
__global__ __launch_bounds__(256) void sb_avec1_kernel(const double* Tm, const double* filt_a, const double* smooth_a, double* df,
                                                       int n, int m, int t) {
  extern __shared__ double sh[];
  const long b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* fa = filt_a + (b * (long)n + t) * m;
  for (int k = threadIdx.x; k < m; k += 256) sh[k] = fa[k];
  __syncthreads();
  const double* gT = Tm + b * (long)m * m;
  const double* sn = smooth_a + (b * (long)n + t + 1) * m;
  for (int rr = wave; rr < AV_ROWS; rr += 4) {
    const int i = blockIdx.x * AV_ROWS + rr;
    if (i >= m) break;
    double s = 0.0;
    for (int k = lane; k < m; k += 64) s = fma(gT[(long)i * m + k], sh[k], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) df[b * m + i] = sn[i] - s;
  }
}
__global__ __launch_bounds__(256) void sb_avec2_kernel(const double* G, long sG, const double* filt_a, const double* df,
                                                       double* smooth_a, int n, int m, int t) {
  extern __shared__ double sh[];
  const long b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = threadIdx.x; k < m; k += 256) sh[k] = df[b * m + k];
  __syncthreads();
  const double* Gb = G + b * sG;
  const double* fa = filt_a + (b * (long)n + t) * m;
  for (int rr = wave; rr < AV_ROWS; rr += 4) {
    const int i = blockIdx.x * AV_ROWS + rr;
    if (i >= m) break;
    double s = 0.0;
    for (int k = lane; k < m; k += 64) s = fma(Gb[(long)i * m + k], sh[k], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) smooth_a[(b * (long)n + t) * m + i] = fa[i] + s;
  }
}

// scale[b] = max diag(P_hat[b]) (P_hat is PSD: its largest entry is on the diagonal)
__global__ __launch_bounds__(256) void sb_scale_kernel(const double* PH, long sP, int m, double* scale) {
  __shared__ double red[256];
  const long b = blockIdx.x;
  double v = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) v = fmax(v, fabs(PH[b * sP + (long)i * m + i]));
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[b] = red[0];
}

constexpr int PNT = 1024;  // threads of the pseudo-inverse workgroup

__device__ __forceinline__ double pinv_block_sum(double v, double* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = PNT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  return red[0];
}

// Moore-Penrose inverse of the symmetric m x m matrix D (global memory, destroyed) -> Pi; V, ev (m), cs (4 ceil(m/2))
// are scratch.  Round-robin (parallel-ordered) cyclic Jacobi as pinv_jacobi of kf_smoother_lds.hip, one workgroup.
__device__ void pinv_jacobi_wg(double* D, double* V, double* Pi, double* ev, double* cs, int m, double* red) {
  const int tid = threadIdx.x;
  const long mm = (long)m * m;
  const int M = (m + 1) & ~1, half = M / 2;
  for (long e = tid; e < mm; e += PNT) V[e] = (e / m == e % m) ? 1.0 : 0.0;
  __syncthreads();
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (long e = tid; e < mm; e += PNT) {
      const double x = D[e];
      if (e / m == e % m) dg += x * x; else off += x * x;
    }
    off = pinv_block_sum(off, red);
    dg = pinv_block_sum(dg, red);
    if (!(off > 1e-30 * dg && off > 0.0)) break;  // block-uniform
    for (int rd = 0; rd < M - 1; ++rd) {
      for (int pr = tid; pr < half; pr += PNT) {
        int p_, q_;
        if (pr == 0) { p_ = M - 1; q_ = rd; }
        else { p_ = (rd + pr) % (M - 1); q_ = (rd - pr + (M - 1)) % (M - 1); }
        if (p_ > q_) { const int t_ = p_; p_ = q_; q_ = t_; }
        double c = 1.0, s_ = 0.0;
        if (q_ < m) {
          const double apq = D[(long)p_ * m + q_];
          if (apq != 0.0) {
            const double app = D[(long)p_ * m + p_], aqq = D[(long)q_ * m + q_];
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            s_ = t * c;
          }
        } else {
          p_ = q_ = -1;
        }
        cs[4 * pr + 0] = c;
        cs[4 * pr + 1] = s_;
        cs[4 * pr + 2] = (double)p_;
        cs[4 * pr + 3] = (double)q_;
      }
      __threadfence_block();
      __syncthreads();
      for (long idx = tid; idx < (long)half * m; idx += PNT) {  // columns of D and V
        const int pr = (int)(idx / m), k = (int)(idx - (long)pr * m);
        const int p_ = (int)cs[4 * pr + 2], q_ = (int)cs[4 * pr + 3];
        if (p_ < 0) continue;
        const double c = cs[4 * pr + 0], s_ = cs[4 * pr + 1];
        const double dkp = D[(long)k * m + p_], dkq = D[(long)k * m + q_];
        D[(long)k * m + p_] = c * dkp - s_ * dkq;
        D[(long)k * m + q_] = s_ * dkp + c * dkq;
        const double vkp = V[(long)k * m + p_], vkq = V[(long)k * m + q_];
        V[(long)k * m + p_] = c * vkp - s_ * vkq;
        V[(long)k * m + q_] = s_ * vkp + c * vkq;
      }
      __threadfence_block();
      __syncthreads();
      for (long idx = tid; idx < (long)half * m; idx += PNT) {  // rows of D
        const int pr = (int)(idx / m), k = (int)(idx - (long)pr * m);
        const int p_ = (int)cs[4 * pr + 2], q_ = (int)cs[4 * pr + 3];
        if (p_ < 0) continue;
        const double c = cs[4 * pr + 0], s_ = cs[4 * pr + 1];
        const double dpk = D[(long)p_ * m + k], dqk = D[(long)q_ * m + k];
        D[(long)p_ * m + k] = c * dpk - s_ * dqk;
        D[(long)q_ * m + k] = s_ * dpk + c * dqk;
      }
      __threadfence_block();
      __syncthreads();
    }
  }
  double smax = 0.0;
  for (int k = 0; k < m; ++k) smax = fmax(smax, fabs(D[(long)k * m + k]));
  for (int k = tid; k < m; k += PNT) {
    const double lam = D[(long)k * m + k];
    ev[k] = (fabs(lam) > 1e-15 * smax) ? 1.0 / lam : 0.0;
  }
  __threadfence_block();
  __syncthreads();
  for (long e = tid; e < mm; e += PNT) {
    const int i = (int)(e / m), j = (int)(e - (long)i * m);
    double s_ = 0.0;
    for (int k = 0; k < m; ++k) s_ += V[(long)i * m + k] * V[(long)j * m + k] * ev[k];
    Pi[e] = s_;
  }
  __threadfence_block();
  __syncthreads();
}

// The whole series again for the draws the Cholesky pass flagged, with pinv at every step (the reference's
// smoother_step, kalman_smoother.py:87-97).  Scratch: the draw's own slices of the workspace arrays.
__global__ __launch_bounds__(PNT) void sb_pinv_redo_kernel(const KfSmoothArgs A, const int32_t* flags, const double* RQRw,
                                                           double* Aw, double* Vw, double* Pw, double* Sw) {
  __shared__ double red[PNT];
  const long b = blockIdx.x;
  if (!flags[b]) return;
  const int tid = threadIdx.x;
  const int n = A.T, m = A.m;
  const long mm = (long)m * m;
  double* Am = Aw + b * mm;   // P_hat, then W = T P_f, then E
  double* V = Vw + b * mm;    // Jacobi vectors, then X = pinv(P_hat) W
  double* Pi = Pw + b * mm;   // pinv, then D = P_s[t+1] - P_hat
  const double* RQR = RQRw + b * mm;
  double* ev = Sw + b * mm;   // m
  double* cs = ev + m;        // 4 ceil(m / 2)
  double* va = cs + 4 * ((m + 1) / 2);  // a_f (m), diff (m)     (m^2 >= 5 m + 4 for m >= 7)
  const double* gT = A.Tm + b * mm;
  const double* gfa = A.filt_a + b * (long)n * m;
  const double* gfP = A.filt_P + b * (long)n * mm;
  double* gsa = A.smooth_a + b * (long)n * m;
  double* gsP = A.smooth_P + b * (long)n * mm;
  for (long e = tid; e < mm; e += PNT) gsP[(long)(n - 1) * mm + e] = gfP[(long)(n - 1) * mm + e];
  for (int e = tid; e < m; e += PNT) gsa[(long)(n - 1) * m + e] = gfa[(long)(n - 1) * m + e];
  auto sync = [&]() {
    __threadfence_block();
    __syncthreads();
  };
  sync();
  for (int t = n - 2; t >= 0; --t) {
    const double* Pf = gfP + (long)t * mm;
    const double* Psn = gsP + (long)(t + 1) * mm;
    for (long e = tid; e < mm; e += PNT) {  // W = T P_f (in V for now)
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(gT[(long)i * m + k], Pf[(long)k * m + j], v);
      V[e] = v;
    }
    sync();
    for (long e = tid; e < mm; e += PNT) {  // P_hat = W T' + RQR'
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double v = RQR[e];
      for (int k = 0; k < m; ++k) v = fma(V[(long)i * m + k], gT[(long)j * m + k], v);
      Am[e] = v;
    }
    sync();
    for (long e = tid; e < mm; e += PNT) {  // the Jacobi iteration wants an exactly symmetric matrix
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      if (i < j) {
        const double v = 0.5 * (Am[(long)i * m + j] + Am[(long)j * m + i]);
        Am[(long)i * m + j] = v;
        Am[(long)j * m + i] = v;
      }
    }
    sync();
    pinv_jacobi_wg(Am, V, Pi, ev, cs, m, red);  // Pi = pinv(P_hat); Am, V destroyed
    for (long e = tid; e < mm; e += PNT) {      // W again (Am)
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(gT[(long)i * m + k], Pf[(long)k * m + j], v);
      Am[e] = v;
    }
    sync();
    for (long e = tid; e < mm; e += PNT) {      // X = Pi W (V)
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(Pi[(long)i * m + k], Am[(long)k * m + j], v);
      V[e] = v;
    }
    sync();
    for (long e = tid; e < mm; e += PNT) {      // D = P_s[t+1] - P_hat (Pi), P_hat recomputed from W
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double v = RQR[e];
      for (int k = 0; k < m; ++k) v = fma(Am[(long)i * m + k], gT[(long)j * m + k], v);
      Pi[e] = Psn[e] - v;
    }
    for (int i = tid; i < m; i += PNT) va[i] = gfa[(long)t * m + i];
    sync();
    for (int i = tid; i < m; i += PNT) {        // diff = a_s[t+1] - T a_f
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(gT[(long)i * m + k], va[k], v);
      va[m + i] = gsa[(long)(t + 1) * m + i] - v;
    }
    for (long e = tid; e < mm; e += PNT) {      // E = D X (Am)
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double v = 0.0;
      for (int k = 0; k < m; ++k) v = fma(Pi[(long)i * m + k], V[(long)k * m + j], v);
      Am[e] = v;
    }
    sync();
    for (long e = tid; e < mm; e += PNT) {      // P_s[t] = P_f + X' E
      const int i = (int)(e / m), j = (int)(e - (long)i * m);
      double v = Pf[e];
      for (int k = 0; k < m; ++k) v = fma(V[(long)k * m + i], Am[(long)k * m + j], v);
      gsP[(long)t * mm + e] = v;
    }
    for (int i = tid; i < m; i += PNT) {        // a_s[t] = a_f + X' diff
      double v = va[i];
      for (int k = 0; k < m; ++k) v = fma(V[(long)k * m + i], va[m + k], v);
      gsa[(long)t * m + i] = v;
    }
    sync();
  }
}

// smooth[n-1] = filt[n-1]
__global__ __launch_bounds__(256) void sb_last_kernel(const double* filt_a, const double* filt_P, double* smooth_a,
                                                      double* smooth_P, int n, int m) {
  const long b = blockIdx.y;
  const long mm = (long)m * m;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < mm) smooth_P[(b * n + (n - 1)) * mm + e] = filt_P[(b * n + (n - 1)) * mm + e];
  if (e < m) smooth_a[(b * n + (n - 1)) * m + e] = filt_a[(b * n + (n - 1)) * m + e];
}

template <bool TRANS_B, int EPI>
void gemm(hipStream_t st, long B, const double* A, long sA, int lda, const double* Bm, long sB, int ldb, const double* E,
          long sE, int lde, double* C, long sC, int ldc, int M, int N, int K) {
  if (M <= 0 || N <= 0) return;
  dim3 grid((unsigned)((N + GTN - 1) / GTN), (unsigned)((M + GTM - 1) / GTM), (unsigned)B);
  hipLaunchKernelGGL((sb_gemm_kernel<TRANS_B, EPI>), grid, dim3(GNT), 0, st, A, sA, lda, Bm, sB, ldb, E, sE, lde, C, sC,
                     ldc, M, N, K);
}

}  // namespace

void kf_sb_gemm(hipStream_t st, long B, bool trans_b, int epi, const double* A, long sA, int lda, const double* Bm,
                long sB, int ldb, const double* E, long sE, int lde, double* C, long sC, int ldc, int M, int N, int K) {
#define KF_SB_CALL(TB_, EPI_) gemm<TB_, EPI_>(st, B, A, sA, lda, Bm, sB, ldb, E, sE, lde, C, sC, ldc, M, N, K)
  if (trans_b) {
    if (epi == EPI_STORE) KF_SB_CALL(true, EPI_STORE);
    else if (epi == EPI_ADD) KF_SB_CALL(true, EPI_ADD);
    else KF_SB_CALL(true, EPI_RSUB);
  } else {
    if (epi == EPI_STORE) KF_SB_CALL(false, EPI_STORE);
    else if (epi == EPI_ADD) KF_SB_CALL(false, EPI_ADD);
    else KF_SB_CALL(false, EPI_RSUB);
  }
#undef KF_SB_CALL
}

void kf_sb_transpose(hipStream_t st, long B, const double* src, long sS, double* dst, long sD, int m) {
  const unsigned tb = (unsigned)((m + 31) / 32);
  hipLaunchKernelGGL(sb_transpose_kernel, dim3(tb, tb, (unsigned)B), dim3(256), 0, st, src, sS, dst, sD, m);
}

size_t kf_smoother_big_workspace_bytes(long B, int m) {
  const long mm = (long)m * m;
  const int nblk = (m + NB - 1) / NB;
  return sizeof(double) * (size_t)B * (size_t)(6 * mm + 2L * nblk * NB * NB + 2) + sizeof(uint32_t) * (size_t)B * ((m + 63) / 64) +
         320;  // + scale, flags per draw; the block map of T
}

int kf_launch_smoother_big(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                           hipStream_t st) {
  const long B = a.B;
  const int m = a.m, n = a.T, r = a.r;
  const long mm = (long)m * m;
  const size_t need = kf_smoother_big_workspace_bytes(B, m);
  if (B > 65535) {  // the draw axis sits in gridDim.y/z of every per-step launch
    KF_SET_ERR(h, "smoother(big): B=%ld > 65535 draws per call (chunk the batch)", B);
    return KF_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < need) {
    KF_SET_ERR(h, "smoother(big): m=%d needs a workspace of %zu bytes (kf_workspace_bytes), got %zu", m, need,
               workspace_bytes);
    return KF_ERR_WORKSPACE;
  }
  if (m > 1024) {
    KF_SET_ERR(h, "smoother(big): m=%d > 1024 not supported", m);
    return KF_ERR_UNSUPPORTED;
  }
  const int nblk = (m + NB - 1) / NB;
  const long sL = (long)nblk * NB * NB;
  double* ws = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  double* RQR = ws;
  double* W = RQR + B * mm;    // T P_f, then Y, then X
  double* PH = W + B * mm;     // P_hat, then its Cholesky factor L (lower)
  double* Dm = PH + B * mm;    // P_s[t+1] - P_hat
  double* U = Dm + B * mm;     // L', then G = X'
  double* Em = U + B * mm;     // D X
  double* LINV = Em + B * mm;
  double* LINVT = LINV + B * sL;
  double* scale = LINVT + B * sL;                       // max diag(P_hat) of the current step
  int32_t* flags = reinterpret_cast<int32_t*>(scale + B);  // draw needs the pseudo-inverse pass
  KF_HIP_CHECK(h, hipMemsetAsync(flags, 0, sizeof(int32_t) * (size_t)B, st));

  const unsigned eb = (unsigned)((mm + 255) / 256);
  hipLaunchKernelGGL(sb_rqr_kernel, dim3(eb, (unsigned)B), dim3(256), 0, st, a.R, a.Q, RQR, m, r);
  hipLaunchKernelGGL(sb_last_kernel, dim3(eb, (unsigned)B), dim3(256), 0, st, a.filt_a, a.filt_P, a.smooth_a,
                     a.smooth_P, n, m);
  const long sTape = (long)n * mm;
  const unsigned tb = (unsigned)((m + 31) / 32);
  // KF_SMOOTHER_BIG_GEMM=0: the four full-size products on this file's own kernel (A/B measurements); m < 128: its smaller tiles fit better
  const char* bge = kf_env_switch("KF_SMOOTHER_BIG_GEMM");
  const bool big_full = m >= 128 && !(bge && bge[0] == '0');
  uint32_t* tmask = nullptr;   // block map of T (KF_BIG_TMASK=0: every chunk)
  {
    const char* te = kf_env_switch("KF_BIG_TMASK");
    if (big_full && !(te && te[0] == '0')) {
      tmask = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(flags + B) + 63) & ~(uintptr_t)63);
      if (kf_big_tmask_launch(st, a.Tm, m, B, tmask) != KF_OK) tmask = nullptr;
    }
  }
  for (int t = n - 2; t >= 0; --t) {
    const double* Pf = a.filt_P + (long)t * mm;
    const double* Psn = a.smooth_P + (long)(t + 1) * mm;
    double* Ps = a.smooth_P + (long)t * mm;
    // W = T P_f ; P_hat = W T' + RQR' ; D = P_s[t+1] - P_hat
    // (the four full-size products on kf_forward_big.hip's 128 x 64-tile kernel, the symmetric ones on the tiles that touch the upper
    //  triangle with both triangles written: 334 / 346 / 335 / 419 us on this file's 64 x 64-tile kernel -> 270 / 255 / 270 / 255 at C5)
    if (big_full) {
      // (both products by T on the K chunks whose 64 x 32 block of T holds a non-zero: kf_forward_big.hip's map, bit for bit the same)
      kf_big_gemm_launch_x(st, false, false, a.Tm, mm, Pf, sTape, nullptr, 0, W, mm, m, B, tmask, /*on_b=*/0, 0, nullptr, nullptr, nullptr,
                           nullptr, false);
      kf_big_gemm_launch_x(st, true, true, W, mm, a.Tm, mm, RQR, mm, PH, mm, m, B, tmask, /*on_b=*/1, 0, nullptr, nullptr, nullptr, nullptr,
                           /*upper_mirror=*/true);
    } else {
      gemm<false, EPI_STORE>(st, B, a.Tm, mm, m, Pf, sTape, m, nullptr, 0, 0, W, mm, m, m, m, m);
      gemm<true, EPI_ADD>(st, B, W, mm, m, a.Tm, mm, m, RQR, mm, m, PH, mm, m, m, m, m);
    }
    hipLaunchKernelGGL(sb_sub_kernel, dim3(eb, (unsigned)B), dim3(256), 0, st, Psn, sTape, PH, mm, Dm, mm, mm);
    hipLaunchKernelGGL(sb_scale_kernel, dim3((unsigned)B), dim3(256), 0, st, PH, mm, m, scale);
    // blocked Cholesky of P_hat (lower, in place)
    for (int k = 0; k < nblk; ++k) {
      const int k1 = k * NB, nb = (m - k1 < NB) ? m - k1 : NB, k2 = k1 + nb, rem = m - k2;
      hipLaunchKernelGGL(sb_potrf_inv_kernel, dim3((unsigned)B), dim3(256), 0, st, PH, mm, m, k1, nb,
                         LINV + (long)k * NB * NB, LINVT + (long)k * NB * NB, sL, scale, flags);
      if (rem > 0) {
        double* panel = PH + (long)k2 * m + k1;
        // L21 = A21 inv(L11)'   (in place: every workgroup reads exactly the rows it writes)
        gemm<true, EPI_STORE>(st, B, panel, mm, m, LINV + (long)k * NB * NB, sL, NB, nullptr, 0, 0, panel, mm, m, rem,
                              nb, nb);
        // A22 -= L21 L21'
        double* A22 = PH + (long)k2 * m + k2;
        gemm<true, EPI_RSUB>(st, B, panel, mm, m, panel, mm, m, A22, mm, m, A22, mm, m, rem, rem, nb);
      }
    }
    hipLaunchKernelGGL(sb_transpose_kernel, dim3(tb, tb, (unsigned)B), dim3(256), 0, st, PH, mm, U, mm, m);
    // forward substitution  L Y = W  (Y overwrites W)
    for (int k = 0; k < nblk; ++k) {
      const int k1 = k * NB, nb = (m - k1 < NB) ? m - k1 : NB, k2 = k1 + nb, rem = m - k2;
      double* Wk = W + (long)k1 * m;
      gemm<false, EPI_STORE>(st, B, LINV + (long)k * NB * NB, sL, NB, Wk, mm, m, nullptr, 0, 0, Wk, mm, m, nb, m, nb);
      if (rem > 0) {
        double* Wr = W + (long)k2 * m;
        gemm<false, EPI_RSUB>(st, B, PH + (long)k2 * m + k1, mm, m, Wk, mm, m, Wr, mm, m, Wr, mm, m, rem, m, nb);
      }
    }
    // backward substitution  L' X = Y  (X overwrites W), U = L'
    for (int k = nblk - 1; k >= 0; --k) {
      const int k1 = k * NB, nb = (m - k1 < NB) ? m - k1 : NB;
      double* Wk = W + (long)k1 * m;
      gemm<false, EPI_STORE>(st, B, LINVT + (long)k * NB * NB, sL, NB, Wk, mm, m, nullptr, 0, 0, Wk, mm, m, nb, m, nb);
      if (k1 > 0) gemm<false, EPI_RSUB>(st, B, U + k1, mm, m, Wk, mm, m, W, mm, m, W, mm, m, k1, m, nb);
    }
    // E = D X ; G = X' ; P_s[t] = G E + P_f[t] ; a_s[t]
    if (big_full) kf_big_gemm_launch(st, false, false, false, Dm, mm, W, mm, nullptr, 0, Em, mm, m, B);
    else gemm<false, EPI_STORE>(st, B, Dm, mm, m, W, mm, m, nullptr, 0, 0, Em, mm, m, m, m, m);
    hipLaunchKernelGGL(sb_transpose_kernel, dim3(tb, tb, (unsigned)B), dim3(256), 0, st, W, mm, U, mm, m);
    if (big_full) kf_big_gemm_launch(st, false, true, true, U, mm, Em, mm, Pf, sTape, Ps, sTape, m, B);
    else gemm<false, EPI_ADD>(st, B, U, mm, m, Em, mm, m, Pf, sTape, m, Ps, sTape, m, m, m, m);
    // (E is free again: its first B m doubles carry df between the two launches)
    hipLaunchKernelGGL(sb_avec1_kernel, dim3((unsigned)((m + AV_ROWS - 1) / AV_ROWS), (unsigned)B), dim3(256), sizeof(double) * m, st,
                       a.Tm, a.filt_a, a.smooth_a, Em, n, m, t);
    hipLaunchKernelGGL(sb_avec2_kernel, dim3((unsigned)((m + AV_ROWS - 1) / AV_ROWS), (unsigned)B), dim3(256), sizeof(double) * m, st,
                       U, mm, a.filt_a, Em, a.smooth_a, n, m, t);
  }
  // draws whose P_hat was not safely positive definite at some step: the whole series again with pinv
  hipLaunchKernelGGL(sb_pinv_redo_kernel, dim3((unsigned)B), dim3(PNT), 0, st, a, flags, RQR, W, PH, Dm, U);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
