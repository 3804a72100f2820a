// kf_forward_big.hip -- forward Kalman filter for LARGE state dimension (per-draw tiles do not
// fit LDS: m > ~40, e.g. BASELINE config C5, m = 367).
//
// Same contract as the small-m kernels (reference pymc_statespace/filters/kalman_filter.py:
// 126-252, 255-351, 444-505).  Here the covariance lives in HBM/L2 (caller-provided workspace,
// 4 B m^2 doubles) and every time step is a short sequence of launches on the caller's stream:
//
//   big_update   one workgroup per draw: mask, v, M = P Z', F, chol(F), K, ll, a_f,
//                P_f = P - K M' - M K' + (K F) K' (rank-p, expanded Joseph form :277),
//                a_hat = T a_f + c; filtered tapes
//   big_gemm     W = T P_f            batched over draws, fp64 MFMA (v_mfma_f64_16x16x4_f64)
//   big_gemm     S = W T' + R Q R'    same kernel, B operand transposed, RQR' in the epilogue
//   big_sym      P_hat = (S + S')/2 (:221) through 32x32 LDS transposes; predicted tapes
//
// The two m x m x m products are >99% of the flops (m = 367: 2 x 98.9 MFLOP per draw-step) and
// are real contractions, so they run on the matrix cores: 128x64 output tile per 512-thread
// workgroup (8 waves, 2x2 MFMA 16x16 tiles each), K staged through LDS in chunks of 32 with the
// next chunk prefetched into registers.
#include <stdlib.h>

#include <type_traits>

#include "kf_internal.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GTM = 128, GTN = 64;   // output tile; measured at m=367 (C5 share, ms per step): 64x64 / 4 waves 0.888,
                                     // 128x64 / 8 waves 0.825, 128x128 / 16 waves (K chunk 16) 0.835
constexpr int GK = 32;               // K chunk
constexpr int GNT = 512;             // threads: 8 waves as 4 x 2, 32 x 32 per wave
constexpr int GWR = 4, GWC = 2;
constexpr int GTI = GTM / GWR / 16;  // MFMA tiles per wave, rows (2)
constexpr int GTJ = GTN / GWC / 16;  // MFMA tiles per wave, cols (4)
constexpr int GLDA = GTM + 1, GLDB = GTN + 1;  // odd LDS row strides
constexpr int GPA = GTM * GK / GNT, GPB = GTN * GK / GNT;  // staged elements per thread

// C[b] (m x m) = A[b] (m x m) * op(B[b])  (+ E[b]);  op = identity or transpose.
// grid: 1-D, 8 * tiles * ceil(draws / 8) workgroups (XCD-aware, see below)  block: 512.  Software pipeline: the next K chunk is fetched
// from HBM/L2 into registers while the MFMAs of the current chunk run out of LDS.
// UPPER: the result is symmetric up to rounding and only its upper triangle is consumed (big_sym_kernel
// mirrors it): tiles entirely below the diagonal are skipped -- a third of them at m = 367
// Round 3, two small per-step kernels folded into the products (they were 13 % of a C5 step, each a pass over a B m^2 array):
// PFIX (p == 1, !TRANS_B, m <= 1024): the B operand is P and becomes P_f = P - k m' - m k' + (k f) k' ON ITS WAY INTO LDS (the three
//   vectors x.kv, x.mv, x.kfv sit in LDS; a thread stages one column and wave-uniform rows, so the fix is three LDS broadcasts and three
//   FMAs per staged element, in big_pf_kernel's order of operations): W = T P_f without big_pf_kernel's pass over P (valid when the
//   filtered covariance tape is not requested).  Measured at C5 (128 x 730): the fix costs 15 us per launch, the a_hat workgroups 17
//   (whether 32 or 64 rows each), against 59 + 44 us for the two kernels they replace.  The same launch carries 8 * ceil(B / 8) * ceil(m / 64) more workgroups AHEAD of the tiles
//   (behind them they were a 40 us tail) that compute a_hat = T a_f + c (big_apred_kernel's arithmetic) while the tiles keep the MFMA pipes busy;
// MIRROR (with UPPER): an element (r, c) with r <= c is stored at (r, c) AND (c, r) of C (and of x.ctape), elements below the
//   diagonal are dropped -- big_sym_kernel's mirror of the upper triangle happens in the epilogue of S = W T' + R Q R'.
//   With x.mrow: the tile also leaves its share of the NEXT step's m = P z (p == 1) -- per row the sum over its columns, per column
//   the sum over its rows (the mirrored elements) -- in x.mrow / x.mcol; big_gain_kernel adds the shares of the tiles that exist, in a
//   fixed order (no atomics: the result does not depend on the order the workgroups ran in), and big_mvec_kernel's pass over P is gone.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct BigGemmExtra {
  const double* kv;     // [B][m]  k
  const double* mv;     // [B][m]  m = P z
  const double* kfv;    // [B][m]  k f
  const double* af;     // [B][m]  filtered state
  const double* cvec;   // [B][m] or NULL
  double* aout;         // [B][m]  a_hat
  double* atape;        // [B] x atape_stride, or NULL
  long atape_stride;
  double* ctape;        // [B] x ctape_stride (m x m images), or NULL
  long ctape_stride;
  const double* zv;     // [B][m]  Z (p == 1), with mrow / mcol
  double* mrow;         // [B][tiles_n][m]  partial sums of C z over the columns of one tile column, or NULL
  double* mcol;         // [B][tiles_m][m]  the mirrored elements' share, over the rows of one tile row
  // Round 5: block sparsity of the STATIC factor.  tmask[b][rb] (rb = 64-row block of T) has bit kc set when T[64 rb .. +64][32 kc .. +32]
  // holds a non-zero (big_tmask_kernel, once per call, m <= 1024); a K chunk whose T block is all zero contributes exact zeros to the
  // accumulators and is skipped -- no loads, no staging, no MFMAs, bit for bit the same product.  tmask_on_b: T is the (transposed)
  // B operand -- the tile's 64 columns are rows c0 .. of T -- else the A operand (the tile's 128 rows = two 64-row blocks).
  // The reference's model classes build T from identity / companion / seasonal blocks (models/SARIMAX.py:59-98, VARMAX.py:95-143,
  // the daily-seasonality notebook): at C5 (m = 367) 61 % of W = T P_f's chunks and 38 % of S = W T' + R Q R''s survive.
  const uint32_t* tmask = nullptr;
  int tmask_rb = 0;     // 64-row blocks per draw
  int tmask_on_b = 0;
  // the same map of E (R Q R': non-zero only where R has rows -- a 3 x 3 corner of the 367 x 367 matrix at C5, yet 138 MB per launch
  // as a dense read): a tile whose block of E is all zero does not load it
  const uint32_t* emask = nullptr;
  // (kf_smoother_bf_big.hip) C = [-] A op(B) [+ E] + u1 v1' + u2 v2': the sign flips the product BEFORE E is added (E - A op(B)); the
  // rank-one terms are per-draw vectors [B][m] (any of them NULL: that term is absent)
  int neg_acc = 0;
  const double *u1 = nullptr, *v1 = nullptr, *u2 = nullptr, *v2 = nullptr;
};

// tmask[b][rb]: which 32-column chunks of the 64-row block rb of T[b] hold a non-zero (m <= 1024: at most 32 chunks)
__global__ __launch_bounds__(256) void big_tmask_kernel(const double* __restrict__ T, int m, uint32_t* __restrict__ tmask) {
  __shared__ unsigned bits;
  const int rb = blockIdx.x, nrb = gridDim.x;
  const long b = blockIdx.y;
  if (threadIdx.x == 0) bits = 0u;
  __syncthreads();
  const double* Tb = T + b * (long)m * m;
  const int r0 = rb * 64, r1 = min(r0 + 64, m);
  unsigned mine = 0u;
  for (long e = (long)r0 * m + threadIdx.x; e < (long)r1 * m; e += 256) {
    const double v = Tb[e];
    if (v != 0.0 || v != v) mine |= 1u << (unsigned)((e % m) / 32);   // (a NaN counts: it must reach the product)
  }
  if (mine) atomicOr(&bits, mine);
  __syncthreads();
  if (threadIdx.x == 0) tmask[b * nrb + rb] = bits;
}
// the same map by COLUMNS: word [b][cb] bit kc set when T[32 kc .. +32][64 cb .. +64] holds a non-zero -- what a product A T (T the
// untransposed B operand: K runs over its rows, a tile's 64 columns are columns of T) skips by.  grid (ceil(m / 64), B), m <= 1024
__global__ __launch_bounds__(256) void big_tmask_cols_kernel(const double* __restrict__ T, int m, uint32_t* __restrict__ tmask) {
  __shared__ unsigned bits;
  const long b = blockIdx.y;
  const int cb = blockIdx.x, ncb = gridDim.x;
  if (threadIdx.x == 0) bits = 0u;
  __syncthreads();
  const double* Tb = T + b * (long)m * m;
  const int c0 = cb * 64, lane = threadIdx.x & 63;
  unsigned mine = 0u;
  if (c0 + lane < m)
    for (int k = threadIdx.x >> 6; k < m; k += 4) {   // rows 4 apart per wavefront, 64 consecutive columns per row: coalesced
      const double v = Tb[(long)k * m + c0 + lane];
      if (v != 0.0 || v != v) mine |= 1u << (unsigned)(k / 32);
    }
  if (mine) atomicOr(&bits, mine);
  __syncthreads();
  if (threadIdx.x == 0) tmask[b * ncb + cb] = bits;
}
constexpr int BIG_PFIX_M = 1024;
constexpr int BIG_AH_ROWS = 64;   // rows of a_hat = T a_f + c per workgroup ahead of the tiles
template <bool TRANS_B, bool ADD_E, bool UPPER = false, bool PFIX = false, bool MIRROR = false>
__global__ __launch_bounds__(GNT) __attribute__((amdgpu_waves_per_eu(4))) void big_gemm_kernel(const double* __restrict__ A, long strideA,
                                                       const double* __restrict__ Bm, long strideB,
                                                       const double* __restrict__ E, long strideE,
                                                       double* __restrict__ Cm, long strideC, int m,
                                                       long B, int tiles_n, int tiles_m, const BigGemmExtra x) {
  // one block: As[k][r] (A tile, k-major) | Bs[k][c]; the MIRROR epilogue transposes half a tile through its first 64 x 65 doubles
  __shared__ double lds_raw[GK * GLDA + GK * GLDB];
  static_assert(GK * GLDA + GK * GLDB >= 64 * 65, "the mirror epilogue's transpose buffer");
  double (*As)[GLDA] = reinterpret_cast<double (*)[GLDA]>(lds_raw);
  double (*Bs)[GLDB] = reinterpret_cast<double (*)[GLDB]>(lds_raw + GK * GLDA);
  __shared__ double vsK[PFIX ? BIG_PFIX_M + GK : 1], vsM[PFIX ? BIG_PFIX_M + GK : 1], vsKF[PFIX ? BIG_PFIX_M + GK : 1];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / GWC, wc = wave % GWC;
  if (PFIX) {  // the workgroups ahead of the tiles: a_hat = T a_f + c, four rows per wavefront at a time
    const int nch = (m + BIG_AH_ROWS - 1) / BIG_AH_ROWS;
    const long xblocks = 8L * nch * ((B + 7) / 8);
    if ((long)blockIdx.x < xblocks) {
      const long Lx = blockIdx.x;
      const long bx = (Lx & 7) + 8 * (Lx / (8L * nch));
      if (bx >= B) return;
      const int ch = (int)((Lx >> 3) % nch);
      double* vs = &As[0][0];
      for (int k = tid; k < m; k += GNT) vs[k] = x.af[bx * m + k];
      __syncthreads();
      const double* Tb = A + bx * strideA;
      constexpr int RW = BIG_AH_ROWS / 8;  // rows per wavefront, all in flight together
      const int i0 = ch * BIG_AH_ROWS + wave * RW;
      double sacc[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) sacc[r] = 0.0;
      // (round 5: 64-column pieces of T whose two 32-column chunks are all zero in this 64-row block are not read -- the same sums)
      const unsigned tbits = x.tmask ? x.tmask[bx * x.tmask_rb + ch] : 0xffffffffu;
      for (int k = lane; k < m; k += 64) {
        if (!((tbits >> ((k - lane) / 32)) & 3u)) continue;
        const double av = vs[k];
#pragma unroll
        for (int r = 0; r < RW; ++r) sacc[r] = fma(Tb[(long)min(i0 + r, m - 1) * m + k], av, sacc[r]);
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const double sv = wave_sum(sacc[r]);
        const int i = i0 + r;
        if (lane == 0 && i < m) {
          const double v = sv + (x.cvec ? x.cvec[bx * m + i] : 0.0);
          x.aout[bx * m + i] = v;
          if (x.atape) x.atape[bx * x.atape_stride + i] = v;
        }
      }
      return;
    }
  }
  // XCD-aware mapping: workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD
  // has its own L2.  All tiles of one draw share the draw's A and B rows, so they must share an
  // XCD: linear id L -> draw = (L % 8) + 8 * (L / (8 * tiles)), tile = (L / 8) % tiles.
  const int tiles = tiles_n * tiles_m;
  const long L = PFIX ? blockIdx.x - 8L * ((m + BIG_AH_ROWS - 1) / BIG_AH_ROWS) * ((B + 7) / 8) : blockIdx.x;  // (a multiple of 8: the XCD of a draw is unchanged)
  const long b = (L & 7) + 8 * (L / (8L * tiles));
  if (b >= B) return;
  const int tile = (int)((L >> 3) % tiles);
  const int r0 = (tile / tiles_n) * GTM, c0 = (tile % tiles_n) * GTN;
  if (UPPER && r0 > c0 + GTN - 1) return;
  const double* Ab = A + b * strideA;
  const double* Bb = Bm + b * strideB;
  double pf_kc = 0.0, pf_mc = 0.0;  // this thread's column of the B tile (tid % GTN) is the same in every chunk
  if (PFIX) {
    const int mp = (m + GK - 1) / GK * GK;
    for (int k = tid; k < mp; k += GNT) {
      vsK[k] = k < m ? x.kv[b * m + k] : 0.0;
      vsM[k] = k < m ? x.mv[b * m + k] : 0.0;
      vsKF[k] = k < m ? x.kfv[b * m + k] : 0.0;
    }
    const int gc = c0 + (tid % GTN);
    if (gc < m) {
      pf_kc = x.kv[b * m + gc];
      pf_mc = x.mv[b * m + gc];
    }
    // (the first __syncthreads() of the loop below orders these stores before the first stage() that reads them?  No: stage() comes
    // first -- hence this one)
    __syncthreads();
  }

  double4_t acc[GTI][GTJ];
#pragma unroll
  for (int i = 0; i < GTI; ++i)
#pragma unroll
    for (int j = 0; j < GTJ; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int li = lane & 15, lk = lane >> 4;
  double ra[GPA], rb[GPB];

  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < GPA; ++q) {  // A tile: rows r0.., cols k0.. (GK contiguous doubles per row)
      const int e = tid + GNT * q;
      const int rr = e / GK, kk = e - rr * GK;
      const int gr = r0 + rr, gk = k0 + kk;
      ra[q] = (gr < m && gk < m) ? Ab[(long)gr * m + gk] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < GPB; ++q) {
      const int e = tid + GNT * q;
      if (TRANS_B) {  // op(B)[k][c] = B[c][k]
        const int cc = e / GK, kk = e - cc * GK;
        const int gc = c0 + cc, gk = k0 + kk;
        rb[q] = (gc < m && gk < m) ? Bb[(long)gc * m + gk] : 0.0;
      } else {
        const int kk = e / GTN, cc = e - kk * GTN;
        const int gc = c0 + cc, gk = k0 + kk;
        rb[q] = (gc < m && gk < m) ? Bb[(long)gk * m + gc] : 0.0;
      }
    }
  };
  auto stage = [&](int k0) {
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
        if (PFIX) {
          double sv = rb[q];
          sv -= vsK[k0 + kk] * pf_mc;
          sv -= vsM[k0 + kk] * pf_kc;
          sv += vsKF[k0 + kk] * pf_kc;
          rb[q] = sv;
        }
        Bs[kk][cc] = rb[q];
      }
    }
  };

  // K chunks: all of them, or (x.tmask) only those whose block of the static factor T holds a non-zero
  const int nch = (m + GK - 1) / GK;
  const bool use_mask = x.tmask != nullptr && nch <= 32;
  unsigned km = 0u;
  if (use_mask) {
    const uint32_t* tm = x.tmask + b * x.tmask_rb;
    if (x.tmask_on_b) {
      km = tm[c0 / 64];
    } else {
      km = tm[r0 / 64];
      if (r0 / 64 + 1 < x.tmask_rb) km |= tm[r0 / 64 + 1];
    }
    km = (unsigned)__builtin_amdgcn_readfirstlane((int)km);
  }
  auto next_chunk = [&](int after) -> int {  // wave-uniform; -1: none left
    if (!use_mask) return after + 1 < nch ? after + 1 : -1;
    const unsigned rest = after >= 31 ? 0u : (km & (0xffffffffu << (after + 1)));
    return rest ? __ffs((int)rest) - 1 : -1;
  };
  int kc = next_chunk(-1);
  if (kc >= 0) fetch(kc * GK);
  while (kc >= 0) {
    const int k0 = kc * GK;
    stage(k0);
    __syncthreads();
    const int kn = next_chunk(kc);
    if (kn >= 0) fetch(kn * GK);  // in flight during the MFMAs below
    const int kend = (m - k0 + 3) & ~3;   // (the last chunk's k beyond m are staged zeros: their MFMAs are skipped)
#pragma unroll
    for (int ks = 0; ks < GK; ks += 4) {
      if (ks < kend) {
        double af[GTI], bf[GTJ];
#pragma unroll
        for (int i = 0; i < GTI; ++i) af[i] = As[ks + lk][wr * (GTM / GWR) + i * 16 + li];  // A[i=l&15][k=l>>4]
#pragma unroll
        for (int j = 0; j < GTJ; ++j) bf[j] = Bs[ks + lk][wc * (GTN / GWC) + j * 16 + li];  // B[k=l>>4][j=l&15]
#pragma unroll
        for (int i = 0; i < GTI; ++i)
#pragma unroll
          for (int j = 0; j < GTJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    kc = kn;
  }
  // (tried first: a_hat = T a_f as one more COLUMN of the B operand of this product.  A conditional load inside the fetch splits its
  // batch of loads -- 337 against 274 us per launch at C5 -- and as a separate instantiation of the loop for the one tile column that
  // carries the vector still 314: the launch lasts as long as its slowest workgroups)
  // ---- epilogue: D[row = (l>>4) + 4*reg][col = l&15] ---------------------------------------------
  double* Cb = Cm + b * strideC;
  const double* Eb = ADD_E ? E + b * strideE : nullptr;
  if (x.neg_acc) {
#pragma unroll
    for (int i = 0; i < GTI; ++i)
#pragma unroll
      for (int j = 0; j < GTJ; ++j) acc[i][j] = -acc[i][j];
  }
  if (x.u1) {   // rank-one terms
    const double *pu1 = x.u1 + b * m, *pv1 = x.v1 + b * m;
    const double *pu2 = x.u2 ? x.u2 + b * m : nullptr, *pv2 = x.u2 ? x.v2 + b * m : nullptr;
#pragma unroll
    for (int j = 0; j < GTJ; ++j) {
      const int gc = c0 + wc * (GTN / GWC) + j * 16 + li;
      const double c1 = gc < m ? pv1[gc] : 0.0, c2 = (pu2 && gc < m) ? pv2[gc] : 0.0;
#pragma unroll
      for (int i = 0; i < GTI; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int gr = r0 + wr * (GTM / GWR) + i * 16 + lk + 4 * rg;
          if (gr < m) acc[i][j][rg] = fma(pu1[gr], c1, fma(pu2 ? pu2[gr] : 0.0, c2, acc[i][j][rg]));
        }
    }
  }
  bool e_here = ADD_E;
  if (ADD_E && x.emask) {   // (wave-uniform) any non-zero of E in this tile's rows x columns?
    const uint32_t* em = x.emask + b * x.tmask_rb;
    unsigned rows = em[r0 / 64];
    if (r0 / 64 + 1 < x.tmask_rb) rows |= em[r0 / 64 + 1];
    e_here = ((rows >> (c0 / 32)) & 3u) != 0u;
  }
  if (ADD_E && e_here) {
#pragma unroll
    for (int i = 0; i < GTI; ++i)
#pragma unroll
      for (int j = 0; j < GTJ; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int gr = r0 + wr * (GTM / GWR) + i * 16 + lk + 4 * rg;
          const int gc = c0 + wc * (GTN / GWC) + j * 16 + li;
          if (gr < m && gc < m && !(MIRROR && gr > gc)) acc[i][j][rg] += Eb[(long)gr * m + gc];
        }
  }
  if (MIRROR) {
    if (x.mrow) {  // (uniform over the launch)
      const double* zb = x.zv + b * m;
      double* rowbuf = &As[0][0];            // [GWC][GTM]   (the loop above ended on a barrier: As is free)
      double* colbuf = rowbuf + GWC * GTM;   // [GWR][GTN]
      double zc[GTJ], csum[GTJ];
#pragma unroll
      for (int j = 0; j < GTJ; ++j) {
        const int gc = c0 + wc * (GTN / GWC) + j * 16 + li;
        zc[j] = gc < m ? zb[gc] : 0.0;
        csum[j] = 0.0;
      }
#pragma unroll
      for (int i = 0; i < GTI; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int lr = wr * (GTM / GWR) + i * 16 + lk + 4 * rg, gr = r0 + lr;
          const double zr = gr < m ? zb[gr] : 0.0;
          double rs = 0.0;
#pragma unroll
          for (int j = 0; j < GTJ; ++j) {
            const int gc = c0 + wc * (GTN / GWC) + j * 16 + li;
            const double v = (gr < m && gc < m && gr <= gc) ? acc[i][j][rg] : 0.0;
            rs = fma(v, zc[j], rs);
            if (gr != gc) csum[j] = fma(v, zr, csum[j]);
          }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) rs += __shfl_xor(rs, o, 64);
          if (li == 0) rowbuf[wc * GTM + lr] = rs;
        }
#pragma unroll
      for (int j = 0; j < GTJ; ++j) {
        double cs = csum[j];
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        if (lk == 0) colbuf[wr * GTN + wc * (GTN / GWC) + j * 16 + li] = cs;
      }
      __syncthreads();
      if (tid < GTM) {
        double sv = 0.0;
#pragma unroll
        for (int w = 0; w < GWC; ++w) sv += rowbuf[w * GTM + tid];
        if (r0 + tid < m) x.mrow[(b * tiles_n + (tile % tiles_n)) * m + r0 + tid] = sv;
      } else if (tid < GTM + GTN) {
        const int cc = tid - GTM;
        double sv = 0.0;
#pragma unroll
        for (int w = 0; w < GWR; ++w) sv += colbuf[w * GTN + cc];
        if (c0 + cc < m) x.mcol[(b * tiles_m + (tile / tiles_n)) * m + c0 + cc] = sv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < GTI; ++i)
#pragma unroll
    for (int j = 0; j < GTJ; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        int gr = r0 + wr * (GTM / GWR) + i * 16 + lk + 4 * rg;
        int gc = c0 + wc * (GTN / GWC) + j * 16 + li;
        if (gr < m && gc < m) {
          if (MIRROR && gr > gc) continue;
          const double v = acc[i][j][rg];
          Cb[(long)gr * m + gc] = v;
          if (MIRROR && x.ctape) __builtin_nontemporal_store(v, x.ctape + b * x.ctape_stride + (long)gr * m + gc);
        }
      }
  if (MIRROR) {
    // The mirrored copy (c, r) of every element above the diagonal.  Round 5: through an LDS transpose, half a tile (64 rows) at a
    // time, so that the stores run along rows -- as 8-byte stores straight from the accumulators every lane of a wave-instruction
    // hit a different line (a third of S = W T' + R Q R''s launch at C5 once the all-zero K chunks were skipped)
    double* tp = x.ctape ? x.ctape + b * x.ctape_stride : nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();   // (the buffer: free of the loop's tiles / of the shares above / of the previous half)
      if ((wr >> 1) == half) {
#pragma unroll
        for (int i = 0; i < GTI; ++i)
#pragma unroll
          for (int j = 0; j < GTJ; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int rl = (wr & 1) * (GTM / GWR) + i * 16 + lk + 4 * rg;   // row within the half
              const int cl = wc * (GTN / GWC) + j * 16 + li;
              lds_raw[cl * 65 + rl] = acc[i][j][rg];
            }
      }
      __syncthreads();
      const int rbase = r0 + 64 * half;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cl = tid >> 3, rl = (tid & 7) + 8 * e;
        const int gr = rbase + rl, gc = c0 + cl;
        if (gr < gc && gc < m) {   // (gr < gc < m)
          const double v = lds_raw[cl * 65 + rl];
          Cb[(long)gc * m + gr] = v;
          if (tp) __builtin_nontemporal_store(v, tp + (long)gc * m + gr);
        }
      }
    }
  }
}

// RQR' per draw, initial state into the workspace, predicted tape slot 0
__global__ __launch_bounds__(256) void big_init_kernel(const KfFwdArgs A, double* a_ws, double* P_ws,
                                                       double* RQR_ws) {
  const long b = blockIdx.y;
  const int m = A.m, r = A.r, n = A.T;
  const long mm = (long)m * m;
  const double* gR = A.R + b * (long)m * r;
  const double* gQ = A.Q + b * (long)r * r;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < mm; e += (long)gridDim.x * 256) {
    int i = (int)(e / m), j = (int)(e - (long)i * m);
    double s = 0.0;
    for (int l = 0; l < r; ++l) {
      double rq = 0.0;
      for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
      s += rq * gR[j * r + l];
    }
    RQR_ws[b * mm + e] = s;
    double p0 = A.P0[b * mm + e];
    P_ws[b * mm + e] = p0;
    if (A.pred_P) A.pred_P[b * (long)(n + 1) * mm + e] = p0;
    if (e < m) {
      double a0 = A.a0[b * m + e];
      a_ws[b * m + e] = a0;
      if (A.pred_a) A.pred_a[b * (long)(n + 1) * m + e] = a0;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.logp[b] = 0.0;
    if (A.info) A.info[b] = 0;
  }
}

// one workgroup (256 threads) per draw; dynamic LDS: Z (p m), M (m p), K (m p), KF (m p), af (m),
// a (m), H/F/L (3 p p), v, ym, ob, misc
__global__ __launch_bounds__(256) void big_update_kernel(const KfFwdArgs A, int t, double* a_ws,
                                                         const double* P_ws, double* Pf_ws) {
  extern __shared__ double sm[];
  const long b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = A.m, p = A.p, n = A.T;
  const long mm = (long)m * m;
  const int pm = p * m, pp = p * p;
  double* Zl = sm;
  double* M = Zl + pm;
  double* K = M + pm;
  double* KF = K + pm;
  double* al = KF + pm;
  double* af = al + m;
  double* Hl = af + m;
  double* F = Hl + pp;
  double* L = F + pp;
  double* vl = L + pp;
  double* ym = vl + p;
  double* ob = ym + p;
  double* wv = ob + p;      // p + 4 scratch
  const double* P = P_ws + b * mm;
  double* Pf = Pf_ws + b * mm;
  double* a = a_ws + b * m;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double* yt = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L) + (long)t * p;

  for (int e = tid; e < pm; e += 256) Zl[e] = A.Z[b * pm + e];
  for (int e = tid; e < pp; e += 256) Hl[e] = A.H[b * pp + e];
  for (int e = tid; e < m; e += 256) al[e] = a[e];
  for (int q = tid; q < p; q += 256) {
    double yv = yt[q];
    bool miss = yv != yv;
    ym[q] = miss ? 0.0 : yv;
    ob[q] = miss ? 0.0 : 1.0;
  }
  __syncthreads();
  double nobs = 0.0;
  for (int q = 0; q < p; ++q) nobs += ob[q];
  double ll = 0.0;

  if (!univariate) {
    // v and M = P Z' (wave per row of P, lanes over k: coalesced)
    for (int q = tid; q < p; q += 256) {
      double s = ym[q] + dsign * (A.d ? A.d[b * p + q] : 0.0);
      for (int j = 0; j < m; ++j) s -= Zl[q * m + j] * al[j];
      vl[q] = ob[q] != 0.0 ? s : 0.0;
    }
    for (int j = wave; j < m; j += 4) {
      for (int q = 0; q < p; ++q) {
        double s = 0.0;
        for (int k = lane; k < m; k += 64) s += P[(long)j * m + k] * Zl[q * m + k];
        s = wave_sum(s);
        if (lane == 0) M[j * p + q] = ob[q] != 0.0 ? s : 0.0;
      }
    }
    __syncthreads();
    for (int e = tid; e < pp; e += 256) {
      int q1 = e / p, q2 = e - q1 * p;
      double s;
      if (ob[q1] != 0.0 && ob[q2] != 0.0) {
        s = Hl[e];
        for (int j = 0; j < m; ++j) s += Zl[q1 * m + j] * M[j * p + q2];
      } else {
        s = (q1 == q2) ? 1.0 : 0.0;
      }
      F[e] = s;
    }
    __syncthreads();
    if (tid == 0) {  // p x p Cholesky, quad form, log det (serial: p is small)
      int fail = 0;
      for (int j = 0; j < p; ++j) {
        double s = F[j * p + j];
        for (int k = 0; k < j; ++k) s -= L[j * p + k] * L[j * p + k];
        if (!(s > 0.0) && fail == 0) fail = j + 1;
        double dj = sqrt(s);
        L[j * p + j] = dj;
        for (int i = j + 1; i < p; ++i) {
          double t_ = F[i * p + j];
          for (int k = 0; k < j; ++k) t_ -= L[i * p + k] * L[j * p + k];
          L[i * p + j] = t_ / dj;
        }
      }
      double q2 = 0.0, ld = 0.0;
      for (int i = 0; i < p; ++i) {
        double t_ = vl[i];
        for (int k = 0; k < i; ++k) t_ -= L[i * p + k] * wv[k];
        t_ /= L[i * p + i];
        wv[i] = t_;
        q2 += t_ * t_;
        ld += log(L[i * p + i]);
      }
      double nconst = exact_const ? nobs : (A.filter == KF_FILTER_CHOLESKY ? (double)p : 1.0);
      wv[p] = (nobs == 0.0) ? 0.0 : (-0.5 * (nconst * KF_LOG_2PI + q2) - ld);
      if (fail != 0 && A.info && A.info[b] == 0) A.info[b] = ((t + 1) << 8) | fail;
    }
    __syncthreads();
    ll = wv[p];
    for (int j = tid; j < m; j += 256) {  // K row j = M row j F^-1
      for (int i = 0; i < p; ++i) {
        double t_ = M[j * p + i];
        for (int k = 0; k < i; ++k) t_ -= L[i * p + k] * K[j * p + k];
        K[j * p + i] = t_ / L[i * p + i];
      }
      for (int i = p - 1; i >= 0; --i) {
        double t_ = K[j * p + i];
        for (int k = i + 1; k < p; ++k) t_ -= L[k * p + i] * K[j * p + k];
        K[j * p + i] = t_ / L[i * p + i];
      }
      double s = al[j];
      for (int q = 0; q < p; ++q) s += K[j * p + q] * vl[q];
      af[j] = s;
      for (int q = 0; q < p; ++q) {
        double kf = 0.0;
        for (int q2 = 0; q2 < p; ++q2) kf += K[j * p + q2] * F[q2 * p + q];
        KF[j * p + q] = kf;
      }
    }
    __syncthreads();
    double* o_fP = A.filt_P ? A.filt_P + (b * (long)n + t) * mm : nullptr;
    for (long e = tid; e < mm; e += 256) {
      int i = (int)(e / m), j = (int)(e - (long)i * m);
      double s = P[e];
      for (int q = 0; q < p; ++q) {
        double kj = K[j * p + q];
        s -= K[i * p + q] * M[j * p + q];
        s -= M[i * p + q] * kj;
        s += KF[i * p + q] * kj;
      }
      Pf[e] = s;
      if (o_fP) __builtin_nontemporal_store(s, o_fP + e);
    }
  } else {
    // UnivariateFilter (:460-505): p sequential scalar updates on (al, Pf)
    for (long e = tid; e < mm; e += 256) Pf[e] = P[e];
    __threadfence_block();
    __syncthreads();
    double llsum = 0.0, ncount = 0.0;
    for (int q = 0; q < p; ++q) {
      const bool obs = ob[q] != 0.0;
      for (int j = wave; j < m; j += 4) {
        double s = 0.0;
        for (int k = lane; k < m; k += 64) s += Pf[(long)j * m + k] * Zl[q * m + k];
        s = wave_sum(s);
        if (lane == 0) M[j] = obs ? s : 0.0;
      }
      __syncthreads();
      double vq = ym[q] - (A.d ? A.d[b * p + q] : 0.0);
      double Fq = obs ? Hl[q * p + q] : 0.0;
      for (int j = 0; j < m; ++j) {
        vq -= Zl[q * m + j] * al[j];
        if (obs) Fq += Zl[q * m + j] * M[j];
      }
      if (!obs) vq = 0.0;
      const bool degenerate = (Fq == 0.0) || !obs;
      if (degenerate) Fq += 1e-8;
      const double keep = degenerate ? 0.0 : 1.0;
      const double invF = 1.0 / Fq;
      __syncthreads();  // everyone has read al before it changes
      for (int j = tid; j < m; j += 256) al[j] += (M[j] * invF * keep) * (vq * keep);
      for (long e = tid; e < mm; e += 256) {
        int i = (int)(e / m), j = (int)(e - (long)i * m);
        double ki = M[i] * invF * keep, kj = M[j] * invF * keep;
        Pf[e] -= ki * kj * Fq * keep;
      }
      double lli = (log(Fq) + vq * vq * invF) * keep;
      llsum += lli;
      ncount += exact_const ? (obs ? 1.0 : 0.0) : ((lli != 0.0) ? 1.0 : 0.0);
      __threadfence_block();
      __syncthreads();
    }
    ll = -0.5 * (ncount * KF_LOG_2PI + llsum);
    for (int j = tid; j < m; j += 256) af[j] = al[j];
    if (A.filt_P) {
      double* o_fP = A.filt_P + (b * (long)n + t) * mm;
      for (long e = tid; e < mm; e += 256) __builtin_nontemporal_store(Pf[e], o_fP + e);
    }
  }
  __syncthreads();
  if (tid == 0) {
    A.logp[b] += ll;
    if (A.ll_obs) A.ll_obs[b * (long)n + t] = ll;
  }
  if (A.filt_a)
    for (int j = tid; j < m; j += 256) A.filt_a[(b * (long)n + t) * m + j] = af[j];
  // a_hat = T a_f + c  (wave per row)
  const double* Tb = A.Tm + b * mm;
  for (int i = wave; i < m; i += 4) {
    double s = 0.0;
    for (int k = lane; k < m; k += 64) s += Tb[(long)i * m + k] * af[k];
    s = wave_sum(s);
    if (lane == 0) {
      s += A.c ? A.c[b * m + i] : 0.0;
      a[i] = s;
      if (A.pred_a) A.pred_a[(b * (long)(n + 1) + t + 1) * m + i] = s;
    }
  }
}


// ---- split update for Standard / Cholesky / Single (parallel over rows and elements) ----------
// (the fused big_update_kernel above, one workgroup per draw, remains the Univariate path)

constexpr int BIG_PMAX = 8;   // observations handled per pass by the row kernels
constexpr int MV_ROWS = 16;   // rows of P per workgroup in big_mvec_kernel

// M[b][j][q] = sum_k P[b][j][k] Z[b][q][k]  (0 for missing q); wave per row, lanes over k
__global__ __launch_bounds__(256) void big_mvec_kernel(const KfFwdArgs A, int t, const double* P_ws,
                                                       double* M_ws) {
  const long b = blockIdx.y;
  const int m = A.m, p = A.p, n = A.T;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* P = P_ws + b * (long)m * m;
  const double* Zg = A.Z + b * (long)p * m;
  const double* yt = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L) + (long)t * p;
  for (int q0 = 0; q0 < p; q0 += BIG_PMAX) {
    for (int rr = wave; rr < MV_ROWS; rr += 4) {
      const int j = blockIdx.x * MV_ROWS + rr;
      if (j >= m) break;
      double acc[BIG_PMAX];
#pragma unroll
      for (int q = 0; q < BIG_PMAX; ++q) acc[q] = 0.0;
      for (int k = lane; k < m; k += 64) {
        const double pk = P[(long)j * m + k];
#pragma unroll
        for (int q = 0; q < BIG_PMAX; ++q)
          if (q0 + q < p) acc[q] = fma(pk, Zg[(q0 + q) * m + k], acc[q]);
      }
#pragma unroll
      for (int q = 0; q < BIG_PMAX; ++q) {
        if (q0 + q < p) {
          double s = wave_sum(acc[q]);
          const double yv = yt[q0 + q];
          if (lane == 0) M_ws[(b * m + j) * p + q0 + q] = (yv != yv) ? 0.0 : s;
        }
      }
    }
  }
}

// per draw: v, F, chol, K, KF, a_f, ll, tapes of the filtered state, a_hat = T a_f + c
// With mrow (p == 1, t >= 1): m = P z comes from the shares the previous step's S = W T' + R Q R' launch left behind
// (big_gemm_kernel<.., MIRROR>), summed here in a fixed order and stored to M_out for the next launch.
__global__ __launch_bounds__(256) void big_gain_kernel(const KfFwdArgs A, int t, const double* a_ws,
                                                       const double* M_ws, double* K_ws, double* KF_ws,
                                                       double* af_ws, const double* mrow, const double* mcol,
                                                       int tiles_n, int tiles_m, double* M_out) {
  extern __shared__ double sm[];
  const long b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = A.m, p = A.p, n = A.T;
  const long mm = (long)m * m;
  const int pm = p * m, pp = p * p;
  double* Zl = sm;
  double* M = Zl + pm;
  double* K = M + pm;
  double* al = K + pm;
  double* af = al + m;
  double* Hl = af + m;
  double* F = Hl + pp;
  double* L = F + pp;
  double* vl = L + pp;
  double* ym = vl + p;
  double* ob = ym + p;
  double* wv = ob + p;
  const double* a = a_ws + b * m;
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;
  const double* yt = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L) + (long)t * p;
  if (mrow) {  // p == 1
    const double y0 = yt[0];
    for (int e = tid; e < m; e += 256) {
      Zl[e] = A.Z[b * m + e];
      const int tr = e / GTM, tc = e / GTN;
      double sv = 0.0;
      for (int c = 0; c < tiles_n; ++c)   // the tiles of this row's tile row that are not wholly below the diagonal
        if (tr * GTM <= c * GTN + GTN - 1) sv += mrow[(b * tiles_n + c) * m + e];
      for (int r = 0; r < tiles_m; ++r)
        if (r * GTM <= tc * GTN + GTN - 1) sv += mcol[(b * tiles_m + r) * m + e];
      sv = (y0 != y0) ? 0.0 : sv;
      M[e] = sv;
      M_out[b * m + e] = sv;
    }
  } else {
    for (int e = tid; e < pm; e += 256) {
      Zl[e] = A.Z[b * pm + e];
      M[e] = M_ws[b * pm + e];
    }
  }
  for (int e = tid; e < pp; e += 256) Hl[e] = A.H[b * pp + e];
  for (int e = tid; e < m; e += 256) al[e] = a[e];
  for (int q = tid; q < p; q += 256) {
    double yv = yt[q];
    bool miss = yv != yv;
    ym[q] = miss ? 0.0 : yv;
    ob[q] = miss ? 0.0 : 1.0;
  }
  __syncthreads();
  double nobs = 0.0;
  for (int q = 0; q < p; ++q) nobs += ob[q];
  // v (p dots) and F (p*p dots) of length m: one wave per dot, lanes over the state index
  for (int e = wave; e < p + pp; e += 4) {
    if (e < p) {
      const int q = e;
      double s = 0.0;
      for (int j = lane; j < m; j += 64) s += Zl[q * m + j] * al[j];
      s = wave_sum(s);
      if (lane == 0)
        vl[q] = ob[q] != 0.0 ? (ym[q] + dsign * (A.d ? A.d[b * p + q] : 0.0)) - s : 0.0;
    } else {
      const int f = e - p, q1 = f / p, q2 = f - q1 * p;
      double s = 0.0;
      if (ob[q1] != 0.0 && ob[q2] != 0.0) {
        for (int j = lane; j < m; j += 64) s += Zl[q1 * m + j] * M[j * p + q2];
        s = wave_sum(s) + Hl[f];
      } else {
        s = (q1 == q2) ? 1.0 : 0.0;
      }
      if (lane == 0) F[f] = s;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int fail = 0;
    for (int j = 0; j < p; ++j) {
      double s = F[j * p + j];
      for (int k = 0; k < j; ++k) s -= L[j * p + k] * L[j * p + k];
      if (!(s > 0.0) && fail == 0) fail = j + 1;
      double dj = sqrt(s);
      L[j * p + j] = dj;
      for (int i = j + 1; i < p; ++i) {
        double t_ = F[i * p + j];
        for (int k = 0; k < j; ++k) t_ -= L[i * p + k] * L[j * p + k];
        L[i * p + j] = t_ / dj;
      }
    }
    double q2 = 0.0, ld = 0.0;
    for (int i = 0; i < p; ++i) {
      double t_ = vl[i];
      for (int k = 0; k < i; ++k) t_ -= L[i * p + k] * wv[k];
      t_ /= L[i * p + i];
      wv[i] = t_;
      q2 += t_ * t_;
      ld += log(L[i * p + i]);
    }
    double nconst = exact_const ? nobs : (A.filter == KF_FILTER_CHOLESKY ? (double)p : 1.0);
    double ll = (nobs == 0.0) ? 0.0 : (-0.5 * (nconst * KF_LOG_2PI + q2) - ld);
    A.logp[b] += ll;
    if (A.ll_obs) A.ll_obs[b * (long)n + t] = ll;
    if (fail != 0 && A.info && A.info[b] == 0) A.info[b] = ((t + 1) << 8) | fail;
  }
  __syncthreads();
  for (int j = tid; j < m; j += 256) {
    for (int i = 0; i < p; ++i) {
      double t_ = M[j * p + i];
      for (int k = 0; k < i; ++k) t_ -= L[i * p + k] * K[j * p + k];
      K[j * p + i] = t_ / L[i * p + i];
    }
    for (int i = p - 1; i >= 0; --i) {
      double t_ = K[j * p + i];
      for (int k = i + 1; k < p; ++k) t_ -= L[k * p + i] * K[j * p + k];
      K[j * p + i] = t_ / L[i * p + i];
    }
    double s = al[j];
    for (int q = 0; q < p; ++q) {
      s += K[j * p + q] * vl[q];
      K_ws[b * pm + j * p + q] = K[j * p + q];
    }
    af_ws[b * m + j] = s;
    if (A.filt_a) A.filt_a[(b * (long)n + t) * m + j] = s;
    for (int q = 0; q < p; ++q) {
      double kf = 0.0;
      for (int q2 = 0; q2 < p; ++q2) kf += K[j * p + q2] * F[q2 * p + q];
      KF_ws[b * pm + j * p + q] = kf;
    }
  }
  (void)mm;
  (void)af;
}

// a_hat = T a_f + c : wave per row of T, MV_ROWS rows per workgroup
__global__ __launch_bounds__(256) void big_apred_kernel(const KfFwdArgs A, int t, const double* af_ws,
                                                        double* a_ws) {
  __shared__ double afs[1024];
  const long b = blockIdx.y;
  const int m = A.m, n = A.T;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* Tb = A.Tm + b * (long)m * m;
  const double* afg = af_ws + b * m;
  const bool in_lds = m <= 1024;
  if (in_lds) {
    for (int k = threadIdx.x; k < m; k += 256) afs[k] = afg[k];
    __syncthreads();
  }
  for (int rr = wave; rr < MV_ROWS; rr += 4) {
    const int i = blockIdx.x * MV_ROWS + rr;
    if (i >= m) break;
    double s = 0.0;
    for (int k = lane; k < m; k += 64) s += Tb[(long)i * m + k] * (in_lds ? afs[k] : afg[k]);
    s = wave_sum(s);
    if (lane == 0) {
      s += A.c ? A.c[b * m + i] : 0.0;
      a_ws[b * m + i] = s;
      if (A.pred_a) A.pred_a[(b * (long)(n + 1) + t + 1) * m + i] = s;
    }
  }
}

// P_f = P - K M' - M K' + (K F) K'   elementwise, rows i0..i0+PF_ROWS of one draw per workgroup
constexpr int PF_ROWS = 8;
__global__ __launch_bounds__(256) void big_pf_kernel(const KfFwdArgs A, int t, const double* P_ws,
                                                     const double* M_ws, const double* K_ws,
                                                     const double* KF_ws, double* Pf_ws) {
  extern __shared__ double sm[];
  const long b = blockIdx.y;
  const int m = A.m, p = A.p, n = A.T;
  const long mm = (long)m * m;
  const int pm = p * m;
  double* M = sm;
  double* K = M + pm;
  for (int e = threadIdx.x; e < pm; e += 256) {
    M[e] = M_ws[b * pm + e];
    K[e] = K_ws[b * pm + e];
  }
  __syncthreads();
  const double* KF = KF_ws + b * pm;
  const double* P = P_ws + b * mm;
  double* Pf = Pf_ws + b * mm;
  double* o_fP = A.filt_P ? A.filt_P + (b * (long)n + t) * mm : nullptr;
  const int i0 = blockIdx.x * PF_ROWS;
  for (int rr = 0; rr < PF_ROWS; ++rr) {
    const int i = i0 + rr;
    if (i >= m) break;
    for (int j = threadIdx.x; j < m; j += 256) {
      const long e = (long)i * m + j;
      double s = P[e];
      for (int q = 0; q < p; ++q) {
        const double kj = K[j * p + q];
        s -= K[i * p + q] * M[j * p + q];
        s -= M[i * p + q] * kj;
        s += KF[i * p + q] * kj;
      }
      Pf[e] = s;
      if (o_fP) __builtin_nontemporal_store(s, o_fP + e);
    }
  }
}

// P = upper triangle of S mirrored (S = T P_f T' + RQR' is symmetric up to rounding; the reference's
// (S + S') / 2, :221, differs from this at rounding level only) via 32 x 32 LDS transposes;
// predicted covariance tape.  Reads the upper half of S only: the GEMM skipped the tiles below.
__global__ __launch_bounds__(256) void big_sym_kernel(const double* __restrict__ S, double* __restrict__ P,
                                                      double* pred_P, long tape_stride, int m) {
  __shared__ double t1[32][33];
  const long b = blockIdx.z;
  const long mm = (long)m * m;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;  // tile (bi, bj), bi <= bj, also writes (bj, bi)
  const double* Sb = S + b * mm;
  double* Pb = P + b * mm;
  double* Tb = pred_P ? pred_P + b * tape_stride : nullptr;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int rr = ty; rr < 32; rr += 8) {
    const int r1 = bi * 32 + rr, c1 = bj * 32 + tx;
    t1[rr][tx] = (r1 < m && c1 < m) ? Sb[(long)r1 * m + c1] : 0.0;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int r1 = bi * 32 + rr, c1 = bj * 32 + tx;
    if (r1 < m && c1 < m) {
      const double v = (bi != bj || rr <= tx) ? t1[rr][tx] : t1[tx][rr];
      Pb[(long)r1 * m + c1] = v;
      if (Tb) __builtin_nontemporal_store(v, Tb + (long)r1 * m + c1);
    }
    if (bi != bj) {
      const int r2 = bj * 32 + rr, c2 = bi * 32 + tx;
      if (r2 < m && c2 < m) {
        const double v = t1[tx][rr];
        Pb[(long)r2 * m + c2] = v;
        if (Tb) __builtin_nontemporal_store(v, Tb + (long)r2 * m + c2);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Sparse transition (round 5).  The structural models that reach m > 64 -- seasonal dummies, trigonometric terms, companion
// forms -- have a T with a handful of non-zeros per row (C5: 1 103 of 134 689), and T P_f T' + R Q R' with such a T is a
// gather of rows and columns of P_f, not a product: m^2 nnz-per-row flops against 4 m^3, one pass over P instead of the
// six of the two MFMA products (T, P -> W;  W, T, R Q R' -> P).  The recursion then runs at HBM speed: P does not fit on chip
// (128 draws x 1.08 MB), so 2 m^2 doubles per draw and step -- read P, write P -- is the floor.
//   bsp_csr_kernel   once per call: T -> CSR per draw (rows with more than BSP_TH non-zeros -- the seasonal sum row -- are
//                    kept out of it as "dense rows", BSP_ND at most); *ok = 0 when a draw does not fit -> the MFMA path runs
//   bsp_step_kernel  one time step: a workgroup owns R rows i of one draw.  Stage 1  w_i = sum_{k in nz(i)} T[i,k] P_f[k,:]
//                    (coalesced reads of the rows k of P, P_f = P - k m' - m k' + (k f) k' formed on the fly with the MFMA
//                    path's order of operations) into LDS; stage 2  S[i,j] = sum_{l in nz(j)} w_i[l] T[j,l] + (R Q R')[i,j]
//                    for every j, written along rows.  The same workgroup leaves what the next step needs of its rows:
//                    (S z)[i] (the next m = P z), a_hat[i], and (S t_d)[i] for every dense row d.
//   dense rows       row d of T P_f is P_f t_d (P_f is symmetric up to rounding), i.e. P t_d corrected by three dot products
//                    -- and P t_d was left by the previous step's row sums, so that no workgroup reads all of P; column d of S
//                    is R dot products of length m per workgroup.
constexpr int BSP_TH = 32;
constexpr int BSP_ND = 4;
constexpr int BSP_CAPF = 4;   // CSR capacity: BSP_CAPF * m entries per draw

using BspCsr = KfBspCsr;   // (kf_internal.h: kf_smoother_bf_big.hip builds one of T' too)

__global__ __launch_bounds__(256) void bsp_csr_kernel(const double* __restrict__ T, int m, BspCsr c) {
  __shared__ int cnt[BIG_PFIX_M + 1];
  __shared__ int s_fits;
  const long b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double* Tb = T + b * (long)m * m;
  for (int row = wave; row < m; row += 4) {
    int n = 0;
    for (int k0 = 0; k0 < m; k0 += 64) {
      const int k = k0 + lane;
      const double v = k < m ? Tb[(long)row * m + k] : 0.0;
      n += __popcll(__ballot(v != 0.0 || v != v));   // (a NaN counts: it must reach the result)
    }
    if (lane == 0) cnt[row] = n;
  }
  __syncthreads();
  if (tid == 0) {
    int nd = 0, off = 0;
    for (int row = 0; row < m; ++row) {
      int n = cnt[row];
      if (n > BSP_TH) {
        if (nd < BSP_ND) c.drow[b * BSP_ND + nd] = row;
        ++nd;
        n = 0;
      }
      cnt[row] = off;
      off += n;
    }
    cnt[m] = off;
    const int fits = (nd <= BSP_ND && off <= c.cap) ? 1 : 0;
    for (int d = nd; d < BSP_ND; ++d) c.drow[b * BSP_ND + d] = -1;
    c.nd[b] = nd <= BSP_ND ? nd : BSP_ND;
    s_fits = fits;
    if (!fits) atomicAnd(c.ok, 0);
    atomicMax(c.ok + 1, nd <= BSP_ND ? nd : BSP_ND);
  }
  __syncthreads();
  for (int e = tid; e <= m; e += 256) c.rowptr[b * (m + 1) + e] = cnt[e];
  if (!s_fits) return;
  for (int row = wave; row < m; row += 4) {
    int off = cnt[row];
    if (cnt[row + 1] == off) continue;   // empty, or a dense row
    for (int k0 = 0; k0 < m; k0 += 64) {
      const int k = k0 + lane;
      const double v = k < m ? Tb[(long)row * m + k] : 0.0;
      const bool nz = v != 0.0 || v != v;
      const unsigned long long bal = __ballot(nz);
      if (nz) {
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        c.col[b * c.cap + pos] = k;
        c.val[b * c.cap + pos] = v;
      }
      off += __popcll(bal);
    }
  }
}

// pt[b][d][i] = sum_k P[i][k] T[drow_d][k]  (step 0; afterwards bsp_step_kernel leaves it): wave per row of P
__global__ __launch_bounds__(256) void bsp_pt_kernel(const double* __restrict__ P, const double* __restrict__ T, int m, BspCsr c,
                                                     double* __restrict__ pt) {
  const long b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (c.ok[0] == 0) return;   // (a captured call replayed on a T that no longer fits the CSR: bsp_guard_kernel reports it)
  const int nd = c.nd[b];
  if (nd == 0) return;
  const double* Pb = P + b * (long)m * m;
  const double* Tb = T + b * (long)m * m;
  for (int rr = wave; rr < 16; rr += 4) {
    const int i = blockIdx.x * 16 + rr;
    if (i >= m) break;
    for (int d = 0; d < nd; ++d) {
      const int jd = c.drow[b * BSP_ND + d];
      double s = 0.0;
      for (int k = lane; k < m; k += 64) s = fma(Pb[(long)i * m + k], Tb[(long)jd * m + k], s);
      s = wave_sum(s);
      if (lane == 0) pt[(b * BSP_ND + d) * m + i] = s;
    }
  }
}

// Per-call vectors of the step kernel: with S = T P_f T' + E and w_i = row i of T P_f, a row sum of S against a fixed vector v is
// (S v)[i] = w_i . (T' v) + (E v)[i] -- a dot product over the row the workgroup already holds in LDS.  v = z gives the next
// step's m = P z, v = t_d (dense row d of T) the next step's P t_d.
//   tz[b][l] = sum_j T[j][l] z[j]     ez[b][i] = sum_j E[i][j] z[j]     ttd[b][d][l] = sum_j T[j][l] T[d][j]     etd[b][d][i] = sum_j E[i][j] T[d][j]
// grid (ceil(m / 64), B): 64 columns l (lanes) x rows j (waves) for the transposed products, then 64 rows i for the plain ones.
__global__ __launch_bounds__(256) void bsp_vec_kernel(const double* __restrict__ T, const double* __restrict__ E, const double* __restrict__ Z,
                                                      int m, BspCsr c, double* __restrict__ tz, double* __restrict__ ez,
                                                      double* __restrict__ ttd, double* __restrict__ etd) {
  __shared__ double part[4][1 + BSP_ND][64];
  const long b = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (c.ok[0] == 0) return;
  const double* Tb = T + b * (long)m * m;
  const double* Eb = E + b * (long)m * m;
  const double* zb = Z + b * m;
  const int nd = c.nd[b];
  int dr[BSP_ND];
#pragma unroll
  for (int d = 0; d < BSP_ND; ++d) dr[d] = d < nd ? c.drow[b * BSP_ND + d] : -1;
  const int l = blockIdx.x * 64 + lane;
  double acc[1 + BSP_ND];
#pragma unroll
  for (int q = 0; q <= BSP_ND; ++q) acc[q] = 0.0;
  for (int j = wave; j < m; j += 4) {
    const double tv = l < m ? Tb[(long)j * m + l] : 0.0;
    acc[0] = fma(tv, zb[j], acc[0]);
#pragma unroll
    for (int d = 0; d < BSP_ND; ++d)
      if (d < nd) acc[1 + d] = fma(tv, Tb[(long)dr[d] * m + j], acc[1 + d]);
  }
#pragma unroll
  for (int q = 0; q <= BSP_ND; ++q) part[wave][q][lane] = acc[q];
  __syncthreads();
  if (wave == 0 && l < m) {
#pragma unroll
    for (int q = 0; q <= BSP_ND; ++q) {
      if (q > nd) break;
      const double sv = ((part[0][q][lane] + part[1][q][lane]) + part[2][q][lane]) + part[3][q][lane];
      if (q == 0) tz[b * m + l] = sv;
      else ttd[(b * BSP_ND + q - 1) * m + l] = sv;
    }
  }
  for (int rr = wave; rr < 64; rr += 4) {
    const int i = blockIdx.x * 64 + rr;
    if (i >= m) break;
    double a2[1 + BSP_ND];
#pragma unroll
    for (int q = 0; q <= BSP_ND; ++q) a2[q] = 0.0;
    for (int j = lane; j < m; j += 64) {
      const double ev = Eb[(long)i * m + j];
      a2[0] = fma(ev, zb[j], a2[0]);
#pragma unroll
      for (int d = 0; d < BSP_ND; ++d)
        if (d < nd) a2[1 + d] = fma(ev, Tb[(long)dr[d] * m + j], a2[1 + d]);
    }
#pragma unroll
    for (int q = 0; q <= BSP_ND; ++q) {
      if (q > nd) break;
      const double sv = wave_sum(a2[q]);
      if (lane == 0) {
        if (q == 0) ez[b * m + i] = sv;
        else etd[(b * BSP_ND + q - 1) * m + i] = sv;
      }
    }
  }
}

__device__ __forceinline__ double bsp_readlane_f64(double v, int src) {   // src wave-uniform
  const long long bits = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits & 0xffffffffll), src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum over the wavefront, the same value in every lane: four DPP rotations inside the rows of 16 lanes, then the four row totals
// through SGPRs -- ~25 instructions, against six dependent ds_bpermute round trips for wave_sum()
template <int CTRL>
__device__ __forceinline__ double bsp_dpp(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bsp_wave_sum(double x) {
  asm volatile("" : "+v"(x));
  x += bsp_dpp<0x128>(x);   // row_ror:8
  x += bsp_dpp<0x124>(x);
  x += bsp_dpp<0x122>(x);
  x += bsp_dpp<0x121>(x);
  return ((bsp_readlane_f64(x, 0) + bsp_readlane_f64(x, 16)) + bsp_readlane_f64(x, 32)) + bsp_readlane_f64(x, 48);
}

struct BspStep {
  const double* P_in;    // [B][m][m]
  double* P_out;         // [B][m][m]  (another buffer: other workgroups still read P_in)
  const double* K;       // [B][m]  gain
  const double* Mc;      // [B][m]  m = P z of this step (0 when the observation is missing)
  const double* KF;      // [B][m]  k f
  const double* af;      // [B][m]  filtered state
  const double* pt_in;   // [B][BSP_ND][m]  P t_d
  double* Mn;            // [B][m]  the next step's m
  double* pt_out;        // [B][BSP_ND][m]
  double* a_ws;          // [B][m]  a_hat (nobody reads the old one in this launch)
  const double* E;       // [B][m][m]  R Q R'
  const uint32_t* emask; // block map of E or NULL
  int emask_rb;
  const double* tz;      // bsp_vec_kernel's vectors
  const double* ez;
  const double* ttd;
  const double* etd;
};

constexpr int BSP_CG = 6;   // columns per lane and pass of stage 1 (384 columns)

// behind a CAPTURED sparse chain: if the CSR build of this replay found a T that does not fit, no draw has a result
__global__ void bsp_guard_kernel(const int* ok, double* logp, int32_t* info, long B) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (ok[0] != 0 || b >= B) return;
  logp[b] = __builtin_nan("");
  if (info) info[b] = 0x7fffff02;
}

template <int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void bsp_step_kernel(const KfFwdArgs A, int t, const BspCsr c, const BspStep x, int nblk) {
  extern __shared__ double sm[];
  constexpr int RW = R / 4;   // rows per wave in stage 1 and in the dot products
  constexpr int NE = R >= 16 ? 4 : 2;   // CSR entries per pass of stage 1 (a wave's rows hold ~1.1 RW of them at C5)
  const int m = A.m, n = A.T;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // all row blocks of one draw on one XCD (they share rows of P, the CSR and the vectors in that L2): see big_gemm_kernel
  const long L = blockIdx.x;
  const long b = (L & 7) + 8 * (L / (8L * nblk));
  if (b >= A.B) return;
  if (c.ok[0] == 0) return;
  const int i0 = (int)((L >> 3) % nblk) * R;
  const long mm = (long)m * m;
  double* w = sm;                    // [R][m]
  double* vK = w + (long)R * m;      // [m]
  double* vM = vK + m;
  double* vKF = vM + m;
  double* vAF = vKF + m;
  double* vTZ = vAF + m;             // T' z (bsp_vec_kernel)
  double* sd = vTZ + m;              // [BSP_ND][R]   column d of S for this block's rows
  const double* Pb = x.P_in + b * mm;
  const double* Tb = A.Tm + b * mm;
  const int* rp = c.rowptr + b * (m + 1);
  const int* col = c.col + b * (long)c.cap;
  const double* val = c.val + b * (long)c.cap;
  const int nd = c.nd[b];
  int dr[BSP_ND];
#pragma unroll
  for (int d = 0; d < BSP_ND; ++d) dr[d] = d < nd ? c.drow[b * BSP_ND + d] : -1;
  // the row pointers this wave needs in stage 1 (lanes 0 .. RW), and this thread's first stage-2 column: issued before anything waits
  const int q0 = wave * RW;
  const int rp_mine = rp[min(i0 + q0 + min(lane, RW), m)];
  // (every load whose address is known now is issued now: a workgroup lives ~12 us, most of it round trips to L2 / HBM)
  int s2_eb[2] = {0, 0}, s2_ee[2] = {0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (tid + 256 * h < m) {
      s2_eb[h] = rp[tid + 256 * h];
      s2_ee[h] = rp[tid + 256 * h + 1];
    }
  unsigned ebits = 0xffffffffu;
  if (x.emask) ebits = x.emask[b * x.emask_rb + i0 / 64];
  // per-row scalars of the wave's rows (lanes 0 .. RW - 1): E z, E t_0
  double ez_mine = 0.0, etd0_mine = 0.0;
  if (lane < RW && i0 + q0 + lane < m) {
    ez_mine = x.ez[b * m + i0 + q0 + lane];
    if (nd > 0) etd0_mine = x.etd[(b * BSP_ND + 0) * m + i0 + q0 + lane];
  }
  for (int k = tid; k < m; k += 256) {
    vK[k] = x.K[b * m + k];
    vM[k] = x.Mc[b * m + k];
    vKF[k] = x.KF[b * m + k];
    vAF[k] = x.af[b * m + k];
    vTZ[k] = x.tz[b * m + k];
  }
  // stage-2 prefetch: the first two entries of this thread's first two columns
  int s2_c0[2] = {0, 0}, s2_c1[2] = {0, 0};
  double s2_v0[2] = {0.0, 0.0}, s2_v1[2] = {0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (s2_eb[h] < s2_ee[h]) {
      s2_c0[h] = col[s2_eb[h]];
      s2_v0[h] = val[s2_eb[h]];
    }
    if (s2_eb[h] + 1 < s2_ee[h]) {
      s2_c1[h] = col[s2_eb[h] + 1];
      s2_v1[h] = val[s2_eb[h] + 1];
    }
  }
  __syncthreads();
  // ---- stage 1: rows q0 .. q0 + RW of the block by this wave: w_r += T[i,k] P_f[k,:] over the wave's contiguous CSR range
  {
    int rpw[RW + 1];
#pragma unroll
    for (int q = 0; q <= RW; ++q) rpw[q] = __builtin_amdgcn_readlane(rp_mine, q);
    const int e_beg = rpw[0], e_end = rpw[RW];
    // the first dense row's vectors for the dot products below (columns lane + 64 g < 64 BSP_CG), in flight during the gather
    double tj0[BSP_CG], tt0[BSP_CG];
#pragma unroll
    for (int g = 0; g < BSP_CG; ++g) {
      const int l = lane + 64 * g;
      const bool on = nd > 0 && l < m;
      tj0[g] = on ? Tb[(long)dr[0] * m + l] : 0.0;
      tt0[g] = on ? x.ttd[(b * BSP_ND + 0) * m + l] : 0.0;
    }
    int prev_row = -1;
    for (int chunk = e_beg; chunk < e_end; chunk += 64) {
      const int me = chunk + lane;
      const bool have = me < e_end;
      const int ck = have ? col[me] : 0;
      const double cv = have ? val[me] : 0.0;
      int cr = 0;
#pragma unroll
      for (int q = 1; q < RW; ++q) cr += (me >= rpw[q]) ? 1 : 0;
      const int nloc = min(64, e_end - chunk);
      for (int e = 0; e < nloc; e += NE) {
        int kq[NE], rq[NE];
        double tq[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
          const int src = min(e + u, nloc - 1);
          kq[u] = __builtin_amdgcn_readlane(ck, src);
          rq[u] = q0 + __builtin_amdgcn_readlane(cr, src);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(__double_as_longlong(cv) & 0xffffffffll), src);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)__double_as_longlong(cv) >> 32), src);
          tq[u] = (e + u < nloc) ? __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) : 0.0;
        }
        for (int cg = 0; cg < m; cg += 64 * BSP_CG) {
          double pv[NE][BSP_CG];
#pragma unroll
          for (int u = 0; u < NE; ++u)
#pragma unroll
            for (int g = 0; g < BSP_CG; ++g) {
              const int l = cg + lane + 64 * g;
              pv[u][g] = (l < m && e + u < nloc) ? Pb[(long)kq[u] * m + l] : 0.0;
            }
#pragma unroll
          for (int u = 0; u < NE; ++u) {
            if (e + u < nloc) {
              const bool first = rq[u] != (u == 0 ? prev_row : rq[u - 1]);   // (entries are sorted by row)
              const double kk = vK[kq[u]], mk = vM[kq[u]], kf = vKF[kq[u]];
#pragma unroll
              for (int g = 0; g < BSP_CG; ++g) {
                const int l = cg + lane + 64 * g;
                if (l < m) {
                  const double ml = vM[l], kl = vK[l];
                  double sv = pv[u][g];
                  sv -= kk * ml;
                  sv -= mk * kl;
                  sv += kf * kl;
                  w[(long)rq[u] * m + l] = first ? tq[u] * sv : fma(tq[u], sv, w[(long)rq[u] * m + l]);
                }
              }
            }
          }
        }
        prev_row = rq[min(NE, nloc - e) - 1];
      }
    }
    // a_hat of the wave's rows; its dense rows (none of their entries is in the CSR) and its empty rows
    for (int q = 0; q < RW; ++q) {
      const int r = q0 + q, i = i0 + r;
      if (i >= m) break;
      int dsel = -1;
#pragma unroll
      for (int d = 0; d < BSP_ND; ++d)
        if (dr[d] == i) dsel = d;
      if (dsel >= 0) {   // P_f t = P t - m (k.t) - k (m.t) + k (kf.t), a_hat[i] = t.a_f + c[i]
        double sK = 0.0, sM = 0.0, sKF = 0.0, sA = 0.0;
        for (int k = lane; k < m; k += 64) {
          const double tv = Tb[(long)i * m + k];
          sK = fma(tv, vK[k], sK);
          sM = fma(tv, vM[k], sM);
          sKF = fma(tv, vKF[k], sKF);
          sA = fma(tv, vAF[k], sA);
        }
        sK = bsp_wave_sum(sK);
        sM = bsp_wave_sum(sM);
        sKF = bsp_wave_sum(sKF);
        sA = bsp_wave_sum(sA);
        const double* ptv = x.pt_in + (b * BSP_ND + dsel) * m;
        for (int l = lane; l < m; l += 64) {
          double sv = ptv[l];
          sv -= sK * vM[l];
          sv -= sM * vK[l];
          sv += sKF * vK[l];
          w[(long)r * m + l] = sv;
        }
        if (lane == 0) {
          const double ah = sA + (A.c ? A.c[b * m + i] : 0.0);
          x.a_ws[b * m + i] = ah;
          if (A.pred_a) A.pred_a[(b * (long)(n + 1) + t + 1) * m + i] = ah;
        }
      } else {
        if (rpw[q] == rpw[q + 1])
          for (int l = lane; l < m; l += 64) w[(long)r * m + l] = 0.0;
        double sA = 0.0;
        for (int e = rpw[q] + lane; e < rpw[q + 1]; e += 64) sA = fma(val[e], vAF[col[e]], sA);
        sA = bsp_wave_sum(sA);
        if (lane == 0) {
          const double ah = sA + (A.c ? A.c[b * m + i] : 0.0);
          x.a_ws[b * m + i] = ah;
          if (A.pred_a) A.pred_a[(b * (long)(n + 1) + t + 1) * m + i] = ah;
        }
      }
    }
    // ---- dot products of the wave's own rows (LDS written by this wave only: no barrier needed before them):
    //      (S z)[i], column d of S, (S t_d)[i]
    double yn = 0.0;
    if (t + 1 < n) yn = A.y[((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L) + t + 1];
    for (int q = 0; q < RW; ++q) {
      const int r = q0 + q, i = i0 + r;
      if (i >= m) break;
      double a0 = 0.0, ac[BSP_ND], at[BSP_ND];
#pragma unroll
      for (int d = 0; d < BSP_ND; ++d) ac[d] = at[d] = 0.0;
#pragma unroll
      for (int g = 0; g < BSP_CG; ++g) {   // the first 64 BSP_CG columns: dense row 0 from the registers
        const int l = lane + 64 * g;
        if (l < m) {
          const double wv = w[(long)r * m + l];
          a0 = fma(wv, vTZ[l], a0);
          ac[0] = fma(wv, tj0[g], ac[0]);
          at[0] = fma(wv, tt0[g], at[0]);
#pragma unroll
          for (int d = 1; d < BSP_ND; ++d)
            if (d < nd) {
              ac[d] = fma(wv, Tb[(long)dr[d] * m + l], ac[d]);
              at[d] = fma(wv, x.ttd[(b * BSP_ND + d) * m + l], at[d]);
            }
        }
      }
      for (int l = lane + 64 * BSP_CG; l < m; l += 64) {
        const double wv = w[(long)r * m + l];
        a0 = fma(wv, vTZ[l], a0);
#pragma unroll
        for (int d = 0; d < BSP_ND; ++d)
          if (d < nd) {
            ac[d] = fma(wv, Tb[(long)dr[d] * m + l], ac[d]);
            at[d] = fma(wv, x.ttd[(b * BSP_ND + d) * m + l], at[d]);
          }
      }
      a0 = bsp_wave_sum(a0);
      const double ezq = bsp_readlane_f64(ez_mine, q);
      if (lane == 0) x.Mn[b * m + i] = (yn != yn) ? 0.0 : a0 + ezq;
#pragma unroll
      for (int d = 0; d < BSP_ND; ++d)
        if (d < nd) {
          const double c1 = bsp_wave_sum(ac[d]), c2 = bsp_wave_sum(at[d]);
          const double ev = d == 0 ? bsp_readlane_f64(etd0_mine, q) : x.etd[(b * BSP_ND + d) * m + i];
          if (lane == 0) {
            sd[d * R + r] = c1;
            x.pt_out[(b * BSP_ND + d) * m + i] = c2 + ev;
          }
        }
    }
  }
  __syncthreads();
  // ---- stage 2: a thread per column j:  S[i0 + r][j] = sum_{l in nz(j)} w_r[l] T[j][l] + E[i0 + r][j]
  double* Pob = x.P_out + b * mm;
  double* tape = A.pred_P ? A.pred_P + (b * (long)(n + 1) + t + 1) * mm : nullptr;
  const double* Eb = x.E + b * mm;
  for (int cb = 0; cb < m; cb += 256) {
    const int j = cb + tid;
    if (j >= m) break;
    int eb, ee, c0_, c1_;
    double v0_, v1_;
    if (cb == 0) {
      eb = s2_eb[0]; ee = s2_ee[0]; c0_ = s2_c0[0]; c1_ = s2_c1[0]; v0_ = s2_v0[0]; v1_ = s2_v1[0];
    } else if (cb == 256) {
      eb = s2_eb[1]; ee = s2_ee[1]; c0_ = s2_c0[1]; c1_ = s2_c1[1]; v0_ = s2_v0[1]; v1_ = s2_v1[1];
    } else {
      eb = rp[j];
      ee = rp[j + 1];
      c0_ = eb < ee ? col[eb] : 0;
      v0_ = eb < ee ? val[eb] : 0.0;
      c1_ = eb + 1 < ee ? col[eb + 1] : 0;
      v1_ = eb + 1 < ee ? val[eb + 1] : 0.0;
    }
    double s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.0;
    if (eb < ee) {
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] = fma(w[(long)r * m + c0_], v0_, s[r]);
    }
    if (eb + 1 < ee) {
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] = fma(w[(long)r * m + c1_], v1_, s[r]);
    }
    for (int e = eb + 2; e < ee; ++e) {
      const int l = col[e];
      const double tv = val[e];
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] = fma(w[(long)r * m + l], tv, s[r]);
    }
#pragma unroll
    for (int d = 0; d < BSP_ND; ++d)
      if (dr[d] == j) {
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] = sd[d * R + r];
      }
    if ((ebits >> (j / 32)) & 1u) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (i0 + r < m) s[r] += Eb[(long)(i0 + r) * m + j];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (i0 + r < m) {
        Pob[(long)(i0 + r) * m + j] = s[r];
        if (tape) __builtin_nontemporal_store(s[r], tape + (long)(i0 + r) * m + j);
      }
  }
}


// out = A X A' + u1 v1' + u2 v2' for a CSR A without dense rows (kf_smoother_bf_big.hip: M_{t-1} = T' M_t T + rank-one terms): the two
// stages of bsp_step_kernel without its per-step vectors -- stage 1  w_i = sum_k A[i,k] X[k,:] into LDS, stage 2  out[i,j] = sum_l w_i[l] A[j,l]
template <int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void bsp_sandwich_kernel(const BspCsr c, const double* __restrict__ X,
                                                                                                 double* __restrict__ out, int m, long B,
                                                                                                 const double* __restrict__ u1,
                                                                                                 const double* __restrict__ v1,
                                                                                                 const double* __restrict__ u2,
                                                                                                 const double* __restrict__ v2, int nblk,
                                                                                                 double* __restrict__ wout) {
  extern __shared__ double sm[];
  constexpr int RW = R / 4;
  constexpr int NE = 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long L = blockIdx.x;
  const long b = (L & 7) + 8 * (L / (8L * nblk));
  if (b >= B) return;
  const int i0 = (int)((L >> 3) % nblk) * R;
  const long mm = (long)m * m;
  double* w = sm;                  // [R][m]
  double* ru = w + (long)R * m;    // [2][R]  u1, u2 of the block's rows
  const double* Xb = X + b * mm;
  const int* rp = c.rowptr + b * (m + 1);
  const int* col = c.col + b * (long)c.cap;
  const double* val = c.val + b * (long)c.cap;
  const int q0 = wave * RW;
  const int rp_mine = rp[min(i0 + q0 + min(lane, RW), m)];
  int s2_eb[2] = {0, 0}, s2_ee[2] = {0, 0};
  double v1j[2] = {0.0, 0.0}, v2j[2] = {0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (tid + 256 * h < m) {
      s2_eb[h] = rp[tid + 256 * h];
      s2_ee[h] = rp[tid + 256 * h + 1];
      v1j[h] = u1 ? v1[b * m + tid + 256 * h] : 0.0;
      v2j[h] = u2 ? v2[b * m + tid + 256 * h] : 0.0;
    }
  if (tid < 2 * R) {
    const int r = tid % R, which = tid / R;
    const double* u = which ? u2 : u1;
    ru[tid] = (u && i0 + r < m) ? u[b * m + i0 + r] : 0.0;
  }
  int s2_c0[2] = {0, 0}, s2_c1[2] = {0, 0};
  double s2_v0[2] = {0.0, 0.0}, s2_v1[2] = {0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (s2_eb[h] < s2_ee[h]) {
      s2_c0[h] = col[s2_eb[h]];
      s2_v0[h] = val[s2_eb[h]];
    }
    if (s2_eb[h] + 1 < s2_ee[h]) {
      s2_c1[h] = col[s2_eb[h] + 1];
      s2_v1[h] = val[s2_eb[h] + 1];
    }
  }
  // ---- stage 1 (rows written by this wave only; the barrier below orders them before stage 2)
  {
    int rpw[RW + 1];
#pragma unroll
    for (int q = 0; q <= RW; ++q) rpw[q] = __builtin_amdgcn_readlane(rp_mine, q);
    const int e_beg = rpw[0], e_end = rpw[RW];
    int prev_row = -1;
    for (int chunk = e_beg; chunk < e_end; chunk += 64) {
      const int me = chunk + lane;
      const bool have = me < e_end;
      const int ck = have ? col[me] : 0;
      const double cv = have ? val[me] : 0.0;
      int cr = 0;
#pragma unroll
      for (int q = 1; q < RW; ++q) cr += (me >= rpw[q]) ? 1 : 0;
      const int nloc = min(64, e_end - chunk);
      for (int e = 0; e < nloc; e += NE) {
        int kq[NE], rq[NE];
        double tq[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
          const int src = min(e + u, nloc - 1);
          kq[u] = __builtin_amdgcn_readlane(ck, src);
          rq[u] = q0 + __builtin_amdgcn_readlane(cr, src);
          tq[u] = bsp_readlane_f64(cv, src);
        }
        for (int cg = 0; cg < m; cg += 64 * BSP_CG) {
          double pv[NE][BSP_CG];
#pragma unroll
          for (int u = 0; u < NE; ++u)
#pragma unroll
            for (int g = 0; g < BSP_CG; ++g) {
              const int l = cg + lane + 64 * g;
              pv[u][g] = (l < m && e + u < nloc) ? Xb[(long)kq[u] * m + l] : 0.0;
            }
#pragma unroll
          for (int u = 0; u < NE; ++u) {
            if (e + u < nloc) {
              const bool first = rq[u] != (u == 0 ? prev_row : rq[u - 1]);
#pragma unroll
              for (int g = 0; g < BSP_CG; ++g) {
                const int l = cg + lane + 64 * g;
                if (l < m) w[(long)rq[u] * m + l] = first ? tq[u] * pv[u][g] : fma(tq[u], pv[u][g], w[(long)rq[u] * m + l]);
              }
            }
          }
        }
        prev_row = rq[min(NE, nloc - e) - 1];
      }
    }
    for (int q = 0; q < RW; ++q)   // empty rows
      if (rpw[q] == rpw[q + 1])
        for (int l = lane; l < m; l += 64) w[(long)(q0 + q) * m + l] = 0.0;
    if (wout)   // the rows of A X themselves (kf_backward_big.hip: X1' = T' Pb is the left factor of another product)
      for (int q = 0; q < RW; ++q)
        if (i0 + q0 + q < m)
          for (int l = lane; l < m; l += 64) wout[b * mm + (long)(i0 + q0 + q) * m + l] = w[(long)(q0 + q) * m + l];
  }
  __syncthreads();
  // ---- stage 2
  double* ob = out + b * mm;
  for (int cb = 0; cb < m; cb += 256) {
    const int j = cb + tid;
    if (j >= m) break;
    int eb, ee, c0_, c1_;
    double v0_, v1_, a1, a2;
    if (cb == 0) {
      eb = s2_eb[0]; ee = s2_ee[0]; c0_ = s2_c0[0]; c1_ = s2_c1[0]; v0_ = s2_v0[0]; v1_ = s2_v1[0]; a1 = v1j[0]; a2 = v2j[0];
    } else if (cb == 256) {
      eb = s2_eb[1]; ee = s2_ee[1]; c0_ = s2_c0[1]; c1_ = s2_c1[1]; v0_ = s2_v0[1]; v1_ = s2_v1[1]; a1 = v1j[1]; a2 = v2j[1];
    } else {
      eb = rp[j];
      ee = rp[j + 1];
      c0_ = eb < ee ? col[eb] : 0;
      v0_ = eb < ee ? val[eb] : 0.0;
      c1_ = eb + 1 < ee ? col[eb + 1] : 0;
      v1_ = eb + 1 < ee ? val[eb + 1] : 0.0;
      a1 = u1 ? v1[b * m + j] : 0.0;
      a2 = u2 ? v2[b * m + j] : 0.0;
    }
    double s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.0;
    if (eb < ee) {
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] = fma(w[(long)r * m + c0_], v0_, s[r]);
    }
    if (eb + 1 < ee) {
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] = fma(w[(long)r * m + c1_], v1_, s[r]);
    }
    for (int e = eb + 2; e < ee; ++e) {
      const int l = col[e];
      const double tv = val[e];
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] = fma(w[(long)r * m + l], tv, s[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (i0 + r < m) ob[(long)(i0 + r) * m + j] = fma(ru[r], a1, fma(ru[R + r], a2, s[r]));
  }
}

}  // namespace

// The m x m x m product of this file for other callers (kf_smoother_big.hip): C = A op(B) (+ E) per draw, leading dimension m,
// per-draw strides in doubles.  upper_mirror (needs add_e): C is symmetric -- only the tiles that touch the upper triangle are
// computed and every element is written to both triangles.
int kf_big_gemm_launch(hipStream_t st, bool trans_b, bool add_e, bool upper_mirror, const double* A, long sA, const double* Bm,
                       long sB, const double* E, long sE, double* C, long sC, int m, long draws) {
  const int tgm = (m + GTM - 1) / GTM, tgn = (m + GTN - 1) / GTN;
  const unsigned ggrid = (unsigned)(8L * tgn * tgm * ((draws + 7) / 8));
  const BigGemmExtra x0 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0L, nullptr, 0L, nullptr, nullptr, nullptr};
  if (!trans_b && !add_e && !upper_mirror)
    hipLaunchKernelGGL((big_gemm_kernel<false, false>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x0);
  else if (trans_b && add_e && upper_mirror)
    hipLaunchKernelGGL((big_gemm_kernel<true, true, true, false, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x0);
  else if (!trans_b && add_e && upper_mirror)
    hipLaunchKernelGGL((big_gemm_kernel<false, true, true, false, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x0);
  else if (!trans_b && add_e && !upper_mirror)   // (E may be C itself: every element is read and written by the same thread)
    hipLaunchKernelGGL((big_gemm_kernel<false, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x0);
  else
    return KF_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_HIP;
}

// the same product with what kf_smoother_bf_big.hip / kf_smoother_big.hip need on top: the block map of a static factor (which
// operand it is: tmask_on_b), a sign, rank-one terms.  trans_b / add_e / upper_mirror as above.
int kf_big_gemm_launch_x(hipStream_t st, bool trans_b, bool add_e, const double* A, long sA, const double* Bm, long sB, const double* E,
                         long sE, double* C, long sC, int m, long draws, const uint32_t* tmask, int tmask_on_b, int neg_acc,
                         const double* u1, const double* v1, const double* u2, const double* v2, bool upper_mirror) {
  const int tgm = (m + GTM - 1) / GTM, tgn = (m + GTN - 1) / GTN;
  const unsigned ggrid = (unsigned)(8L * tgn * tgm * ((draws + 7) / 8));
  BigGemmExtra x = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0L, nullptr, 0L, nullptr, nullptr, nullptr};
  x.tmask = (m <= 1024) ? tmask : nullptr;
  x.tmask_rb = (m + 63) / 64;
  x.tmask_on_b = tmask_on_b;
  x.neg_acc = neg_acc;
  x.u1 = u1; x.v1 = v1; x.u2 = u2; x.v2 = v2;
  if (upper_mirror) {   // C symmetric (sign and rank-one terms included): the tiles that touch the upper triangle, written to both
    if (!trans_b && add_e)
      hipLaunchKernelGGL((big_gemm_kernel<false, true, true, false, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x);
    else if (trans_b && !add_e)
      hipLaunchKernelGGL((big_gemm_kernel<true, false, true, false, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x);
    else if (trans_b && add_e)
      hipLaunchKernelGGL((big_gemm_kernel<true, true, true, false, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x);
    else
      return KF_ERR_UNSUPPORTED;
  } else if (!trans_b && !add_e)
    hipLaunchKernelGGL((big_gemm_kernel<false, false>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x);
  else if (!trans_b && add_e)
    hipLaunchKernelGGL((big_gemm_kernel<false, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x);
  else if (trans_b && !add_e)
    hipLaunchKernelGGL((big_gemm_kernel<true, false>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x);
  else
    hipLaunchKernelGGL((big_gemm_kernel<true, true>), dim3(ggrid), dim3(GNT), 0, st, A, sA, Bm, sB, E, sE, C, sC, m, draws, tgn, tgm, x);
  return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_HIP;
}

// block map of a per-draw m x m matrix (64-row blocks x 32-column chunks, m <= 1024): tmask[B][ceil(m / 64)]
int kf_big_tmask_launch(hipStream_t st, const double* T, int m, long B, uint32_t* tmask) {
  hipLaunchKernelGGL(big_tmask_kernel, dim3((unsigned)((m + 63) / 64), (unsigned)B), dim3(256), 0, st, T, m, tmask);
  return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_HIP;
}

int kf_big_tmask_cols_launch(hipStream_t st, const double* T, int m, long B, uint32_t* tmask) {
  hipLaunchKernelGGL(big_tmask_cols_kernel, dim3((unsigned)((m + 63) / 64), (unsigned)B), dim3(256), 0, st, T, m, tmask);
  return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_HIP;
}

size_t kf_bsp_csr_bytes(long B, int m) {
  const size_t cap = (size_t)BSP_CAPF * m;
  return (size_t)B * (8 * cap + 4 * ((size_t)m + 1 + cap + BSP_ND + 1)) + 8 + 256;
}

int kf_bsp_csr_build(hipStream_t st, const double* Tm, int m, long B, void* mem, KfBspCsr* c) {
  if (m > BIG_PFIX_M) return KF_ERR_UNSUPPORTED;
  c->cap = BSP_CAPF * m;
  double* q = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(mem) + 63) & ~(uintptr_t)63);
  c->val = q;
  int* qi = reinterpret_cast<int*>(q + B * (size_t)c->cap);
  c->rowptr = qi;  qi += B * (size_t)(m + 1);
  c->col = qi;     qi += B * (size_t)c->cap;
  c->drow = qi;    qi += B * (size_t)BSP_ND;
  c->nd = qi;      qi += B;
  c->ok = qi;
  if (hipMemsetAsync(c->ok, 1, sizeof(int), st) != hipSuccess || hipMemsetAsync(c->ok + 1, 0, sizeof(int), st) != hipSuccess) return KF_ERR_HIP;
  hipLaunchKernelGGL(bsp_csr_kernel, dim3((unsigned)B), dim3(256), 0, st, Tm, m, *c);
  return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_HIP;
}

int kf_bsp_sandwich_launch(hipStream_t st, const KfBspCsr& c, const double* X, double* out, int m, long B, const double* u1,
                           const double* v1, const double* u2, const double* v2, double* wout) {
  auto lds_of = [&](int rows) { return sizeof(double) * ((size_t)rows * m + 2 * (size_t)rows); };
  const int R = lds_of(8) <= KF_LDS_BYTES_MAX / 2 ? 8 : 4;
  const size_t lds = lds_of(R);
  const void* fn = R == 8 ? reinterpret_cast<const void*>(bsp_sandwich_kernel<8>) : reinterpret_cast<const void*>(bsp_sandwich_kernel<4>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KF_ERR_HIP;
  const int nblk = (m + R - 1) / R;
  const unsigned grid = (unsigned)(8L * nblk * ((B + 7) / 8));
  if (R == 8)
    hipLaunchKernelGGL(bsp_sandwich_kernel<8>, dim3(grid), dim3(256), lds, st, c, X, out, m, B, u1, v1, u2, v2, nblk, wout);
  else
    hipLaunchKernelGGL(bsp_sandwich_kernel<4>, dim3(grid), dim3(256), lds, st, c, X, out, m, B, u1, v1, u2, v2, nblk, wout);
  return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_HIP;
}

size_t kf_forward_big_workspace_bytes(long B, int m, int p) {
  size_t mm = (size_t)m * m;
  const size_t tiles = (size_t)((m + GTM - 1) / GTM + (m + GTN - 1) / GTN);  // shares of m = P z per tile row / tile column
  const size_t tmask_words = 2 * (size_t)B * ((m + 63) / 64);   // big_tmask_kernel's block maps of T and of R Q R'
  return sizeof(double) * ((size_t)B * (4 * mm + 2 * m + 3 * (size_t)m * p + tiles * m) + 64) + 4 * tmask_words + 64;
}

int kf_launch_forward_big(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
  const size_t need = kf_forward_big_workspace_bytes(a.B, a.m, a.p);
  if (!workspace || workspace_bytes < need) {
    KF_SET_ERR(h, "forward(big): m=%d needs a workspace of %zu bytes (got %zu); see kf_workspace_bytes",
               a.m, need, workspace_bytes);
    return KF_ERR_WORKSPACE;
  }
  if (a.B > 65535) {
    KF_SET_ERR(h, "forward(big): B=%ld > 65535 draws per call", a.B);
    return KF_ERR_UNSUPPORTED;
  }
  const int m = a.m, p = a.p;
  const size_t mm = (size_t)m * m;
  double* ws = (double*)(((uintptr_t)workspace + 63) & ~(uintptr_t)63);
  double* P_ws = ws;
  double* Pf_ws = P_ws + a.B * mm;   // P_f, then S
  double* W_ws = Pf_ws + a.B * mm;
  double* RQR_ws = W_ws + a.B * mm;
  double* a_ws = RQR_ws + a.B * mm;
  double* M_ws = a_ws + a.B * m;
  double* K_ws = M_ws + a.B * (size_t)m * p;
  double* KF_ws = K_ws + a.B * (size_t)m * p;
  double* af_ws = KF_ws + a.B * (size_t)m * p;

  size_t lds_upd = sizeof(double) * ((size_t)4 * p * m + 2 * m + 3 * p * p + 4 * p + 8);
  if (lds_upd > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "forward(big): p=%d m=%d needs %zu B of LDS in the update kernel", p, m, lds_upd);
    return KF_ERR_UNSUPPORTED;
  }
  KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(big_update_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_upd));
  const int init_blocks = (int)((mm + 255) / 256 < 64 ? (mm + 255) / 256 : 64);
  hipLaunchKernelGGL(big_init_kernel, dim3(init_blocks, (unsigned)a.B), dim3(256), 0, stream, a, a_ws,
                     P_ws, RQR_ws);
  const int tgm = (m + GTM - 1) / GTM, tgn = (m + GTN - 1) / GTN, ts = (m + 31) / 32;
  const bool fused_update = a.filter == KF_FILTER_UNIVARIATE;
  const size_t lds_gain = sizeof(double) * ((size_t)3 * p * m + 2 * m + 3 * p * p + 4 * p + 8);
  const size_t lds_pf = sizeof(double) * ((size_t)2 * p * m);
  if (!fused_update) {
    if (lds_gain > KF_LDS_BYTES_MAX || lds_pf > KF_LDS_BYTES_MAX) {
      KF_SET_ERR(h, "forward(big): p=%d m=%d needs %zu B of LDS in the gain kernel", p, m, lds_gain);
      return KF_ERR_UNSUPPORTED;
    }
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(big_gain_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gain));
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(big_pf_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pf));
  }
  // KF_BIG_FUSE=0 in the environment: the separate a_hat / mirror kernels of round 2 (A/B measurements)
  const char* fe = kf_env_switch("KF_BIG_FUSE");
  const bool fuse = !(fe && fe[0] == '0');
  // P_f formed on the way into LDS instead of by its own pass: scalar observation, m <= 1024, the filtered covariance tape not requested
  const bool pfix = fuse && !fused_update && p == 1 && m <= BIG_PFIX_M && !a.filt_P;
  const BigGemmExtra x0 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0L, nullptr, 0L, nullptr, nullptr, nullptr};
  double* mrow_ws = af_ws + a.B * (size_t)m;             // [B][tgn][m]
  double* mcol_ws = mrow_ws + a.B * (size_t)tgn * m;     // [B][tgm][m]
  // block map of T (static, m <= 1024): K chunks of all-zero blocks are skipped in both products.  KF_BIG_TMASK=0 in the
  // environment: every chunk (A/B measurements; the results are bit for bit the same)
  uint32_t *tmask_ws = nullptr, *emask_ws = nullptr;
  const int tmask_rb = (m + 63) / 64;
  {
    const char* te = kf_env_switch("KF_BIG_TMASK");
    if (m <= 1024 && !(te && te[0] == '0')) {
      tmask_ws = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(mcol_ws + a.B * (size_t)tgm * m) + 63) & ~(uintptr_t)63);
      hipLaunchKernelGGL(big_tmask_kernel, dim3((unsigned)tmask_rb, (unsigned)a.B), dim3(256), 0, stream, a.Tm, m, tmask_ws);
      emask_ws = tmask_ws + a.B * (size_t)tmask_rb;
      hipLaunchKernelGGL(big_tmask_kernel, dim3((unsigned)tmask_rb, (unsigned)a.B), dim3(256), 0, stream, RQR_ws, m, emask_ws);
    }
  }
  // Sparse transition: T -> CSR once per call, then one HBM-bound gather kernel per step instead of the two products (see
  // bsp_step_kernel).  Whether every draw's T fits the CSR is known on the device only: ONE 4-byte read-back and stream
  // synchronisation per call (include/kf.h) -- skipped, with the MFMA path, while the stream is being captured into a graph
  // or with KF_BIG_SPARSE=0 in the environment.
  if (pfix) {
    const char* se = kf_env_switch("KF_BIG_SPARSE");
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap_status);
    const bool capturing = cap_status != hipStreamCaptureStatusNone;
    const bool cached_sparse = h->big_cache_sparse && h->big_cache_T == a.Tm && h->big_cache_B == a.B && h->big_cache_m == m;
    if (!(se && se[0] == '0') && (!capturing || cached_sparse)) {
      BspCsr csr;
      csr.cap = BSP_CAPF * m;
      // carved from W_ws (B m^2 doubles, unused on this path; m > 64)
      double* q = W_ws;
      csr.val = q;                                 q += a.B * (size_t)csr.cap;
      double* Mb[2] = {q, q + a.B * (size_t)m};    q += 2 * a.B * (size_t)m;
      double* ptb[2] = {q, q + a.B * (size_t)BSP_ND * m};  q += 2 * a.B * (size_t)BSP_ND * m;
      double* tz_ws = q;                           q += a.B * (size_t)m;
      double* ez_ws = q;                           q += a.B * (size_t)m;
      double* ttd_ws = q;                          q += a.B * (size_t)BSP_ND * m;
      double* etd_ws = q;                          q += a.B * (size_t)BSP_ND * m;
      int* qi = reinterpret_cast<int*>(q);
      csr.rowptr = qi;                             qi += a.B * (size_t)(m + 1);
      csr.col = qi;                                qi += a.B * (size_t)csr.cap;
      csr.drow = qi;                               qi += a.B * (size_t)BSP_ND;
      csr.nd = qi;                                 qi += a.B;
      csr.ok = qi;
      KF_HIP_CHECK(h, hipMemsetAsync(csr.ok, 1, sizeof(int), stream));   // (any non-zero value)
      KF_HIP_CHECK(h, hipMemsetAsync(csr.ok + 1, 0, sizeof(int), stream));   // (the largest number of dense rows: unused here)
      hipLaunchKernelGGL(bsp_csr_kernel, dim3((unsigned)a.B), dim3(256), 0, stream, a.Tm, m, csr);
      int fits = 0;
      if (capturing) {
        fits = 1;   // the verdict of the last un-captured call on this (T pointer, B, m); the replay checks it again on the device
      } else {
        KF_HIP_CHECK(h, hipMemcpyAsync(&fits, csr.ok, sizeof(int), hipMemcpyDeviceToHost, stream));
        KF_HIP_CHECK(h, hipStreamSynchronize(stream));
        h->big_cache_T = a.Tm; h->big_cache_B = a.B; h->big_cache_m = m; h->big_cache_sparse = fits != 0;
      }
      if (fits) {
        // rows per workgroup: 8 (measured at C5, 128 draws: 4 / 8 / 16 rows -> 98 / 92 / 112 ms), 4 when 8 would leave one workgroup per CU
        auto lds_of = [&](int rows) { return sizeof(double) * ((size_t)(rows + 5) * m + (size_t)BSP_ND * rows); };
        int R = lds_of(8) <= KF_LDS_BYTES_MAX / 2 ? 8 : 4;
        if (const char* re = kf_env_experiment("KF_BSP_ROWS")) R = atoi(re) == 16 ? 16 : (atoi(re) == 4 ? 4 : 8);
        const size_t lds_sp = lds_of(R);
        if (lds_sp <= KF_LDS_BYTES_MAX) {
          const void* fn = R == 16 ? reinterpret_cast<const void*>(bsp_step_kernel<16>)
                           : (R == 8 ? reinterpret_cast<const void*>(bsp_step_kernel<8>) : reinterpret_cast<const void*>(bsp_step_kernel<4>));
          KF_HIP_CHECK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp));
          const int nblk = (m + R - 1) / R;
          const unsigned sgrid = (unsigned)(8L * nblk * ((a.B + 7) / 8));
          hipLaunchKernelGGL(big_mvec_kernel, dim3((m + MV_ROWS - 1) / MV_ROWS, (unsigned)a.B), dim3(256), 0, stream, a, 0, P_ws, Mb[0]);
          hipLaunchKernelGGL(bsp_pt_kernel, dim3((m + 15) / 16, (unsigned)a.B), dim3(256), 0, stream, P_ws, a.Tm, m, csr, ptb[0]);
          hipLaunchKernelGGL(bsp_vec_kernel, dim3((m + 63) / 64, (unsigned)a.B), dim3(256), 0, stream, a.Tm, RQR_ws, a.Z, m, csr, tz_ws, ez_ws,
                             ttd_ws, etd_ws);
          double* Pb[2] = {P_ws, Pf_ws};
          for (int t = 0; t < a.T; ++t) {
            const int c0 = t & 1, c1 = c0 ^ 1;
            hipLaunchKernelGGL(big_gain_kernel, dim3((unsigned)a.B), dim3(256), lds_gain, stream, a, t, a_ws, Mb[c0], K_ws, KF_ws,
                               af_ws, (const double*)nullptr, (const double*)nullptr, tgn, tgm, Mb[c0]);
            const BspStep xs = {Pb[c0], Pb[c1], K_ws, Mb[c0], KF_ws, af_ws, ptb[c0], Mb[c1], ptb[c1], a_ws, RQR_ws, emask_ws, tmask_rb,
                                tz_ws, ez_ws, ttd_ws, etd_ws};
            if (R == 16)
              hipLaunchKernelGGL(bsp_step_kernel<16>, dim3(sgrid), dim3(256), lds_sp, stream, a, t, csr, xs, nblk);
            else if (R == 8)
              hipLaunchKernelGGL(bsp_step_kernel<8>, dim3(sgrid), dim3(256), lds_sp, stream, a, t, csr, xs, nblk);
            else
              hipLaunchKernelGGL(bsp_step_kernel<4>, dim3(sgrid), dim3(256), lds_sp, stream, a, t, csr, xs, nblk);
          }
          if (capturing)
            hipLaunchKernelGGL(bsp_guard_kernel, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, stream, csr.ok, a.logp, a.info, (long)a.B);
          KF_HIP_CHECK(h, hipGetLastError());
          h->last_forward_kernel = "bsp_step_kernel";
          return KF_OK;
        }
      }
    }
  }
  h->last_forward_kernel = "big_gemm_kernel";
  for (int t = 0; t < a.T; ++t) {
    if (fused_update) {
      hipLaunchKernelGGL(big_update_kernel, dim3((unsigned)a.B), dim3(256), lds_upd, stream, a, t, a_ws,
                         P_ws, Pf_ws);
    } else {
      const bool shares = pfix && t > 0;  // m = P z from the shares the previous step's second product left
      if (!shares)
        hipLaunchKernelGGL(big_mvec_kernel, dim3((m + MV_ROWS - 1) / MV_ROWS, (unsigned)a.B), dim3(256), 0,
                           stream, a, t, P_ws, M_ws);
      hipLaunchKernelGGL(big_gain_kernel, dim3((unsigned)a.B), dim3(256), lds_gain, stream, a, t, a_ws,
                         M_ws, K_ws, KF_ws, af_ws, shares ? mrow_ws : nullptr, shares ? mcol_ws : nullptr, tgn, tgm, M_ws);
      if (!pfix) {
        hipLaunchKernelGGL(big_apred_kernel, dim3((m + MV_ROWS - 1) / MV_ROWS, (unsigned)a.B), dim3(256), 0,
                           stream, a, t, af_ws, a_ws);
        hipLaunchKernelGGL(big_pf_kernel, dim3((m + PF_ROWS - 1) / PF_ROWS, (unsigned)a.B), dim3(256), lds_pf,
                           stream, a, t, P_ws, M_ws, K_ws, KF_ws, Pf_ws);
      }
    }
    // 1-D grid: 8 * tiles * ceil(B / 8) workgroups (see the XCD mapping in the kernel)
    const unsigned ggrid = (unsigned)(8L * tgn * tgm * ((a.B + 7) / 8));
    if (pfix) {  // W = T (P - k m' - m k' + (k f) k'), and a_hat = T a_f + c by the workgroups ahead of the tiles
      BigGemmExtra x1 = {K_ws, M_ws, KF_ws, af_ws, a.c, a_ws, a.pred_a ? a.pred_a + (size_t)(t + 1) * m : nullptr,
                         (long)(a.T + 1) * m, nullptr, 0L, nullptr, nullptr, nullptr};
      x1.tmask = tmask_ws; x1.tmask_rb = tmask_rb; x1.tmask_on_b = 0;
      const unsigned xgrid = (unsigned)(8L * ((m + BIG_AH_ROWS - 1) / BIG_AH_ROWS) * ((a.B + 7) / 8));
      hipLaunchKernelGGL((big_gemm_kernel<false, false, false, true, false>), dim3(ggrid + xgrid), dim3(GNT), 0, stream,
                         a.Tm, (long)mm, P_ws, (long)mm, nullptr, 0L, W_ws, (long)mm, m, a.B, tgn, tgm, x1);
    } else {
      BigGemmExtra xa = x0;
      xa.tmask = tmask_ws; xa.tmask_rb = tmask_rb; xa.tmask_on_b = 0;
      hipLaunchKernelGGL((big_gemm_kernel<false, false>), dim3(ggrid), dim3(GNT), 0, stream,
                         a.Tm, (long)mm, Pf_ws, (long)mm, nullptr, 0L, W_ws, (long)mm, m, a.B, tgn, tgm, xa);
    }
    BigGemmExtra xb = x0;   // S = W T' + R Q R': T is the transposed B operand
    xb.tmask = tmask_ws; xb.tmask_rb = tmask_rb; xb.tmask_on_b = 1; xb.emask = emask_ws;
    if (fuse) {
      BigGemmExtra x2 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0L,
                         a.pred_P ? a.pred_P + (size_t)(t + 1) * mm : nullptr, (long)(a.T + 1) * (long)mm,
                         pfix ? a.Z : nullptr, pfix ? mrow_ws : nullptr, pfix ? mcol_ws : nullptr};
      x2.tmask = tmask_ws; x2.tmask_rb = tmask_rb; x2.tmask_on_b = 1; x2.emask = emask_ws;
      hipLaunchKernelGGL((big_gemm_kernel<true, true, true, false, true>), dim3(ggrid), dim3(GNT), 0, stream,
                         W_ws, (long)mm, a.Tm, (long)mm, RQR_ws, (long)mm, P_ws, (long)mm, m, a.B, tgn, tgm, x2);
    } else {
      hipLaunchKernelGGL((big_gemm_kernel<true, true, true>), dim3(ggrid), dim3(GNT), 0, stream,
                         W_ws, (long)mm, a.Tm, (long)mm, RQR_ws, (long)mm, Pf_ws, (long)mm, m, a.B, tgn, tgm, xb);
      hipLaunchKernelGGL(big_sym_kernel, dim3(ts, ts, (unsigned)a.B), dim3(256), 0, stream, Pf_ws, P_ws,
                         a.pred_P ? a.pred_P + (size_t)(t + 1) * mm : nullptr, (long)(a.T + 1) * (long)mm, m);
    }
  }
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
