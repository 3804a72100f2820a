// kf_forward_b4_steady.hip -- the steady half of the SPLIT forward pipeline of kf_forward_b4.hip (p == 1, m <= 10, static
// matrices, steady-state phase allowed): compaction between chunks, and the time-parallel steady-state kernels.
//
// Why.  For a static model with no missing observation ahead, the covariance half of the filter (reference
// kalman_filter.py:215-223, 255-284: P_hat, F, K, P_filt) is a Riccati iteration that does not see the data.  It converges
// geometrically; afterwards both covariance tapes are constants and the mean obeys a_{t+1} = A a_t + g y~_t + c with the
// constant closed-loop matrix A = T - T K Z, g = T K.  Round 2 ran that phase inside the recursion kernel, step by step on
// the wavefront that owned the draw -- and gained little: a launch lasts as long as its slowest SIMD, the four wavefronts of
// a SIMD are fixed at dispatch, and the draws converge anywhere between step 100 and never (profiles/r02/steady_state.txt).
// Here the work is re-dealt at KERNEL BOUNDARIES, where rebalancing is free:
//   * the recursion runs in chunks of CH steps (kf_fwd_b4_kernel<.., SS = 2>), one launch per chunk over a COMPACTED list
//     of the draws that have not converged yet (kf_b4_split_compact_kernel: order-preserving, one workgroup, no atomics);
//     a wavefront leaves as soon as its draw has converged and parks the tile in the hand-over buffer;
//   * the draws that converged during a chunk go to the steady kernels, launched on a second stream so that they fill the
//     SIMD slots and the HBM bandwidth the shrinking recursion launches leave idle:
//       kf_b4_steady_prep_kernel   per draw: F, K, P_filt, A, g and the powers A^(2^k), k = 0..5
//       kf_b4_steady_fill_kernel   both covariance tapes of all remaining steps: element i = image[i mod m^2], written as
//                                  whole aligned lines, 16 bytes per lane -- a pure store stream
//       kf_b4_steady_scan_kernel   the mean recursion, 64 steps at a time, ONE STEP PER LANE: the affine recursion is a
//                                  prefix scan, x_j = sum_{i <= j} A^(j-i) w_i (Hillis-Steele, six levels, each a matrix-
//                                  vector product with A^(2^k) whose entries come from scalar loads), then every lane
//                                  finishes its own step (v, a_filt, ll) and the block's rows leave through LDS as whole lines.
// Per draw-step the steady phase costs ~15 instructions instead of ~200, and it is balanced by construction.
// Results: the same function of the inputs as the step-by-step recursion up to the convergence tolerance (entrywise 1e-15
// on two consecutive checks, kf_forward_b4.hip) and the summation order of the scan (~1e-15 relative); not bit-identical.
#include <stdlib.h>

#include "kf_b4.h"
#include "kf_b4_split.h"

namespace {

typedef const double __attribute__((address_space(4)))* kf_cptr_t;  // constant address space: uniform loads become s_load
__device__ __forceinline__ kf_cptr_t as_const(const double* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (kf_cptr_t)(p);
#pragma clang diagnostic pop
}

// ---- compaction: hs_ts[b] == -1 -> recursion list of the next chunk; >= 0 -> appended to the steady list (hs_ts := -3) -------
__global__ __launch_bounds__(1024) void kf_b4_split_compact_kernel(const B4SplitWs W, int B, int chunk) {
  __shared__ int wsR[16], wsS[16];
  __shared__ int baseR, baseS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    baseR = 0;
    baseS = chunk ? W.offS[chunk] : 0;
    if (chunk == 0) W.offS[0] = 0;
  }
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + tid;
    const int st = b < B ? W.hs_ts[b] : -2;
    const bool isR = st == -1, isS = st >= 0;
    const unsigned long long mR = __ballot(isR), mS = __ballot(isS);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int pR = __popcll(mR & below), pS = __popcll(mS & below);
    if (lane == 0) {
      wsR[wv] = __popcll(mR);
      wsS[wv] = __popcll(mS);
    }
    __syncthreads();
    int oR = baseR, oS = baseS;
    for (int w = 0; w < wv; ++w) {
      oR += wsR[w];
      oS += wsS[w];
    }
    if (isR) W.listR[oR + pR] = b;
    if (isS) {
      W.listS[oS + pS] = b;
      W.listS_ts[oS + pS] = st;
      W.hs_ts[b] = -3;
    }
    __syncthreads();
    if (tid == 0) {
      int r = 0, s = 0;
      for (int w = 0; w < 16; ++w) {
        r += wsR[w];
        s += wsS[w];
      }
      baseR += r;
      baseS += s;
    }
    __syncthreads();
  }
  if (tid == 0) {
    W.cnt[0] = baseR;
    W.offS[chunk + 1] = baseS;
  }
}

// ---- per-draw constants of the steady phase ------------------------------------------------------------------------------
// hs_tile[b] is the register image of the predicted tile = row-major [N4][16] (register I, lane 16 g + c holds X[4I+g][c]):
// rows / columns 0..m-1 the covariance, JA = N4-2 the row (P Z')' with Z P Z' on the diagonal, JC = N4-1 twice the mean.
__global__ __launch_bounds__(KF_WAVE) void kf_b4_steady_prep_kernel(const KfFwdArgs A, const B4SplitWs W, int chunk) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, m = A.m, mm = m * m;
  const int N4 = 4 * ((m + 2 + 3) / 4), JA = N4 - 2, JC = N4 - 1;
  const int lo = W.offS[chunk], hi = W.offS[chunk + 1];
  double* sA = smem;        // A^(2^k), current
  double* sB = smem + mm;   // next
  double* sK = sB + mm;     // K | g
  for (int e = lo + blockIdx.x; e < hi; e += gridDim.x) {
    const long b = W.listS[e];
    const double* ht = W.hs_tile + b * (long)B4_HS_STRIDE;
    double* sp = W.sp + b * (long)W.sps;
    const double* gT = A.Tm + b * (long)mm;
    const double* gZ = A.Z + b * (long)m;
    const double F = A.H[b] + ht[JA * 16 + JA];
    const double rs = fast_rsqrt(fmax(F, 1e-300));
    if (lane == 0) {
      sp[0] = F;
      sp[1] = rs;
      sp[2] = 0.0;
      sp[3] = 0.0;
    }
    if (lane < m) {
      const double uh = ht[JA * 16 + lane] * rs;
      sK[lane] = uh * rs;  // K = P Z' / F
    }
    wave_lds_fence();
    if (lane < m) {
      double gv = 0.0;
      for (int l = 0; l < m; ++l) gv = fma(gT[lane * m + l], sK[l], gv);
      sK[m + lane] = gv;   // g = T K
      sp[4 + lane] = sK[lane];
      sp[4 + m + lane] = gv;
      sp[4 + 2 * m + lane] = gZ[lane];
      sp[4 + 3 * m + lane] = A.c ? A.c[b * (long)m + lane] : 0.0;
      sp[4 + 4 * m + lane] = 0.5 * ht[JC * 16 + lane];  // a at the hand-over step
    }
    wave_lds_fence();
    double* spPf = sp + 4 + 5 * m;
    double* spPp = spPf + mm;
    double* spA = spPp + mm;
    for (int x = lane; x < mm; x += KF_WAVE) {
      const int i = x / m, j = x - i * m;
      const double pij = ht[i * 16 + j];
      const double ui = ht[JA * 16 + i] * rs, uj = ht[JA * 16 + j] * rs;
      spPp[x] = pij;
      spPf[x] = fma(-ui, uj, pij);                  // the rank-one update of kf_forward_b4.hip: P - (u rs)(u rs)'
      const double aij = fma(-sK[m + i], gZ[j], gT[x]);  // A = T - g Z
      sA[x] = aij;
      spA[x] = aij;
    }
    wave_lds_fence();
    for (int k = 1; k < 6; ++k) {
      for (int x = lane; x < mm; x += KF_WAVE) {
        const int i = x / m, j = x - i * m;
        double v = 0.0;
        for (int l = 0; l < m; ++l) v = fma(sA[i * m + l], sA[l * m + j], v);
        sB[x] = v;
        spA[k * mm + x] = v;
      }
      wave_lds_fence();
      double* t = sA;
      sA = sB;
      sB = t;
    }
    wave_lds_fence();
  }
}

// ---- constant covariance tapes -----------------------------------------------------------------------------------------
// One tape range [dst, dst + count) of a periodic image: element i = img[(phase + i) mod mm]; img has mm + 1 entries
// (img[mm] = img[0]).  Single elements up to the first 128-byte line boundary, then 16 bytes per lane, whole lines.
// Windows [w_lo, w_hi) of 128 elements (counted from the first line boundary) are this caller's share; the head belongs
// to the caller with w_lo == 0.
__device__ __forceinline__ void fill_periodic(double* dst, long count, const double* img, int mm, int phase, int lane,
                                              long w_lo, long w_hi) {
  const long a0 = (long)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15ull);
  long lead = (16 - a0) & 15;
  if (lead > count) lead = count;
  if (w_lo == 0 && lane < lead) dst[lane] = img[(phase + lane) % mm];
  const int step = 128 % mm;
  long pos = lead + 128 * w_lo + 2 * lane;
  int off = (int)((phase + pos) % mm);
  const long end = (lead + 128 * w_hi < count) ? lead + 128 * w_hi : count;
  for (; pos < end; pos += 128) {
    if (pos + 1 < end) {
      double2_t v = {img[off], img[off + 1]};
      *reinterpret_cast<double2_t*>(dst + pos) = v;
    } else {
      dst[pos] = img[off];
    }
    off += step;
    if (off >= mm) off -= mm;
  }
}

__global__ __launch_bounds__(KF_WAVE) void kf_b4_steady_fill_kernel(const KfFwdArgs A, const B4SplitWs W, int chunk, int nslice) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, m = A.m, mm = m * m, n = A.T;
  const int lo = W.offS[chunk], hi = W.offS[chunk + 1];
  const long items = (long)(hi - lo) * nslice;
  double* imgF = smem;
  double* imgP = smem + mm + 2;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int e = lo + (int)(it / nslice), sl = (int)(it - (long)(e - lo) * nslice);
    const long b = W.listS[e];
    const int ts = W.listS_ts[e];
    const double* sp = W.sp + b * (long)W.sps + 4 + 5 * m;
    wave_lds_fence();
    for (int x = lane; x <= mm; x += KF_WAVE) {
      const int xx = x == mm ? 0 : x;
      imgF[x] = sp[xx];
      imgP[x] = sp[mm + xx];
    }
    wave_lds_fence();
    const long cnt = (long)(n - ts) * mm;
    if (A.filt_P) {
      double* dst = A.filt_P + (b * (long)n + ts) * mm;
      const long a0 = (long)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15ull);
      const long lead = ((16 - a0) & 15) < cnt ? ((16 - a0) & 15) : cnt;
      const long nw = (cnt - lead + 127) / 128;
      fill_periodic(dst, cnt, imgF, mm, 0, lane, nw * sl / nslice, nw * (sl + 1) / nslice);
    }
    if (A.pred_P) {
      double* dst = A.pred_P + (b * (long)(n + 1) + ts + 1) * mm;
      const long a0 = (long)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15ull);
      const long lead = ((16 - a0) & 15) < cnt ? ((16 - a0) & 15) : cnt;
      const long nw = (cnt - lead + 127) / 128;
      fill_periodic(dst, cnt, imgP, mm, 0, lane, nw * sl / nslice, nw * (sl + 1) / nslice);
    }
  }
}

// ---- the mean recursion as a prefix scan ----------------------------------------------------------------------------------
// rows of one 64-step block, [lane][M] in LDS -> the draw's tape, 16 bytes per lane where the address allows
__device__ __forceinline__ void store_rows(double* dst, const double* rows, int count, int lane) {
  int lead = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & 1ull);
  if (lead > count) lead = count;
  if (lead && lane == 0) dst[0] = rows[0];
  for (int pos = lead + 2 * lane; pos < count; pos += 128) {
    if (pos + 1 < count) {
      double2_t v = {rows[pos], rows[pos + 1]};
      *reinterpret_cast<double2_t*>(dst + pos) = v;
    } else {
      dst[pos] = rows[pos];
    }
  }
}

template <int M>
__global__ __launch_bounds__(KF_WAVE) void kf_b4_steady_scan_kernel(const KfFwdArgs A, const B4SplitWs W, int chunk) {
  constexpr int MM = M * M;
  __shared__ __attribute__((aligned(16))) double rowsP[KF_WAVE * M];
  __shared__ __attribute__((aligned(16))) double rowsF[KF_WAVE * M];
  const int lane = threadIdx.x, n = A.T;
  const int lo = W.offS[chunk], hi = W.offS[chunk + 1];
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  for (int e = lo + blockIdx.x; e < hi; e += gridDim.x) {
    const long b = W.listS[e];
    const int ts = W.listS_ts[e];
    const kf_cptr_t sp = as_const(W.sp + b * (long)W.sps);
    const double F = sp[0], rs = sp[1];
    const kf_cptr_t cK = sp + 4, cg = sp + 4 + M, cZ = sp + 4 + 2 * M, cc = sp + 4 + 3 * M, ca = sp + 4 + 4 * M;
    const kf_cptr_t cA = sp + 4 + 5 * M + 2 * MM;
    const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
    const double d1 = A.d ? A.d[b] : 0.0;
    const bool skipped = univariate && F == 0.0;  // :468-471
    const bool notpd = !skipped && !(F > 0.0);
    const double llconst = -0.5 * (KF_LOG_2PI + log(F));
    double acur[M];
#pragma unroll
    for (int i = 0; i < M; ++i) acur[i] = ca[i];
    double acc = 0.0;
    for (int tb = ts; tb < n; tb += KF_WAVE) {
      const int t = tb + lane;
      const bool act = t < n;
      const int cnt = (n - tb) < KF_WAVE ? (n - tb) : KF_WAVE;
      const double yt = act ? ybase[t] + dsign * d1 : 0.0;
      // w_j = g y~_j + c; lane 0 also carries the state the block starts from: w_0 += A a
      double x[M], xs[M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double aa = 0.0;
#pragma unroll
        for (int l = 0; l < M; ++l) aa = fma(cA[i * M + l], acur[l], aa);
        x[i] = fma(cg[i], yt, cc[i]) + (lane == 0 ? aa : 0.0);
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
          double v = x[i];
#pragma unroll
          for (int l = 0; l < M; ++l) v = fma(cA[k * MM + i * M + l], xs[l], v);
          x[i] = v;
        }
      }
      // x = a_{t+1}; the state this step starts from comes from the lane below
      double za = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const double up = __shfl_up(x[i], 1);
        xs[i] = lane == 0 ? acur[i] : up;
        za = fma(cZ[i], xs[i], za);
      }
      const double v = yt - za;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        rowsP[lane * M + i] = x[i];
        rowsF[lane * M + i] = fma(cK[i], v, xs[i]);
      }
      const double q = v * rs;
      const double quad = q * q;
      double llv;
      if (skipped) llv = exact_const ? -0.5 * KF_LOG_2PI : 0.0;
      else if (univariate && !exact_const && F == 1.0 && quad == 0.0) llv = 0.0;  // :497-500
      else llv = llconst - 0.5 * quad;
      if (act) {
        acc += llv;
        if (A.ll_obs) A.ll_obs[b * (long)n + t] = llv;
      }
      wave_lds_fence();
      if (A.pred_a) store_rows(A.pred_a + (b * (long)(n + 1) + tb + 1) * M, rowsP, cnt * M, lane);
      if (A.filt_a) store_rows(A.filt_a + (b * (long)n + tb) * M, rowsF, cnt * M, lane);
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < M; ++i) acur[i] = __shfl(x[i], KF_WAVE - 1);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
      A.logp[b] = W.hs_logp[b] + acc;
      int info = W.hs_info[b];
      if (notpd && info == 0) info = ((ts + 1) << 8) | 1;
      if (A.info) A.info[b] = info;
    }
  }
}

}  // namespace

size_t kf_b4_split_workspace_bytes(long B, int m, int n) {
  B4SplitWs W;
  return kf_b4_split_carve(nullptr, B, m, n, &W);
}

int kf_b4_split_compact(kf_handle* h, const B4SplitWs& W, long B, int chunk, hipStream_t stream) {
  hipLaunchKernelGGL(kf_b4_split_compact_kernel, dim3(1), dim3(1024), 0, stream, W, (int)B, chunk);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

int kf_b4_split_steady(kf_handle* h, const KfFwdArgs& a, const B4SplitWs& W, int chunk, hipStream_t stream) {
  const int m = a.m, mm = m * m;
  const unsigned grid = (unsigned)(a.B < 1024 ? a.B : 1024);
  hipLaunchKernelGGL(kf_b4_steady_prep_kernel, dim3(grid), dim3(KF_WAVE), sizeof(double) * (size_t)(2 * mm + 2 * m + 2), stream,
                     a, W, chunk);
  if (a.filt_P || a.pred_P) {
    // slices of ~64 KiB per tape: enough workgroups to fill the slots the recursion launch leaves free
    long ns = ((long)a.T * mm * 8 + 65535) / 65536;
    const int nslice = (int)(ns < 1 ? 1 : ns > 32 ? 32 : ns);
    hipLaunchKernelGGL(kf_b4_steady_fill_kernel, dim3(2048), dim3(KF_WAVE), sizeof(double) * (size_t)(2 * mm + 4), stream, a, W,
                       chunk, nslice);
  }
#define KF_SCAN(M_) \
  case M_: hipLaunchKernelGGL((kf_b4_steady_scan_kernel<M_>), dim3(grid), dim3(KF_WAVE), 0, stream, a, W, chunk); break;
  switch (m) {
    KF_SCAN(1) KF_SCAN(2) KF_SCAN(3) KF_SCAN(4) KF_SCAN(5) KF_SCAN(6) KF_SCAN(7) KF_SCAN(8) KF_SCAN(9) KF_SCAN(10)
    default:
      KF_SET_ERR(h, "forward(b4 split): no steady kernel for m = %d", m);
      return KF_ERR_UNSUPPORTED;
  }
#undef KF_SCAN
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
