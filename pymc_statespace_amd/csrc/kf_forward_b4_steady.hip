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
//       kf_b4_steady_fill_kernel   both covariance tapes of all remaining steps: element i = image[i mod m^2], written as
//                                  whole aligned lines, 16 bytes per lane -- a pure store stream
//       kf_b4_steady_scan_kernel   the mean recursion, 63 steps at a time, ONE STEP PER LANE: the affine recursion is a
//                                  prefix scan, x_j = sum_{i <= j} A^(j-i) w_i (Hillis-Steele, six levels, each a matrix-
//                                  vector product with A^(2^k), broadcast from LDS), then every lane finishes its own
//                                  step (v, a_filt, ll) and the block's rows leave through LDS as whole lines.
// Per draw-step the steady phase costs ~20 wavefront instructions instead of ~200, and it is balanced by construction.
// Results: the same function of the inputs as the step-by-step recursion up to the convergence tolerance (entrywise 1e-15
// on two consecutive checks, kf_forward_b4.hip) and rounding (~1e-15 relative); not bit-identical.
#include <stdlib.h>

#include "kf_b4.h"
#include "kf_b4_split.h"

namespace {

// ---- compaction: hs_ts[b] == -1 -> recursion list of the next chunk; >= 0 -> appended to the steady list (hs_ts := -3) -------
__global__ __launch_bounds__(1024) void kf_b4_split_compact_kernel(const B4SplitWs W, int B, int chunk, int t_next, int pair_ok) {
  __shared__ int sw;
  __shared__ int wsR[16], wsS[16];
  __shared__ int baseR, baseS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    baseR = 0;
    baseS = chunk ? W.offS[chunk] : 0;
    if (chunk == 0) W.offS[0] = 0;
  }
  __syncthreads();
  // Round 4: the status words of 8192 draws are loaded FIRST, eight independent loads per thread, then compacted out of registers.
  // The kernel runs beside the constant-tape fill, which keeps the HBM write queues full: with one load -> barrier -> store ->
  // barrier round per 1024 draws every round paid a memory round trip of tens of microseconds (profiles/r03/c3_split_timeline.txt:
  // 12.5 us alone, 737 and 840 us beside the fill -- and the next recursion launch waits behind it).
  constexpr int CU_ = 8;
  for (int b00 = 0; b00 < B; b00 += 1024 * CU_) {
    int stv[CU_];
#pragma unroll
    for (int u = 0; u < CU_; ++u) {
      const int b = b00 + u * 1024 + tid;
      stv[u] = b < B ? W.hs_ts[b] : -2;
    }
#pragma unroll
    for (int u = 0; u < CU_; ++u) {
      const int b0 = b00 + u * 1024;
      if (b0 >= B) break;   // (uniform)
      const int b = b0 + tid;
      const int st = stv[u];
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
  }
  // Once at most B4_PAIR_MAX draws are left in the full recursion a SIMD holds one of them at most, and compaction has nothing
  // more to give: they go to the pair kernel (kf_forward_b4p.hip: recursion + tape-writer wavefront), which runs them to the end
  // of the series; the fused chunks that follow find an empty list
  if (tid == 0) {
    const bool go = pair_ok && chunk > 0 && W.pair_ctl[2] == 0 && baseR > 0 && baseR <= B4_PAIR_MAX;
    sw = go ? 1 : 0;
    W.fillq[chunk] = 0;
    if (chunk == 0) {
      W.pair_ctl[0] = 0;
      W.pair_ctl[1] = 0;
      W.pair_ctl[2] = 0;
    }
    if (go) {
      W.pair_ctl[0] = baseR;
      W.pair_ctl[1] = t_next;
      W.pair_ctl[2] = chunk + 1;
    }
    W.cnt[0] = go ? 0 : baseR;
    W.offS[chunk + 1] = baseS;
  }
  __syncthreads();
  if (sw)
    for (int i = tid; i < baseR; i += 1024) W.hs_ts[W.listR[i]] = -4;
}

// ---- constant covariance tapes: fill_periodic (kf_b4_split.h) ---------------------------------------------------------------
// hs_tile[b] is the register image of the predicted tile = row-major [N4][16] (register I, lane 16 g + c holds X[4I+g][c]):
// rows / columns 0..m-1 the covariance, JA = N4-2 the row (P Z')' with Z P Z' on the diagonal, JC = N4-1 twice the mean.
// Work items = (hand-over entry, slice of its remaining steps); every chunk has its own queue counter fillq[c] over the entries
// it handed over, and a launch drains the queues of the chunks [0, chunk].  Two kinds of launches share the queues:
//   * THROTTLED ones behind every compaction, a few workgroups each: while recursion launches are running the fill must
//     not saturate the HBM -- the recursion's own tape stores queue behind the fill's and its dependency chain stretches
//     (measured: recursion chunks 250 / 204 / 166 us alone, 290 / 329 / 283 us beside an unthrottled fill);
//   * one full-width launch after the last recursion launch, which takes whatever is left at the full HBM rate.
__global__ __launch_bounds__(KF_WAVE) void kf_b4_steady_fill_kernel(const KfFwdArgs A, const B4SplitWs W, int chunk, int nslice) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, m = A.m, mm = m * m, n = A.T;
  const int N4 = 4 * ((m + 2 + 3) / 4), JA = N4 - 2;
  double* imgF = smem;
  double* imgP = smem + mm + 2;
  for (int c = 0; c <= chunk; ++c) {
    const int lo = W.offS[c], hi = W.offS[c + 1];
    const int items = (hi - lo) * nslice;
    for (;;) {
      int it = 0;
      if (lane == 0) it = atomicAdd(W.fillq + c, 1);
      it = __builtin_amdgcn_readfirstlane(it);
      if (it >= items) break;
      const int e = lo + it / nslice, sl = it % nslice;
      const long b = W.listS[e];
      const int ts = W.listS_ts[e];
      const double* ht = W.hs_tile + b * (long)B4_HS_STRIDE;
      const double F = A.H[b] + ht[JA * 16 + JA];
      const double rs = fast_rsqrt(fmax(F, 1e-300));
      wave_lds_fence();
      for (int x = lane; x <= mm; x += KF_WAVE) {
        const int xx = x == mm ? 0 : x;
        const int i = xx / m, j = xx - i * m;
        const double pij = ht[i * 16 + j];
        const double ui = ht[JA * 16 + i] * rs, uj = ht[JA * 16 + j] * rs;
        imgP[x] = pij;
        imgF[x] = fma(-ui, uj, pij);  // the rank-one update of kf_forward_b4.hip: P - (u rs)(u rs)'
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
}

// ---- the mean recursion as a prefix scan over time ---------------------------------------------------------------------------
// rows of one block, [row][M] in LDS -> the draw's tape, 16 bytes per lane where the address allows
__device__ __forceinline__ void store_rows(double* dst, const double* rows, int count, int lane) {
  int lead = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & 1ull);
  if (lead > count) lead = count;
  if (lead && lane == 0) dst[0] = rows[0];
  for (int pos = lead + 2 * lane; pos < count; pos += 2 * KF_WAVE) {
    if (pos + 1 < count) {
      double2_t v = {rows[pos], rows[pos + 1]};
      *reinterpret_cast<double2_t*>(dst + pos) = v;
    } else {
      dst[pos] = rows[pos];
    }
  }
}

// One wavefront per draw, ONE STEP PER LANE, 63 steps per block.  a_{t+1} = A a_t + w_t (A = T - g Z, g = T K, w_t = g y~_t + c)
// is a prefix "sum" under x -> A x: lane 0 carries the state the block starts from, lane j >= 1 the forcing w_{tb+j-1};
// six Hillis-Steele levels x_j += A^(2^k) x_{j - 2^k} leave a_{tb+j} in lane j.  The matrices A^(2^k) are built per draw in
// LDS (five squarings) and read back as broadcasts (the operands of a level are uniform across the wavefront; 600 doubles
// do not fit the scalar registers: a first version with scalar loads spilled them through v_writelane / v_readlane).
// Then every lane finishes its own step (v, a_filt, ll) and the block's rows leave through LDS as whole lines.
// A lane-per-draw version (sequential in time, the matrix in registers) was measured too: 3 instructions per draw-step, but
// its per-lane rows are scattered 8-byte stores (64 lines per instruction): 1.9 us per step, 1.46 ms for one chunk's draws.
template <int M>
__global__ __launch_bounds__(KF_WAVE) void kf_b4_steady_scan_kernel(const KfFwdArgs A, const B4SplitWs W, int chunk) {
  constexpr int MM = M * M, N4 = 4 * ((M + 2 + 3) / 4), JA = N4 - 2, JC = N4 - 1;
  constexpr int STEPS = KF_WAVE - 1;
  __shared__ __attribute__((aligned(16))) double sAP[6 * MM + 2];
  __shared__ __attribute__((aligned(16))) double sV[5 * M + 2];
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
    const double* ht = W.hs_tile + b * (long)B4_HS_STRIDE;
    const double* gT = A.Tm + b * (long)MM;
    const double* gZ = A.Z + b * (long)M;
    const double F = A.H[b] + ht[JA * 16 + JA];
    const double rs = fast_rsqrt(fmax(F, 1e-300));
    wave_lds_fence();
    if (lane < M) {
      sV[lane] = ht[JA * 16 + lane] * rs * rs;  // K = P Z' / F
      sV[2 * M + lane] = gZ[lane];
      sV[3 * M + lane] = A.c ? A.c[b * (long)M + lane] : 0.0;
      sV[4 * M + lane] = 0.5 * ht[JC * 16 + lane];  // the mean at the hand-over step
    }
    wave_lds_fence();
    if (lane < M) {
      double gv = 0.0;
#pragma unroll
      for (int l = 0; l < M; ++l) gv = fma(gT[lane * M + l], sV[l], gv);
      sV[M + lane] = gv;  // g = T K
    }
    wave_lds_fence();
    for (int x = lane; x < MM; x += KF_WAVE) {
      const int i = x / M, j = x - i * M;
      sAP[x] = fma(-sV[M + i], sV[2 * M + j], gT[x]);  // A = T - g Z
    }
    wave_lds_fence();
#pragma unroll 1
    for (int k = 1; k < 6; ++k) {
      const double* src = sAP + (k - 1) * MM;
      for (int x = lane; x < MM; x += KF_WAVE) {
        const int i = x / M, j = x - i * M;
        double v = 0.0;
#pragma unroll
        for (int l = 0; l < M; ++l) v = fma(src[i * M + l], src[l * M + j], v);
        sAP[k * MM + x] = v;
      }
      wave_lds_fence();
    }
    double K[M], g[M], Z[M], c[M], acur[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      K[i] = sV[i];
      g[i] = sV[M + i];
      Z[i] = sV[2 * M + i];
      c[i] = sV[3 * M + i];
      acur[i] = sV[4 * M + i];
    }
    const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
    const double d1 = A.d ? A.d[b] : 0.0;
    const bool skipped = univariate && F == 0.0;  // :468-471
    const bool notpd = !skipped && !(F > 0.0);
    const double llconst = -0.5 * (KF_LOG_2PI + log(F));
    double acc = 0.0;
    for (int tb = ts; tb < n; tb += STEPS) {
      const int t = tb + lane;
      const int cnt = (n - tb) < STEPS ? (n - tb) : STEPS;  // steps of this block
      const bool act = lane < cnt;
      const double yt = t < n ? ybase[t] + dsign * d1 : 0.0;
      const double yprev = __shfl_up(yt, 1);
      double x[M], xs[M];
#pragma unroll
      for (int i = 0; i < M; ++i) x[i] = lane == 0 ? acur[i] : fma(g[i], yprev, c[i]);
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
          for (int l = 0; l < M; ++l) v = fma(sAP[k * MM + i * M + l], xs[l], v);
          x[i] = v;
        }
      }
      // lane j holds a_{tb+j}; the predicted mean of its step is the state of the lane above
      double za = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) za = fma(Z[i], x[i], za);
      const double v = yt - za;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const double nx = __shfl_down(x[i], 1);
        rowsP[lane * M + i] = nx;
        rowsF[lane * M + i] = fma(K[i], v, x[i]);
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

int kf_b4_split_compact(kf_handle* h, const B4SplitWs& W, long B, int chunk, int t_next, int pair_ok, hipStream_t stream) {
  hipLaunchKernelGGL(kf_b4_split_compact_kernel, dim3(1), dim3(1024), 0, stream, W, (int)B, chunk, t_next, pair_ok);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

static int b4_fill_slices(const KfFwdArgs& a) {
  // slices of ~64 KiB per tape
  const long ns = ((long)a.T * a.m * a.m * 8 + 65535) / 65536;
  return (int)(ns < 1 ? 1 : ns > 32 ? 32 : ns);
}

// behind the compaction of `chunk`: a throttled fill launch (s_fill) and the mean scan of the draws that chunk handed over (s_mean)
int kf_b4_split_steady(kf_handle* h, const KfFwdArgs& a, const B4SplitWs& W, int chunk, hipStream_t s_fill, hipStream_t s_mean) {
  const int m = a.m, mm = m * m;
  if (a.filt_P || a.pred_P) {
    const char* e = kf_env_experiment("KF_B4_FILL_WGS");
    int wgs = e && *e ? atoi(e) : 1024;  // (C3, 8192 x 2000: 512 / 1024 / 2048 workgroups 6.1 / 5.7 / 5.9 ms)
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(kf_b4_steady_fill_kernel, dim3((unsigned)wgs), dim3(KF_WAVE), sizeof(double) * (size_t)(2 * mm + 4), s_fill, a, W,
                       chunk, b4_fill_slices(a));
  }
  const unsigned grid = (unsigned)(a.B < 2048 ? a.B : 2048);
#define KF_MEAN(M_) \
  case M_: hipLaunchKernelGGL((kf_b4_steady_scan_kernel<M_>), dim3(grid), dim3(KF_WAVE), 0, s_mean, a, W, chunk); break;
  switch (m) {
    KF_MEAN(1) KF_MEAN(2) KF_MEAN(3) KF_MEAN(4) KF_MEAN(5) KF_MEAN(6) KF_MEAN(7) KF_MEAN(8) KF_MEAN(9) KF_MEAN(10)
    default:
      KF_SET_ERR(h, "forward(b4 split): no steady kernel for m = %d", m);
      return KF_ERR_UNSUPPORTED;
  }
#undef KF_MEAN
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

// behind the last recursion launch: whatever the throttled launches left, at full width
int kf_b4_split_drain(kf_handle* h, const KfFwdArgs& a, const B4SplitWs& W, int last_chunk, hipStream_t stream) {
  if (!(a.filt_P || a.pred_P)) return KF_OK;
  const int mm = a.m * a.m;
  hipLaunchKernelGGL(kf_b4_steady_fill_kernel, dim3(2048), dim3(KF_WAVE), sizeof(double) * (size_t)(2 * mm + 4), stream, a, W,
                     last_chunk, b4_fill_slices(a));
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
