// kf_forward_whiten.hip -- dense observation covariance H with p > 1 on the scalar-observation kernels, by WHITENING.
//
// The reference's Standard / Cholesky updates take any positive definite H (filters/kalman_filter.py:262-275, 289-305).  The block /
// tile kernels here process scalar observations one (or four) at a time, which needs a DIAGONAL H; rounds 1-3 sent every dense H to
// the round-1 LDS kernel (C4 with a dense H: 106 ms against 12.6 ms, profiles/r04/tv_dense_lines.txt).  With H = L L' (Cholesky, per
// draw) the model  y = Z alpha + d + eps, eps ~ N(0, H)  is the model  y* = Z* alpha + d* + eps*, eps* ~ N(0, I)  with
//     y* = L^-1 y     Z* = L^-1 Z     d* = L^-1 d
// and the two filters produce the SAME state moments at every step (the innovation covariances are congruent, F = L F* L', the gain
// K = K* L^-1, K v = K* v*), while  log det F = log det F* + 2 log det L  and  v' F^-1 v = v*' F*^-1 v*:  the log-likelihood of an
// observed step is the whitened one minus log det L.  So:
//     kf_whiten_kernel      per draw: L = chol(H), Z*, d*, the whole series y* (one series PER DRAW from here on), log det L, the
//                           number of fully observed steps; raises a device flag when the trick does not apply --
//                           a PARTIALLY missing row (L^-1 spreads the NaN over the observations behind it: the row-deletion
//                           semantics of such a step, pinned by statsmodels, need the original observations) or an H that is not
//                           positive definite (a Cholesky pivot below 1e-8 of H's largest diagonal entry counts as that)
//     (the selected diagonal-H forward kernel, KF_FLAG_H_DIAGONAL | KF_FLAG_Y_PER_DRAW, H* = I)
//     kf_whiten_fix_kernel  logp and ll_obs minus log det L per observed step
//     kf_fwd_lds_kernel     launched behind them with KfFwdArgs::run_if = the flag: exits at once when the flag is clear, redoes the
//                           whole batch on the original inputs when it is set (no host synchronisation, graph-capturable)
// A whole-row NaN whitens to a whole-row NaN (the reference's own missing-data case, :195-213).  The leading minors of F* have the
// signs of those of F (congruence by a lower-triangular factor), so `info` is unchanged.  Results agree with the dense path to
// rounding (tests/test_gpu_r04.py, 1e-9).
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int WH_P = 8;

struct KfWhitenWs {
  double* ys;       // [B][T][p]
  double* Zs;       // [B][p][m]
  double* ds;       // [B][p]
  double* Hs;       // [B][p][p] identity
  double* logdet;   // [B]
  int32_t* nobs;    // [B] fully observed steps
  int32_t* flag;    // [1] != 0: fall back to the dense path
  double* Li = nullptr;   // [B][p][p] L^-1, kept for the gradient's way back (kf_unwhiten_grad_kernel); NULL for a forward call
};

__global__ __launch_bounds__(KF_WAVE) void kf_whiten_kernel(const KfFwdArgs A, const KfWhitenWs W) {
  __shared__ double L[WH_P * WH_P];
  __shared__ double Li[WH_P * WH_P];  // L^-1 (lower triangular)
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const int p = A.p, m = A.m, n = A.T;
  const double* gH = A.H + b * (long)p * p;
  if (lane < WH_P * WH_P) {
    const int i = lane / WH_P, j = lane % WH_P;
    L[lane] = (i < p && j < p) ? 0.5 * (gH[i * p + j] + gH[j * p + i]) : (i == j ? 1.0 : 0.0);
    Li[lane] = 0.0;
  }
  __syncthreads();
  bool bad = false;
  if (lane == 0) {
    double ld = 0.0;
    //  a RELATIVE pivot rule: a pivot below 1e-8 of the largest diagonal entry -- cond(H) beyond ~1e8, L^-1 of norm
    // ~1e4 -- is left to the dense path like an H that is not positive definite (the smoother split's Cholesky uses 1e-2 for the
    // same reason: an absolute `s > 0` accepts a pivot of 1e-300)
    double hmax = 0.0;
    for (int j = 0; j < p; ++j) hmax = fmax(hmax, L[j * WH_P + j]);
    for (int j = 0; j < p; ++j) {
      double s = L[j * WH_P + j];
      for (int k = 0; k < j; ++k) s -= L[j * WH_P + k] * L[j * WH_P + k];
      if (!(s > 1e-8 * hmax)) {
        bad = true;
        s = 1.0;
      }
      const double dj = sqrt(s);
      L[j * WH_P + j] = dj;
      ld += log(dj);
      for (int i = j + 1; i < p; ++i) {
        double t = L[i * WH_P + j];
        for (int k = 0; k < j; ++k) t -= L[i * WH_P + k] * L[j * WH_P + k];
        L[i * WH_P + j] = t / dj;
      }
    }
    W.logdet[b] = ld;
    // L^-1 column by column
    for (int c = 0; c < p; ++c) {
      for (int i = 0; i < p; ++i) {
        double t = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; ++k) t -= L[i * WH_P + k] * Li[k * WH_P + c];
        Li[i * WH_P + c] = (i >= c) ? t / L[i * WH_P + i] : 0.0;
      }
    }
    if (bad) atomicOr(W.flag, 2);
  }
  __syncthreads();
  if (W.Li)
    for (int e = lane; e < p * p; e += KF_WAVE) W.Li[b * (long)p * p + e] = Li[(e / p) * WH_P + e % p];
  // Z* = L^-1 Z, d* = L^-1 d, H* = I
  const double* gZ = A.Z + b * (long)p * m;
  for (int e = lane; e < p * m; e += KF_WAVE) {
    const int i = e / m, j = e - i * m;
    double s = 0.0;
    for (int k = 0; k <= i; ++k) s = fma(Li[i * WH_P + k], gZ[k * m + j], s);
    W.Zs[b * (long)p * m + e] = s;
  }
  if (lane < p) {
    double s = 0.0;
    if (A.d)
      for (int k = 0; k <= lane; ++k) s = fma(Li[lane * WH_P + k], A.d[b * (long)p + k], s);
    W.ds[b * (long)p + lane] = s;
  }
  for (int e = lane; e < p * p; e += KF_WAVE) W.Hs[b * (long)p * p + e] = (e / p == e % p) ? 1.0 : 0.0;
  // y* = L^-1 y, one step per lane
  const double* gy = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L);
  double* oy = W.ys + b * (long)n * p;
  int full = 0;
  bool partial = false;
  for (int t = lane; t < n; t += KF_WAVE) {
    double yv[WH_P];
    int nmiss = 0;
    for (int i = 0; i < p; ++i) {
      yv[i] = gy[(long)t * p + i];
      nmiss += (yv[i] != yv[i]) ? 1 : 0;
    }
    if (nmiss == 0) ++full;
    else if (nmiss < p) partial = true;
    for (int i = 0; i < p; ++i) {
      double s = 0.0;
      for (int k = 0; k <= i; ++k) s = fma(Li[i * WH_P + k], yv[k], s);
      oy[(long)t * p + i] = (nmiss == 0) ? s : __longlong_as_double(0x7ff8000000000000LL);   // a missing row stays a missing row
    }
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) full += __shfl_xor(full, off);
  if (lane == 0) W.nobs[b] = full;
  if (__ballot(partial) != 0ull && lane == 0) atomicOr(W.flag, 1);
}

__global__ void kf_whiten_fix_kernel(const KfFwdArgs A, const KfWhitenWs W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = A.T, p = A.p;
  if (A.ll_obs) {
    if (i < A.B * (long)n) {
      const long b = i / n;
      const int t = (int)(i - b * n);
      const double* gy = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n * p : 0L) + (long)t * p;
      bool obs = true;
      for (int k = 0; k < p; ++k) obs = obs && !(gy[k] != gy[k]);
      if (obs) A.ll_obs[i] -= W.logdet[b];
    }
  }
  if (i < A.B) A.logp[i] -= (double)W.nobs[i] * W.logdet[i];
}

__global__ void kf_whiten_clear_kernel(int32_t* flag) { *flag = 0; }

size_t wh_round(size_t x) { return (x + 255) & ~(size_t)255; }
size_t wh_carve(void* base, long B, int T, int p, int m, KfWhitenWs* W) {
  char* q = reinterpret_cast<char*>(base);
  size_t off = 0;
  W->ys = reinterpret_cast<double*>(q + off); off += wh_round(8 * (size_t)B * T * p);
  W->Zs = reinterpret_cast<double*>(q + off); off += wh_round(8 * (size_t)B * p * m);
  W->ds = reinterpret_cast<double*>(q + off); off += wh_round(8 * (size_t)B * p);
  W->Hs = reinterpret_cast<double*>(q + off); off += wh_round(8 * (size_t)B * p * p);
  W->logdet = reinterpret_cast<double*>(q + off); off += wh_round(8 * (size_t)B);
  W->nobs = reinterpret_cast<int32_t*>(q + off); off += wh_round(4 * (size_t)B);
  W->flag = reinterpret_cast<int32_t*>(q + off); off += 256;
  return off + 256;
}

// The gradient's way back (round 5).  The whitened call returned, per draw, g_Z* = dlogp/dZ*, g_d*, and Gamma~ = sum_t vb_t y~*_t'
// (vb: the adjoint of the innovation; y~* = y* - d*).  The joint likelihood is invariant under y*, Z*, d* -> A y*, A Z*, A d*,
// H* -> A H* A' up to -n log|det A| per fully observed step; its derivative at A = I, H* = I gives the FULL p x p
//     G* = dlogp*/dH* = -(g n I + Gamma~ + g_Z* Z*') / 2        (g: the draw's cotangent; the d terms cancel),
// which no scalar-observation adjoint can produce by itself (they see diag H* only).  With H = L H* L' for the fixed L:
//     g_H = L^-T sym(G*) L^-1        g_Z = L^-T g_Z*        g_d = L^-T g_d*        logp = logp* - n log det L.
__global__ __launch_bounds__(KF_WAVE) void kf_unwhiten_grad_kernel(long B, int p, int m, const KfWhitenGrad W, const double* g_logp,
                                                                    double* logp, double* g_Z, double* g_H, double* g_d) {
  __shared__ double Li[WH_P * WH_P], G[WH_P * WH_P], X[WH_P * WH_P];
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  if (*W.flag != 0) return;   // (the dense path redoes the batch)
  const double gl = g_logp ? g_logp[b] : 1.0;
  const double nb = (double)W.nobs[b];
  const int i = lane / WH_P, j = lane % WH_P;
  Li[lane] = (i < p && j < p) ? W.Li[b * (long)p * p + i * p + j] : 0.0;
  double gij = 0.0;
  if (i < p && j < p) {
    gij = W.gyy[b * (long)p * p + i * p + j] + ((i == j) ? gl * nb : 0.0);
    const double* gz = W.gZs + b * (long)p * m + (long)i * m;
    const double* zs = W.Zs + b * (long)p * m + (long)j * m;
    for (int k = 0; k < m; ++k) gij = fma(gz[k], zs[k], gij);
  }
  G[lane] = -0.5 * gij;
  __syncthreads();
  const double gs = 0.5 * (G[i * WH_P + j] + G[j * WH_P + i]);
  __syncthreads();
  G[lane] = gs;
  __syncthreads();
  {  // X = sym(G*) L^-1, then g_H = L^-T X
    double s = 0.0;
    for (int k = 0; k < p; ++k) s = fma(G[i * WH_P + k], Li[k * WH_P + j], s);
    X[lane] = s;
  }
  __syncthreads();
  if (i < p && j < p && g_H) {
    double s = 0.0;
    for (int k = 0; k < p; ++k) s = fma(Li[k * WH_P + i], X[k * WH_P + j], s);
    g_H[b * (long)p * p + i * p + j] = s;
  }
  if (g_Z)
    for (int e = lane; e < p * m; e += KF_WAVE) {
      const int q = e / m, c = e - q * m;
      double s = 0.0;
      for (int k = q; k < p; ++k) s = fma(Li[k * WH_P + q], W.gZs[b * (long)p * m + (long)k * m + c], s);   // (L^-1 is lower triangular)
      g_Z[b * (long)p * m + e] = s;
    }
  if (g_d && lane < p) {
    double s = 0.0;
    for (int k = lane; k < p; ++k) s = fma(Li[k * WH_P + lane], W.gds[b * (long)p + k], s);
    g_d[b * (long)p + lane] = s;
  }
  if (lane == 0 && logp) logp[b] -= nb * W.logdet[b];
}

}  // namespace

size_t kf_whiten_grad_carve(void* base, long B, int T, int p, int m, KfWhitenGrad* W) {
  char* q = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = q + off; off += wh_round(bytes); return r; };
  W->flag = reinterpret_cast<int32_t*>(take(256));   // FIRST: the dense redo behind the whitened call reuses everything after it
  W->ys = reinterpret_cast<double*>(take(8 * (size_t)B * T * p));
  W->Zs = reinterpret_cast<double*>(take(8 * (size_t)B * p * m));
  W->ds = reinterpret_cast<double*>(take(8 * (size_t)B * p));
  W->Hs = reinterpret_cast<double*>(take(8 * (size_t)B * p * p));
  W->logdet = reinterpret_cast<double*>(take(8 * (size_t)B));
  W->Li = reinterpret_cast<double*>(take(8 * (size_t)B * p * p));
  W->gZs = reinterpret_cast<double*>(take(8 * (size_t)B * p * m));
  W->gds = reinterpret_cast<double*>(take(8 * (size_t)B * p));
  W->gHs = reinterpret_cast<double*>(take(8 * (size_t)B * p * p));
  W->gyy = reinterpret_cast<double*>(take(8 * (size_t)B * p * p));
  W->nobs = reinterpret_cast<int32_t*>(take(4 * (size_t)B));
  return off + 256;
}

int kf_launch_whiten_for_grad(kf_handle* h, const KfFwdArgs& a, const KfWhitenGrad& G, hipStream_t stream) {
  if (a.p > WH_P || a.p < 2) {
    KF_SET_ERR(h, "whitening: needs 2 <= p <= %d", WH_P);
    return KF_ERR_UNSUPPORTED;
  }
  KfWhitenWs W;
  W.ys = G.ys; W.Zs = G.Zs; W.ds = G.ds; W.Hs = G.Hs; W.logdet = G.logdet; W.nobs = G.nobs; W.flag = G.flag; W.Li = G.Li;
  hipLaunchKernelGGL(kf_whiten_clear_kernel, dim3(1), dim3(1), 0, stream, W.flag);
  hipLaunchKernelGGL(kf_whiten_kernel, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, W);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

int kf_launch_unwhiten_grad(kf_handle* h, long B, int p, int m, const KfWhitenGrad& W, const double* g_logp, double* logp,
                            double* g_Z, double* g_H, double* g_d, hipStream_t stream) {
  hipLaunchKernelGGL(kf_unwhiten_grad_kernel, dim3((unsigned)B), dim3(KF_WAVE), 0, stream, B, p, m, W, g_logp, logp, g_Z, g_H, g_d);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

size_t kf_forward_whiten_workspace_bytes(long B, int T, int p, int m) {
  KfWhitenWs W;
  return wh_carve(nullptr, B, T, p, m, &W);
}

// `inner`: the launcher of the diagonal-H path for the whitened arguments (kf_launch_forward itself: kf_api.hip decides eligibility)
int kf_launch_forward_whitened(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream,
                               int (*inner)(kf_handle*, const KfFwdArgs&, void*, size_t, hipStream_t)) {
  const size_t need = kf_forward_whiten_workspace_bytes(a.B, a.T, a.p, a.m);
  if (!workspace || workspace_bytes < need || a.p > WH_P || a.p < 2) {
    KF_SET_ERR(h, "forward(whitened): needs 2 <= p <= %d and %zu bytes of workspace", WH_P, need);
    return KF_ERR_WORKSPACE;
  }
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  KfWhitenWs W;
  const size_t used = wh_carve(base, a.B, a.T, a.p, a.m, &W);
  const size_t slack = (size_t)(base - reinterpret_cast<char*>(workspace));
  hipLaunchKernelGGL(kf_whiten_clear_kernel, dim3(1), dim3(1), 0, stream, W.flag);
  hipLaunchKernelGGL(kf_whiten_kernel, dim3((unsigned)a.B), dim3(KF_WAVE), 0, stream, a, W);
  KF_HIP_CHECK(h, hipGetLastError());
  KfFwdArgs w = a;
  w.y = W.ys; w.Z = W.Zs; w.d = W.ds; w.H = W.Hs;
  w.flags = a.flags | KF_FLAG_H_DIAGONAL | KF_FLAG_Y_PER_DRAW;
  int rc = inner(h, w, base + used, workspace_bytes > slack + used ? workspace_bytes - slack - used : 0, stream);
  if (rc) return rc;
  const long nfix = a.ll_obs ? a.B * (long)a.T : a.B;
  hipLaunchKernelGGL(kf_whiten_fix_kernel, dim3((unsigned)((nfix + 255) / 256)), dim3(256), 0, stream, a, W);
  KF_HIP_CHECK(h, hipGetLastError());
  // the dense path on the original inputs, only if the flag was raised (a partially missing row, an H that is not positive definite)
  KfFwdArgs f = a;
  f.run_if = W.flag;
  return kf_launch_forward_lds(h, f, stream);
}
