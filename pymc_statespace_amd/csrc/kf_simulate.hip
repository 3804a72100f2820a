// kf_simulate.hip -- batched conditional simulation: the step immediately AFTER the filter path
// (SURVEY.md section 8f, row 2).
//
// Reference: conditional_simulation / numba_mvn_draws (pymc_statespace/utils/simulation.py:7-26)
// draw trajectories x ~ N(mu, blockdiag(cov_1..cov_n) + jitter I) per posterior sample by
// building the (n k) x (n k) block-diagonal matrix and Cholesky-factoring it -- O((n k)^3) per
// trajectory.  A block-diagonal Cholesky is the Cholesky of its blocks, so here one lane owns one
// (draw, time step): it factors the k x k block once (in LDS) and maps every simulation's
// standard-normal vector through it:  x[b, j, t, :] = mu[b, t, :] + L[b, t] z[b, j, t, :].
// The normals are an INPUT (drawn by the caller's generator), which keeps the kernel
// deterministic and testable; the reference's random jitter U(1e-12, 1e-8) becomes a parameter.
#include "kf_internal.h"

namespace {

__global__ __launch_bounds__(KF_WAVE) void kf_simulate_kernel(long B, int n, int k, int n_sims,
                                                              const double* mu, const double* cov,
                                                              const double* z, double jitter,
                                                              double* out, int32_t* info) {
  extern __shared__ double smem[];
  const long idx = (long)blockIdx.x * KF_WAVE + threadIdx.x;  // (draw, t) flattened
  const long total = B * (long)n;
  const bool live = idx < total;
  const long id = live ? idx : total - 1;
  const long b = id / n;
  const int t = (int)(id - b * n);
  const int kk = k * k;
  double* L = smem + (size_t)threadIdx.x * (kk | 1);  // odd stride: lanes on different banks
  const double* C = cov + id * kk;
  int fail = 0;
  for (int j = 0; j < k; ++j) {
    double s = C[j * k + j] + jitter;
    for (int q = 0; q < j; ++q) s -= L[j * k + q] * L[j * k + q];
    if (!(s > 0.0) && fail == 0) fail = j + 1;
    const double dj = sqrt(s);
    L[j * k + j] = dj;
    for (int i = j + 1; i < k; ++i) {
      double v = C[i * k + j];
      for (int q = 0; q < j; ++q) v -= L[i * k + q] * L[j * k + q];
      L[i * k + j] = v / dj;
    }
  }
  if (live && info && fail) atomicMax(info + b, ((t + 1) << 8) | fail);
  if (!live) return;
  const double* m_ = mu + id * k;
  for (int js = 0; js < n_sims; ++js) {
    const long row = ((b * n_sims + js) * (long)n + t) * k;
    const double* zz = z + row;
    double* o = out + row;
    for (int i = 0; i < k; ++i) {
      double s = m_[i];
      for (int q = 0; q <= i; ++q) s += L[i * k + q] * zz[q];
      o[i] = s;
    }
  }
}

}  // namespace

int kf_launch_simulate(kf_handle* h, long B, int n, int k, int n_sims, const double* mu,
                       const double* cov, const double* z, double jitter, double* out, int32_t* info,
                       hipStream_t stream) {
  size_t lds = sizeof(double) * KF_WAVE * (size_t)((k * k) | 1);
  if (lds > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "simulate: k=%d needs %zu B of LDS per workgroup", k, lds);
    return KF_ERR_UNSUPPORTED;
  }
  KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_simulate_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (info) KF_HIP_CHECK(h, hipMemsetAsync(info, 0, sizeof(int32_t) * B, stream));
  long blocks = (B * (long)n + KF_WAVE - 1) / KF_WAVE;
  hipLaunchKernelGGL(kf_simulate_kernel, dim3((unsigned)blocks), dim3(KF_WAVE), lds, stream, B, n, k,
                     n_sims, mu, cov, z, jitter, out, info);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
