// kf_lyapunov.hip -- batched discrete Lyapunov solve  P = T P T' + S  and batched theta ->
// matrix assembly: the step immediately BEFORE the filter path.
//
// Reference: the models' ``update(theta)`` writes slices of the parameter vector into the named
// state-space matrices (models/SARIMAX.py:59-98, models/VARMAX.py:95-143) and, with
// ``stationary_initialization``, sets P0 = solve_discrete_lyapunov(T, R Q R', method="bilinear")
// (models/SARIMAX.py:100-107, models/VARMAX.py:145-150) -- once per draw, on the host.
// Here both run for all B draws at once:
//   kf_assemble_kernel   out[b][:] = template[:]; out[b][dst[i]] = theta[b][src[i]]
//   kf_lyap_lds_kernel   Smith's doubling iteration in LDS per draw:
//                          P <- S, A <- T;  repeat  P <- P + A P A',  A <- A A
//                        (k iterations sum 2^k terms of sum_j T^j S T'^j; quadratic convergence
//                        for spectral radius < 1; info[b] = 1 if not converged in 48 iterations)
#include "kf_internal.h"

namespace {

__device__ __forceinline__ void wsync() { __syncthreads(); }

template <int G>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, G));
  return v;
}

__host__ __device__ inline int lyap_region_doubles(int m) { return (4 * m * m + 2) | 1; }

template <int G>
__global__ __launch_bounds__(KF_WAVE) void kf_lyap_lds_kernel(long B, int m, int r, const double* Tm,
                                                              const double* Rm, const double* Qm,
                                                              const double* Sm, double* Pout,
                                                              int32_t* info) {
  extern __shared__ double smem[];
  constexpr int DPB = KF_WAVE / G;
  const int lane = threadIdx.x, lig = lane % G, grp = lane / G;
  const int mm = m * m;
  long b = (long)blockIdx.x * DPB + grp;
  const bool live = b < B;
  if (!live) b = B - 1;
  double* S = smem + (size_t)grp * lyap_region_doubles(m);
  double* A = S;
  double* P = A + mm;
  double* W = P + mm;
  double* A2 = W + mm;
  for (int e = lig; e < mm; e += G) {
    A[e] = Tm[b * mm + e];
    double s;
    if (Sm) {
      s = Sm[b * mm + e];
    } else {  // S = R Q R'
      const double* gR = Rm + b * (long)m * r;
      const double* gQ = Qm + b * (long)r * r;
      int i = e / m, j = e - i * m;
      s = 0.0;
      for (int l = 0; l < r; ++l) {
        double rq = 0.0;
        for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
        s += rq * gR[j * r + l];
      }
    }
    P[e] = s;
  }
  wsync();
  int status = 1;
  for (int it = 0; it < 48; ++it) {
    // W = A P ; A2 = A A
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double s = 0.0, s2 = 0.0;
      for (int k = 0; k < m; ++k) {
        double aik = A[i * m + k];
        s += aik * P[k * m + j];
        s2 += aik * A[k * m + j];
      }
      W[e] = s;
      A2[e] = s2;
    }
    wsync();
    // P += sym(W A') ; A = A2 ; convergence: max |increment| <= 1e-17 * max |P|
    double dmax = 0.0, pmax = 0.0;
    for (int e = lig; e < mm; e += G) {
      int i = e / m, j = e - i * m;
      double sij = 0.0, sji = 0.0;
      for (int k = 0; k < m; ++k) {
        sij += W[i * m + k] * A[j * m + k];
        sji += W[j * m + k] * A[i * m + k];
      }
      double inc = 0.5 * (sij + sji);
      double pn = P[e] + inc;
      dmax = fmax(dmax, fabs(inc));
      pmax = fmax(pmax, fabs(pn));
      P[e] = pn;  // each lane owns element e: no cross-lane hazard on P here
    }
    dmax = group_max<G>(dmax);
    pmax = group_max<G>(pmax);
    wsync();
    for (int e = lig; e < mm; e += G) A[e] = A2[e];
    wsync();
    const bool done = !(dmax > 1e-17 * pmax);  // also true for NaN-free exact zero increments
    if (done && pmax < 1.0e300 && dmax == dmax) status = 0;  // finite fixed point reached
    if (!__any(!(done))) break;
  }
  if (live) {
    for (int e = lig; e < mm; e += G) Pout[b * mm + e] = P[e];
    if (info && lig == 0) info[b] = status;
  }
}

__global__ void kf_assemble_kernel(long B, int n_theta, const double* theta, long total,
                                   const double* tmpl, int n_map, const int32_t* dst,
                                   const int32_t* src, double* out) {
  const long b = blockIdx.y;
  double* ob = out + b * total;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    ob[e] = tmpl[e];
  __syncthreads();
  // scatter after the template copy of THIS block's range would race across blocks: every block
  // scatters only destinations inside its own grid-stride range
  for (int i = threadIdx.x; i < n_map; i += blockDim.x) {
    long e = dst[i];
    long owner = (e / blockDim.x) % gridDim.x;
    if (owner == blockIdx.x) ob[e] = theta[b * n_theta + src[i]];
  }
}

template <int G>
int launch_lyap(kf_handle* h, long B, int m, int r, const double* T, const double* R, const double* Q,
                const double* S, double* P, int32_t* info, hipStream_t stream) {
  const int dpb = KF_WAVE / G;
  size_t lds = sizeof(double) * (size_t)dpb * lyap_region_doubles(m);
  if (lds > KF_LDS_BYTES_MAX) {
    KF_SET_ERR(h, "lyapunov(lds): m=%d needs %zu B of LDS per workgroup", m, lds);
    return KF_ERR_UNSUPPORTED;
  }
  KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kf_lyap_lds_kernel<G>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long blocks = (B + dpb - 1) / dpb;
  hipLaunchKernelGGL(kf_lyap_lds_kernel<G>, dim3((unsigned)blocks), dim3(KF_WAVE), lds, stream, B, m, r,
                     T, R, Q, S, P, info);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

}  // namespace

int kf_launch_lyapunov(kf_handle* h, long B, int m, int r, const double* T, const double* R,
                       const double* Q, const double* S, double* P, int32_t* info, hipStream_t stream) {
  int G = kf_pick_lanes(h, m, B);
  while (G < KF_WAVE && sizeof(double) * (size_t)(KF_WAVE / G) * lyap_region_doubles(m) > KF_LDS_BYTES_MAX / 2)
    G *= 2;
  switch (G) {
    case 4: return launch_lyap<4>(h, B, m, r, T, R, Q, S, P, info, stream);
    case 8: return launch_lyap<8>(h, B, m, r, T, R, Q, S, P, info, stream);
    case 16: return launch_lyap<16>(h, B, m, r, T, R, Q, S, P, info, stream);
    case 32: return launch_lyap<32>(h, B, m, r, T, R, Q, S, P, info, stream);
    default: return launch_lyap<64>(h, B, m, r, T, R, Q, S, P, info, stream);
  }
}

int kf_launch_assemble(kf_handle* h, long B, int n_theta, const double* theta, long total,
                       const double* tmpl, int n_map, const int32_t* dst, const int32_t* src,
                       double* out, hipStream_t stream) {
  if (B > 65535) {
    KF_SET_ERR(h, "assemble: B=%ld > 65535 draws per call", B);
    return KF_ERR_UNSUPPORTED;
  }
  int bx = (int)((total + 255) / 256);
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(kf_assemble_kernel, dim3(bx, (unsigned)B), dim3(256), 0, stream, B, n_theta, theta,
                     total, tmpl, n_map, dst, src, out);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
