// Issue cost (shader cycles per wave-instruction, s_memtime) of the fp64 / cross-lane instructions the
// tile kernels are made of, at 1 and 4 wavefronts per SIMD, plus the operand layout of
// v_mfma_f64_4x4x4_4b_f64 (discovered with one-hot operands).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/f64_ops.hip -o scripts/probes/f64_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double dpp_ror1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x121, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x121, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pl16(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0] ^ b[1], a[0] ^ a[1]);
}
__device__ __forceinline__ double rdl(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// MODE: 0 mfma16x16x4 (4 indep acc) | 1 mfma 4x4x4_4b (8 indep acc) | 2 v_fma_f64 (8 indep) | 3 v_mul_f64 | 4 v_add_f64
//       5 dpp row_ror mov pair | 6 permlane16_swap pair (+2 xor) | 7 readlane pair | 8 v_rcp_f64 | 9 mfma16 dependent chain
//       10 mfma4x4 dependent chain | 11 v_fma_f64 dependent chain | 12 ds_bpermute pair | 13 v_fma_f32 8 indep
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters) {
  double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.5 + threadIdx.x * 1e-7;
  double f[8];
  for (int j = 0; j < 8; ++j) f[j] = 1.0 + j + threadIdx.x * 1e-3;
  float g[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float af = (float)a, bf = (float)b;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f[j], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fma(f[j], a, b);
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = f[j] * a;
    } else if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = f[j] + b;
    } else if (MODE == 5) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = dpp_ror1(f[j]);
    } else if (MODE == 6) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = pl16(f[j]);
    } else if (MODE == 7) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = rdl(f[j], (i + j) & 63);
    } else if (MODE == 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_rcp(f[j]);
    } else if (MODE == 9) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
    } else if (MODE == 10) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f[0], 0, 0, 0);
    } else if (MODE == 11) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[0] = fma(f[0], a, b);
    } else if (MODE == 12) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int lo = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) * 4, __double2loint(f[j]));
        int hi = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) * 4, __double2hiint(f[j]));
        f[j] = __hiloint2double(hi, lo);
      }
    } else if (MODE == 13) {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = fmaf(g[j], af, bf);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
  for (int j = 0; j < 8; ++j) s += f[j] + g[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(double* d, long long* c, const char* name, int wg_per_cu) {
  const int iters = 4000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, c, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, c, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), c, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= blocks;
  // per wave-instruction as seen by ONE wave; x waves/SIMD = SIMD occupancy of the pipe
  printf("%-34s waves/SIMD %d  %8.3f ms  %8.1f cyc per 8-group per wave  -> %6.1f cyc/instr-slot at the SIMD\n", name,
         wg_per_cu, ms, mean / iters, mean / iters / 8.0 / wg_per_cu);
}

__global__ void layout(const double* A, const double* B, double* D) {
  // 64 x 64 one-hot probes: probe q = (la, lb): A one-hot at lane la, B one-hot at lane lb
  const int q = blockIdx.x, la = q >> 6, lb = q & 63, l = threadIdx.x;
  double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  D[q * 64 + l] = d;
}

int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 1024 * 4);
  long long* c; hipMalloc(&c, sizeof(long long) * 4096);
  for (int w : {1, 4}) {
    run<0>(d, c, "v_mfma_f64_16x16x4 (indep)", w);
    run<9>(d, c, "v_mfma_f64_16x16x4 (dependent)", w);
    run<1>(d, c, "v_mfma_f64_4x4x4_4b (indep)", w);
    run<10>(d, c, "v_mfma_f64_4x4x4_4b (dependent)", w);
    run<2>(d, c, "v_fma_f64 (indep)", w);
    run<11>(d, c, "v_fma_f64 (dependent)", w);
    run<3>(d, c, "v_mul_f64", w);
    run<4>(d, c, "v_add_f64", w);
    run<8>(d, c, "v_rcp_f64", w);
    run<13>(d, c, "v_fma_f32", w);
    run<5>(d, c, "dpp row_ror mov x2 (one f64)", w);
    run<6>(d, c, "permlane16_swap x2 + 2 xor (f64)", w);
    run<7>(d, c, "v_readlane x2 (+2 v_mov) (f64)", w);
    run<12>(d, c, "ds_bpermute x2 (one f64)", w);
  }
  // layout of the 4x4x4 4-block instruction
  double* D; hipMalloc(&D, sizeof(double) * 4096 * 64);
  hipLaunchKernelGGL(layout, dim3(4096), dim3(64), 0, 0, nullptr, nullptr, D);
  std::vector<double> h(4096 * 64);
  hipMemcpy(h.data(), D, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
  printf("# v_mfma_f64_4x4x4_4b_f64 one-hot map: A lane la x B lane lb -> D lane (only non-zero products listed)\n");
  int count = 0;
  for (int q = 0; q < 4096; ++q)
    for (int l = 0; l < 64; ++l)
      if (h[q * 64 + l] != 0.0) {
        ++count;
        if ((q >> 6) < 20 || (q >> 6) % 16 == 0) printf("la=%2d lb=%2d -> ld=%2d\n", q >> 6, q & 63, l);
      }
  printf("# %d non-zero (la, lb) pairs (expect 4 blocks x 4 x 4 x 4 = 256)\n", count);
  return 0;
}
