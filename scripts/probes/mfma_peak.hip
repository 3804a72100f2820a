// Achievable fp64 MFMA and fp64 VALU FMA rates (register operands only), and whether they overlap.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: MFMA only, 1: VALU FMA only, 2: both interleaved
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
  double f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int i = 0; i < iters; ++i) {
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    if (MODE != 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fma(f[j], a, b);
    }
  }
  double s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
  for (int j = 0; j < 8; ++j) s += f[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(double* d, const char* name) {
  const int iters = 20000, blocks = 256 * 4;  // 4 workgroups of 4 waves per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks * 4.0;
  const double mf = (MODE != 1) ? waves * iters * 4.0 * 2048.0 : 0.0;
  const double vf = (MODE != 0) ? waves * iters * 64.0 * 64.0 * 2.0 : 0.0;
  printf("%-28s %8.3f ms   MFMA %6.1f TF   VALU %6.1f TF\n", name, ms, mf / ms / 1e9, vf / ms / 1e9);
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 1024 * 4);
  run<0>(d, "mfma_f64_16x16x4 only");
  run<1>(d, "v_fma_f64 only");
  run<2>(d, "both in one wave");
  return 0;
}
