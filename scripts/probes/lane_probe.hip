// Probe of the gfx950 cross-lane primitives the MFMA forward kernel relies on (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ double dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__global__ void k(double* o) {
  const int l = threadIdx.x;
  double x = (double)l;
  // permlane32_swap / permlane16_swap with both operands = x
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  o[0 * 64 + l] = __hiloint2double(b[0], a[0]);
  o[1 * 64 + l] = __hiloint2double(b[1], a[1]);
  auto c = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  o[2 * 64 + l] = __hiloint2double(d[0], c[0]);
  o[3 * 64 + l] = __hiloint2double(d[1], c[1]);
  o[4 * 64 + l] = dpp<0x128>(x);  // row_ror:8
  o[5 * 64 + l] = dpp<0x124>(x);  // row_ror:4
  o[6 * 64 + l] = dpp<0x122>(x);  // row_ror:2
  o[7 * 64 + l] = dpp<0x121>(x);  // row_ror:1
  // mfma f64 16x16x4: A[i][k] = 100*i + k (lane: i = l&15, k = l>>4), B = identity-like probe
  double4_t acc = {0, 0, 0, 0};
  const double av = 100.0 * (l & 15) + (l >> 4);
  const double bv = ((l >> 4) == 0 && (l & 15) == 3) ? 1.0 : (((l >> 4) == 2 && (l & 15) == 5) ? 1.0 : 0.0);
  // B[k][j]: B[0][3] = 1, B[2][5] = 1  ->  D[i][3] = A[i][0] = 100 i ; D[i][5] = A[i][2] = 100 i + 2
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) o[(8 + r) * 64 + l] = acc[r];
}

int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 64 * 12);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[64 * 12];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[12] = {"p32swap[0]", "p32swap[1]", "p16swap[0]", "p16swap[1]", "row_ror8", "row_ror4",
                           "row_ror2", "row_ror1", "mfma r0", "mfma r1", "mfma r2", "mfma r3"};
  for (int r = 0; r < 12; ++r) {
    printf("%-11s:", names[r]);
    for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]);
    printf("\n");
  }
  return 0;
}
