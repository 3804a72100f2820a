// one-hot map of the A operand under cbsz/abid for v_mfma_f64_4x4x4_4b_f64: which D lanes does A lane la feed (B = ones)?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CBSZ, int ABID>
__global__ void k(double* D) {
  const int la = blockIdx.x, l = threadIdx.x;
  D[la * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0 + 0.0 * l, l == la ? 1.0 : 0.0, 0.0, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID>
void run(double* d) {
  static double h[4096];
  hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(64), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("cbsz=%d abid=%d: B lane (g=k,b,q=j) -> D lanes (g,b) with q=all:\n", CBSZ, ABID);
  for (int la = 0; la < 64; ++la) {
    printf("  B(g%d b%d q%d) ->", la >> 4, (la >> 2) & 3, la & 3);
    for (int l = 0; l < 64; l += 1)
      if (h[la * 64 + l] != 0.0) printf(" (g%d b%d q%d)", l >> 4, (l >> 2) & 3, l & 3);
    printf("\n");
  }
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 4096);
  run<0, 0>(d); run<1, 0>(d); run<1, 1>(d); run<2, 0>(d); run<2, 1>(d); run<2, 3>(d);
  return 0;
}
