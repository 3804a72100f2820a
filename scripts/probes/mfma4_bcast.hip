// Semantics of cbsz / abid (A-block broadcast) and of the blgp "neg" bits on v_mfma_f64_4x4x4_4b_f64.
// Layout (scripts/probes/f64_ops.hip): lane = 16 g + 4 b + q;  A_b[i][k] at (g=k, q=i), B_b[k][j] at (g=k, q=j),
// D_b[i][j] at (g=i, q=j).  Expectation: with cbsz = s the A block `abid` of every group of 2^s blocks feeds the group.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

template <int CBSZ, int ABID, int BLGP>
__global__ void k(const double* A, const double* B, const double* C, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], CBSZ, ABID, BLGP);
}

static double hA[64], hB[64], hC[64], hD[64];
static inline int ln(int g, int b, int q) { return 16 * g + 4 * b + q; }

template <int CBSZ, int ABID, int BLGP>
int run(double* dA, double* dB, double* dC, double* dD) {
  hipLaunchKernelGGL((k<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  // hypotheses for the source block of A
  int ok_group = 1, ok_plain = 1, ok_neg[8];
  for (int n = 0; n < 8; ++n) ok_neg[n] = 1;
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        const int grp = 1 << CBSZ;
        const int src = (b / grp) * grp + (ABID % grp);
        double e_group = 0, e_plain = 0;
        for (int kk = 0; kk < 4; ++kk) {
          e_group += hA[ln(kk, src, i)] * hB[ln(kk, b, j)];
          e_plain += hA[ln(kk, b, i)] * hB[ln(kk, b, j)];
        }
        const double c = hC[ln(i, b, j)], d = hD[ln(i, b, j)];
        if (fabs(d - (e_group + c)) > 1e-12) ok_group = 0;
        if (fabs(d - (e_plain + c)) > 1e-12) ok_plain = 0;
        for (int n = 0; n < 8; ++n) {
          const double sa = (n & 1) ? -1 : 1, sb = (n & 2) ? -1 : 1, sc = (n & 4) ? -1 : 1;
          if (fabs(d - (sa * sb * e_group + sc * c)) > 1e-12) ok_neg[n] = 0;
        }
      }
  printf("cbsz=%d abid=%d blgp=%d : broadcast-in-group %s, plain %s ; neg pattern match (a|b<<1|c<<2):", CBSZ, ABID, BLGP,
         ok_group ? "YES" : "no", ok_plain ? "YES" : "no");
  for (int n = 0; n < 8; ++n)
    if (ok_neg[n]) printf(" %d", n);
  printf("\n");
  return ok_group;
}

int main() {
  srand(1);
  for (int l = 0; l < 64; ++l) {
    hA[l] = (rand() % 1000) / 500.0 - 1.0;
    hB[l] = (rand() % 1000) / 500.0 - 1.0;
    hC[l] = (rand() % 1000) / 500.0 - 1.0;
  }
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 512); hipMalloc(&dD, 512);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipMemcpy(dC, hC, 512, hipMemcpyHostToDevice);
  run<0, 0, 0>(dA, dB, dC, dD);
  run<2, 0, 0>(dA, dB, dC, dD);
  run<2, 1, 0>(dA, dB, dC, dD);
  run<2, 2, 0>(dA, dB, dC, dD);
  run<2, 3, 0>(dA, dB, dC, dD);
  run<1, 0, 0>(dA, dB, dC, dD);
  run<1, 1, 0>(dA, dB, dC, dD);
  run<0, 0, 1>(dA, dB, dC, dD);
  run<0, 0, 2>(dA, dB, dC, dD);
  run<0, 0, 4>(dA, dB, dC, dD);
  run<2, 1, 1>(dA, dB, dC, dD);
  return 0;
}
