// How fast can 4096 wavefronts stream the headline tapes (2 x 800 B per step, 1000 steps) with no compute?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int CHUNK>  // steps gathered per store burst
__global__ __launch_bounds__(64) void k(double* f, double* p, int n, int mm) {
  const long b = blockIdx.x;
  const int lane = threadIdx.x;
  double* of = f + b * (long)n * mm;
  double* op = p + b * (long)(n + 1) * mm;
  const int per = CHUNK * mm / 2;  // double2 per burst
  double2_t v = {1.0 * lane, 2.0};
  for (int t = 0; t < n; t += CHUNK) {
    for (int e = lane; e < per; e += 64) {
      reinterpret_cast<double2_t*>(of + (long)t * mm)[e] = v;
      reinterpret_cast<double2_t*>(op + (long)t * mm)[e] = v;
    }
    v.x += 1.0;
    __builtin_amdgcn_s_sleep(CHUNK == 1 ? 4 : 4 * CHUNK);  // crude pacing: ~ compute time between bursts
  }
}
template <int CHUNK>
void run(double* f, double* p, int B, int n, int mm) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHUNK>, dim3(B), dim3(64), 0, 0, f, p, n, mm);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<CHUNK>, dim3(B), dim3(64), 0, 0, f, p, n, mm);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double gb = 2.0 * B * (double)n * mm * 8 / 1e9;
  printf("chunk %2d steps: %.3f ms per launch, %.0f GB/s\n", CHUNK, ms / 5, gb / (ms / 5) * 1e3);
}
int main() {
  const int B = 4096, n = 1000, mm = 100;
  double *f, *p;
  hipMalloc(&f, sizeof(double) * (size_t)B * n * mm);
  hipMalloc(&p, sizeof(double) * (size_t)B * (n + 1) * mm);
  run<1>(f, p, B, n, mm);
  run<2>(f, p, B, n, mm);
  run<4>(f, p, B, n, mm);
  run<8>(f, p, B, n, mm);
  return 0;
}
