// How well do the tape stores of the forward filter overlap with its arithmetic?  4096 wavefronts (one per
// draw, 16 per CU, 10 KiB of LDS each like kf_fwd_b4_kernel), per step NMF v_mfma_f64_4x4x4 + NFMA v_fma_f64 in
// dependent chains, plus a store pattern:
//   0 none | 1 two 800-byte pieces per step (16 B x 50 lanes) | 2 = 1 + two 80-byte pieces (8 B x 10 lanes)
//   3 = 1 with the pieces staged through LDS (3 ds_write_b64 + 1 ds_read_b128 each) | 4 = 2 staged
//   5 two tapes as 1 KiB windows, 25 of 32 steps (the TapeStream pattern) | 6 = 5 + 512-byte windows every 6th step
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int NMF, int NFMA, int MODE, int NLDS = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k(double* f, double* p, double* fa, double* pa, int n) {
  extern __shared__ double smem[];
  const long b = blockIdx.x;
  const int lane = threadIdx.x;
  const int mm = 100, m = 10;
  char* of = reinterpret_cast<char*>(f + b * (long)n * mm);
  char* op = reinterpret_cast<char*>(p + b * (long)n * mm);
  char* oa = reinterpret_cast<char*>(fa + b * (long)n * m);
  char* ob = reinterpret_cast<char*>(pa + b * (long)n * m);
  double acc[3] = {1.0 + lane * 1e-3, 2.0, 3.0}, x = 1.0 + lane * 1e-9, y = 0.5;
  double w = 0.25 + lane;
  long wf = 0, wp = 0;  // bytes streamed (modes 5, 6)
  for (int t = 0; t < n; ++t) {
    // ---- arithmetic: NMF MFMAs in three chains, NFMA FMAs in one chain, the store data depends on them ----
#pragma unroll
    for (int i = 0; i < NMF; ++i) acc[i % 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[i % 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NFMA; ++i) w = fma(w, 0.999999, acc[i % 3] * 1e-30);
    // NLDS transposes through LDS per step (3 ds_write_b64 + 1 ds_read_b128 each, result feeds the chains)
#pragma unroll
    for (int r = 0; r < NLDS; ++r) {
      double* tr = smem + 512;
#pragma unroll
      for (int i = 0; i < 3; ++i) tr[((4 * i + (lane >> 4)) * 18 + (lane & 15))] = acc[i];
      asm volatile("" ::: "memory");
      const double2_t q = *reinterpret_cast<double2_t*>(tr + (lane & 15) * 18 + 4 * (lane >> 4));
      asm volatile("" ::: "memory");
      acc[r % 3] += q.x * 1e-30 + q.y * 1e-31;
    }
    double2_t v = {w, acc[0]};
    if (MODE == 3 || MODE == 4) {
      double* stg = smem;
#pragma unroll
      for (int i = 0; i < 3; ++i) stg[(i * 64 + lane) % 128 + 128 * 0] = acc[i];
      asm volatile("" ::: "memory");
      v = *reinterpret_cast<double2_t*>(stg + 2 * lane);
    }
    if (MODE >= 1 && MODE <= 4) {
      if (lane < 50) *reinterpret_cast<double2_t*>(of + (long)t * 800 + 16 * lane) = v;
    }
    if (MODE == 2 || MODE == 4) {
      if (lane < 10) *reinterpret_cast<double*>(oa + (long)t * 80 + 8 * lane) = v.x;
    }
    if (MODE == 3 || MODE == 4) {
      double* stg = smem + 256;
#pragma unroll
      for (int i = 0; i < 3; ++i) stg[(i * 64 + lane) % 128] = acc[i] + 1.0;
      asm volatile("" ::: "memory");
      v = *reinterpret_cast<double2_t*>(stg + 2 * lane);
    }
    if (MODE >= 1 && MODE <= 4) {
      if (lane < 50) *reinterpret_cast<double2_t*>(op + (long)t * 800 + 16 * lane) = v;
    }
    if (MODE == 2 || MODE == 4) {
      if (lane < 10) *reinterpret_cast<double*>(ob + (long)t * 80 + 8 * lane) = v.y;
    }
    if (MODE == 7 || MODE == 8) {  // straight from the accumulator registers: 3 x (4 rows x 80 B), 8 B per lane
      const bool on = (lane & 15) < 10;
      const unsigned lo = ((lane >> 4) * 10 + (lane & 15)) * 8;
      if (on) *reinterpret_cast<double*>(of + (long)t * 800 + lo) = acc[0];
      if (on) *reinterpret_cast<double*>(of + (long)t * 800 + 320 + lo) = acc[1];
      if (on && lane < 32) *reinterpret_cast<double*>(of + (long)t * 800 + 640 + lo) = acc[2];
      if (on) *reinterpret_cast<double*>(op + (long)t * 800 + lo) = acc[0] + 1.0;
      if (on) *reinterpret_cast<double*>(op + (long)t * 800 + 320 + lo) = acc[1] + 1.0;
      if (on && lane < 32) *reinterpret_cast<double*>(op + (long)t * 800 + 640 + lo) = acc[2] + 1.0;
      if (MODE == 8 && lane < 10) {
        *reinterpret_cast<double*>(oa + (long)t * 80 + 8 * lane) = v.x;
        *reinterpret_cast<double*>(ob + (long)t * 80 + 8 * lane) = v.y;
      }
    }
    if (MODE == 5 || MODE == 6) {
      if ((t & 31) % 5 != 4 || (t & 31) == 4) {  // 25 or 26 of 32 steps; bytes: ~ 800 per step
        if (wf + 1024 <= (long)n * 800) {
          *reinterpret_cast<double2_t*>(of + wf + 16 * lane) = v;
          *reinterpret_cast<double2_t*>(op + wp + 16 * lane) = v;
          wf += 1024;
          wp += 1024;
        }
      }
      if (MODE == 6 && t % 6 == 5 && lane < 30) {
        *reinterpret_cast<double2_t*>(oa + (long)(t - 5) * 80 + 16 * lane) = v;
        *reinterpret_cast<double2_t*>(ob + (long)(t - 5) * 80 + 16 * lane) = v;
      }
    }
  }
  if (lane == 0) f[b * (long)n * mm] = acc[0] + acc[1] + acc[2] + w;
}

template <int NMF, int NFMA, int MODE, int NLDS = 0>
void run(double* f, double* p, double* fa, double* pa, int B, int n, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = k<NMF, NFMA, MODE, NLDS>;
  hipLaunchKernelGGL(kern, dim3(B), dim3(64), 10240, 0, f, p, fa, pa, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(B), dim3(64), 10240, 0, f, p, fa, pa, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("MFMA %2d FMA %2d LDS-transposes %d  %-58s %.3f ms\n", NMF, NFMA, NLDS, what, ms / 5);
}

template <int NMF, int NFMA>
void suite(double* f, double* p, double* fa, double* pa, int B, int n) {
  run<NMF, NFMA, 0>(f, p, fa, pa, B, n, "no stores");
  run<NMF, NFMA, 1>(f, p, fa, pa, B, n, "2 x 800 B per step");
  run<NMF, NFMA, 2>(f, p, fa, pa, B, n, "2 x 800 B + 2 x 80 B per step");
  run<NMF, NFMA, 3>(f, p, fa, pa, B, n, "2 x 800 B per step, staged through LDS");
  run<NMF, NFMA, 4>(f, p, fa, pa, B, n, "2 x 800 B + 2 x 80 B per step, staged");
  run<NMF, NFMA, 5>(f, p, fa, pa, B, n, "2 x 1 KiB windows (25 of 32 steps)");
  run<NMF, NFMA, 6>(f, p, fa, pa, B, n, "2 x 1 KiB windows + 2 x 480 B every 6th step");
  run<NMF, NFMA, 7>(f, p, fa, pa, B, n, "2 x (320 + 320 + 160 B) from registers, 8 B per lane");
  run<NMF, NFMA, 8>(f, p, fa, pa, B, n, "2 x (320 + 320 + 160 B) + 2 x 80 B from registers");
}

int main() {
  const int B = 4096, n = 1000;
  double *f, *p, *fa, *pa;
  hipMalloc(&f, sizeof(double) * (size_t)B * n * 100);
  hipMalloc(&p, sizeof(double) * (size_t)B * n * 100);
  hipMalloc(&fa, sizeof(double) * (size_t)B * n * 10);
  hipMalloc(&pa, sizeof(double) * (size_t)B * n * 10);
  run<21, 40, 0, 0>(f, p, fa, pa, B, n, "no stores");
  run<21, 40, 0, 3>(f, p, fa, pa, B, n, "no stores");
  run<21, 40, 2, 0>(f, p, fa, pa, B, n, "2 x 800 B + 2 x 80 B per step");
  run<21, 40, 2, 3>(f, p, fa, pa, B, n, "2 x 800 B + 2 x 80 B per step");
  run<21, 40, 8, 0>(f, p, fa, pa, B, n, "registers: 2 x (320 + 320 + 160 B) + 2 x 80 B");
  run<21, 40, 8, 3>(f, p, fa, pa, B, n, "registers: 2 x (320 + 320 + 160 B) + 2 x 80 B");
  run<21, 40, 6, 3>(f, p, fa, pa, B, n, "2 x 1 KiB windows + 2 x 480 B every 6th step");
  run<21, 40, 1, 3>(f, p, fa, pa, B, n, "2 x 800 B per step");
  run<21, 40, 8, 6>(f, p, fa, pa, B, n, "registers: 2 x (320 + 320 + 160 B) + 2 x 80 B");
  return 0;
}
