// Does an LDS instruction cost less when fewer lanes are active?  16 one-wave workgroups per CU (4 per SIMD), each issuing a long
// stream of ds_read_b128 / ds_write_b64 with (a) all 64 lanes, (b) one DPP row (16 lanes), (c) one lane active; time per instruction.
//   hipcc --offload-arch=gfx950 -O2 lds_mask_probe.hip -o lds_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int MODE, int OP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void probe(double* out, int iters) {
  __shared__ __attribute__((aligned(16))) double lds[1280];
  const int lane = threadIdx.x;
  for (int e = lane; e < 1280; e += 64) lds[e] = e;
  __syncthreads();
  const bool on = MODE == 0 ? true : (MODE == 1 ? (lane >> 4) == 2 : lane == 7);
  double2_t acc = {0.0, 0.0};
  const double2_t* src = reinterpret_cast<const double2_t*>(lds) + lane;     // 16 bytes per lane, consecutive: conflict-free
  double* dst = lds + lane;
  for (int i = 0; i < iters; ++i) {
    if (on) {
      if (OP == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          double2_t v = src[64 * (k & 3)];
          asm volatile("" : "+v"(v));
          acc += v;
        }
      } else if (OP == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          dst[64 * k] = acc.x + k;
          asm volatile("" ::: "memory");
        }
      } else if (OP == 2) {  // ds_write_b128
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          double2_t v = {acc.x + k, acc.y};
          reinterpret_cast<double2_t*>(lds)[lane + 64 * (k & 3)] = v;
          asm volatile("" ::: "memory");
        }
      } else if (OP == 3) {  // ds_read_b64
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          double v = lds[lane + 64 * k];
          asm volatile("" : "+v"(v));
          acc.x += v;
        }
      } else if (OP == 4) {  // two b64 writes the compiler may fuse into ds_write2_b64 (stride 72 doubles)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dst[8 * k] = acc.x + k;
          dst[8 * k + 72 * 8] = acc.y + k;
          asm volatile("" ::: "memory");
        }
      }
    }
  }
  if (acc.x == 12345.678) out[lane] = acc.y + lds[lane];
}
template <int MODE, int OP>
static void run(const char* name) {
  double* d;
  hipMalloc(&d, 4096);
  const int iters = 20000, grid = 256 * 16;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<MODE, OP>), dim3(grid), dim3(64), 0, 0, d, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, OP>), dim3(grid), dim3(64), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 16 waves x iters x 8 instructions
  const double ns_per_inst_per_cu = ms * 1e6 / (16.0 * iters * 8);
  printf("%-34s %8.3f ms  -> %6.2f ns per wave-instruction on a CU's LDS (16 waves/CU)\n", name, ms, ns_per_inst_per_cu);
  hipFree(d);
}
int main() {
  run<0, 0>("ds_read_b128, 64 lanes");
  run<1, 0>("ds_read_b128, 16 lanes (one row)");
  run<2, 0>("ds_read_b128, 1 lane");
  run<0, 1>("ds_write_b64, 64 lanes");
  run<1, 1>("ds_write_b64, 16 lanes (one row)");
  run<2, 1>("ds_write_b64, 1 lane");
  run<0, 2>("ds_write_b128, 64 lanes");
  run<0, 3>("ds_read_b64, 64 lanes");
  run<0, 4>("ds_write2_b64 (4 per iteration, x2)");
  return 0;
}
