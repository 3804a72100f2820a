// Where does the dispatcher put workgroup i of a 1-wave-per-workgroup launch (10 KB of LDS, <= 128 registers: 16 per CU)?
// Records (XCC, SE, SH, CU, SIMD, wave slot) per workgroup and the order they started in.   hipcc --offload-arch=gfx950 -O2 wg_placement.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void probe(unsigned* out, int spin) {
  extern __shared__ double lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = wall_clock64();
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;   // keep every workgroup resident while the rest are dispatched
  lds[threadIdx.x] = x;
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = hw;
    out[4 * blockIdx.x + 1] = xcc;
    out[4 * blockIdx.x + 2] = (unsigned)(t0 & 0xffffffffu);
    out[4 * blockIdx.x + 3] = (unsigned)lds[0];
  }
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  unsigned* d;
  hipMalloc(&d, 16 * B);
  std::vector<unsigned> h(4 * B);
  hipLaunchKernelGGL(probe, dim3(B), dim3(64), 10240, 0, d, 200000);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(probe, dim3(B), dim3(64), 10240, 0, d, 200000);
  hipMemcpy(h.data(), d, 16 * B, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx90a+: [14:13]), tg_id [19:16], vm_id, queue_id, state_id, me_id
  std::map<unsigned, std::vector<int>> per_simd;
  printf("# wg  xcc se sh cu simd wave  (first 72 workgroups)\n");
  for (int i = 0; i < B; ++i) {
    unsigned hw = h[4 * i], xcc = h[4 * i + 1] & 0xf;
    unsigned wave = hw & 0xf, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    if (i < 72) printf("%4d  %u %u %u %2u %u %u\n", i, xcc, se, sh, cu, simd, wave);
    unsigned key = (xcc << 16) | (se << 12) | (sh << 10) | (cu << 4) | simd;
    per_simd[key].push_back(i);
  }
  printf("# distinct SIMDs: %zu\n", per_simd.size());
  int shown = 0;
  std::map<size_t, int> occ;
  for (auto& kv : per_simd) {
    occ[kv.second.size()]++;
    if (shown++ < 24) {
      printf("simd %06x:", kv.first);
      for (int w : kv.second) printf(" %d", w);
      printf("\n");
    }
  }
  for (auto& kv : occ) printf("# %d SIMDs hold %zu workgroups\n", kv.second, kv.first);
  // does workgroup i share its SIMD with i + k * stride? print the differences within the first SIMDs
  return 0;
}
