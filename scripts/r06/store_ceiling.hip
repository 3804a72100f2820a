// What the memory side gives the headline's WRITE PATTERN, with nothing to compute: 4096 wavefronts (one per draw, four per SIMD),
// each writing its four output tapes front to back the way kf_fwd_b4_kernel does -- covariance tapes as whole 128-byte lines, 16 bytes
// per lane, ~1.6 KB every second step, alternating between the two tapes; state tapes as 512-byte windows every 6.4 steps -- plus
// variants: everything in 4 KB bursts; one tape per wavefront-loop.  Build: hipcc --offload-arch=gfx950 -O3 store_ceiling.hip -o store_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double2_t __attribute__((ext_vector_type(2)));

// mode 0: the kernel's pattern.  mode 1: each tape in bursts of `burst` doubles (whole lines), round-robin over the four tapes.
template <bool NT>
__device__ __forceinline__ void st16(double* p, double2_t v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<double2_t*>(p));
  else *reinterpret_cast<double2_t*>(p) = v;
}

// mode 0 with switches: NT non-temporal stores; states: the two state tapes too; each workgroup does `per` draws one after the other
template <bool NT>
__global__ __launch_bounds__(64) void writer2(double* fa, double* pa, double* fP, double* pP, int n, int m, int states, int per) {
  const int lane = threadIdx.x;
  const int mm = m * m;
  const double2_t v = {1.0 + lane, 2.0};
  for (int k = 0; k < per; ++k) {
    const long b = (long)blockIdx.x * per + k;
    double* t_fa = fa + b * (long)n * m;
    double* t_pa = pa + b * (long)(n + 1) * m;
    double* t_fP = fP + b * (long)n * mm;
    double* t_pP = pP + b * (long)(n + 1) * mm;
    t_fP -= (reinterpret_cast<unsigned long long>(t_fP) >> 3) & 15; t_pP -= (reinterpret_cast<unsigned long long>(t_pP) >> 3) & 15;
    t_fa -= (reinterpret_cast<unsigned long long>(t_fa) >> 3) & 15; t_pa -= (reinterpret_cast<unsigned long long>(t_pa) >> 3) & 15;
    long ef = 0, ep = 0, eaf = 0, eap = 0;
    for (int t = 0; t < n; ++t) {
      if (t & 1) {
        const long hi = ((long)(t + 1) * mm) & ~15L;
        for (long e = ef + 2 * lane; e < hi; e += 128) st16<NT>(t_fP + e, v);
        ef = hi;
      } else {
        const long hi = ((long)(t + 2) * mm) & ~15L;
        for (long e = ep + 2 * lane; e < hi; e += 128) st16<NT>(t_pP + e, v);
        ep = hi;
      }
      if (states) {
        const long ha = ((long)(t + 1) * m) & ~63L;
        if (ha > eaf) {
          if (lane < 32) st16<NT>(t_fa + eaf + 2 * lane, v);
          eaf = ha;
        }
        const long hp = ((long)(t + 2) * m) & ~63L;
        if (hp > eap) {
          if (lane < 32) st16<NT>(t_pa + eap + 2 * lane, v);
          eap = hp;
        }
      }
    }
  }
}

__global__ __launch_bounds__(64) void writer(double* fa, double* pa, double* fP, double* pP, int n, int m, int mode, int burst, int spin) {
  const long b = blockIdx.x;
  const int lane = threadIdx.x;
  const int mm = m * m;
  double* t_fa = fa + b * (long)n * m;
  double* t_pa = pa + b * (long)(n + 1) * m;
  double* t_fP = fP + b * (long)n * mm;
  double* t_pP = pP + b * (long)(n + 1) * mm;
  const double2_t v = {1.0 + lane, 2.0};
  double acc = lane;
  if (mode == 0) {
    // (whole lines by ABSOLUTE address: a tape that starts inside a line is shifted to the line's start, as the kernel's rings do)
    t_fP -= (reinterpret_cast<unsigned long long>(t_fP) >> 3) & 15; t_pP -= (reinterpret_cast<unsigned long long>(t_pP) >> 3) & 15;
    t_fa -= (reinterpret_cast<unsigned long long>(t_fa) >> 3) & 15; t_pa -= (reinterpret_cast<unsigned long long>(t_pa) >> 3) & 15;
    long ef = 0, ep = 0, eaf = 0, eap = 0;   // elements stored so far
    for (int t = 0; t < n; ++t) {
      for (int s = 0; s < spin; ++s) acc = __builtin_fma(acc, 1.0000001, 1e-9);   // stand-in for the step's arithmetic
      // covariance tapes: whole lines of what has been appended, every second step
      if (t & 1) {
        const long hi = ((long)(t + 1) * mm) & ~15L;
        for (long e = ef + 2 * lane; e < hi; e += 128) *reinterpret_cast<double2_t*>(t_fP + e) = v;
        ef = hi;
      } else {
        const long hi = ((long)(t + 2) * mm) & ~15L;
        for (long e = ep + 2 * lane; e < hi; e += 128) *reinterpret_cast<double2_t*>(t_pP + e) = v;
        ep = hi;
      }
      const long ha = ((long)(t + 1) * m) & ~63L;
      if (ha > eaf) {
        if (lane < 32) *reinterpret_cast<double2_t*>(t_fa + eaf + 2 * lane) = v;
        eaf = ha;
      }
      const long hp = ((long)(t + 2) * m) & ~63L;
      if (hp > eap) {
        if (lane < 32) *reinterpret_cast<double2_t*>(t_pa + eap + 2 * lane) = v;
        eap = hp;
      }
    }
  } else {
    const long tot_P = (long)n * mm, tot_a = (long)n * m;
    for (long e0 = 0; e0 < tot_P; e0 += burst) {
      for (int s = 0; s < spin; ++s) acc = __builtin_fma(acc, 1.0000001, 1e-9);
      const long hi = e0 + burst < tot_P ? e0 + burst : tot_P;
      for (long e = e0 + 2 * lane; e < hi; e += 128) *reinterpret_cast<double2_t*>(t_fP + e) = v;
      for (long e = e0 + 2 * lane; e < hi; e += 128) *reinterpret_cast<double2_t*>(t_pP + e) = v;
      const long a0 = e0 / m, a1 = hi / m < tot_a ? hi / m : tot_a;
      for (long e = (a0 & ~1L) + 2 * lane; e < a1; e += 128) {
        *reinterpret_cast<double2_t*>(t_fa + e) = v;
        *reinterpret_cast<double2_t*>(t_pa + e) = v;
      }
    }
  }
  if (acc == 12345.678) fa[0] = acc;
}

int main() {
  const int B = 4096, n = 1000, m = 10, mm = 100;
  double *fa, *pa, *fP, *pP;
  hipMalloc(&fa, sizeof(double) * (size_t)B * n * m + 4096);
  hipMalloc(&pa, sizeof(double) * (size_t)B * (n + 1) * m + 4096);
  hipMalloc(&fP, sizeof(double) * (size_t)B * n * mm + 4096);
  hipMalloc(&pP, sizeof(double) * (size_t)B * (n + 1) * mm + 4096);
  const double gb = 8.0 * B * ((double)n * (2 * m + 2 * mm) + m + mm) / 1e9;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct Cfg { int mode, burst, spin; const char* name; };
  std::vector<Cfg> cfgs = {{0, 0, 0, "kernel pattern, no arithmetic"}, {0, 0, 100, "kernel pattern, 100 dependent fma per step"},
                           {0, 0, 250, "kernel pattern, 250 dependent fma per step"}, {0, 0, 400, "kernel pattern, 400 dependent fma per step"},
                           {1, 512, 0, "4 KB bursts"}, {1, 2048, 0, "16 KB bursts"}, {1, 200, 0, "1.6 KB bursts, both cov tapes together"}};
  for (const Cfg& c : cfgs) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(writer, dim3(B), dim3(64), 0, 0, fa, pa, fP, pP, n, m, c.mode, c.burst, c.spin);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(writer, dim3(B), dim3(64), 0, 0, fa, pa, fP, pP, n, m, c.mode, c.burst, c.spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float msv = 0;
    hipEventElapsedTime(&msv, e0, e1);
    msv /= reps;
    printf("%-52s %.3f ms  %.2f TB/s\n", c.name, msv, gb / msv);
  }
  auto run2 = [&](const char* name, bool nt, int states, int per) {
    auto go = [&]() {
      if (nt) hipLaunchKernelGGL(writer2<true>, dim3(B / per), dim3(64), 0, 0, fa, pa, fP, pP, n, m, states, per);
      else hipLaunchKernelGGL(writer2<false>, dim3(B / per), dim3(64), 0, 0, fa, pa, fP, pP, n, m, states, per);
    };
    for (int w = 0; w < 3; ++w) go();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float msv = 0;
    hipEventElapsedTime(&msv, e0, e1);
    msv /= 10;
    const double g2 = states ? gb : 8.0 * B * ((double)n * 2 * mm + mm) / 1e9;
    printf("%-52s %.3f ms  %.2f TB/s\n", name, msv, g2 / msv);
  };
  run2("pattern again (writer2)", false, 1, 1);
  run2("pattern, non-temporal stores", true, 1, 1);
  run2("covariance tapes only", false, 0, 1);
  run2("covariance tapes only, non-temporal", true, 0, 1);
  run2("2048 workgroups, two draws each", false, 1, 2);
  run2("1024 workgroups, four draws each", false, 1, 4);
  run2("1024 workgroups, four draws each, non-temporal", true, 1, 4);
  return 0;
}
