// A plain C++/HIP consumer of include/kf.h -- no Python, no torch: what a non-Python host (or the
// cgo/JNI-style binding INTEGRATION.md describes) does with the boundary.
//   usage: abi_consumer <libkf_hip.so> <nile.csv>
// Runs the two known answers the reference's own parity inputs give:
//   Nile local level  (Z=T=R=1, H=15099, Q=1469.1, a0=0, P0=1e7): logp = -641.5855784594156
//   Nile local linear trend (tests/utilities/test_helpers.py:110-121): logp = -606737.514026642
// then the smoother on the second (last smoothed == last filtered) and a gradient call whose
// d logp / d H is checked against a central finite difference.  Prints "ABI_CONSUMER_OK".
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/kf.h"

#define CHECK(x)                                                         \
  do {                                                                   \
    if (!(x)) {                                                          \
      fprintf(stderr, "FAILED %s (%s:%d)\n", #x, __FILE__, __LINE__);    \
      return 1;                                                          \
    }                                                                    \
  } while (0)

template <typename F>
static F sym(void* lib, const char* name) {
  void* p = dlsym(lib, name);
  if (!p) {
    fprintf(stderr, "missing symbol %s\n", name);
    exit(2);
  }
  return reinterpret_cast<F>(p);
}

static double* to_dev(const std::vector<double>& v) {
  double* d = nullptr;
  if (hipMalloc(&d, sizeof(double) * (v.empty() ? 1 : v.size())) != hipSuccess) exit(3);
  if (!v.empty()) hipMemcpy(d, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice);
  return d;
}
static std::vector<double> to_host(const double* d, size_t n) {
  std::vector<double> v(n);
  hipMemcpy(v.data(), d, sizeof(double) * n, hipMemcpyDeviceToHost);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  void* lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 2;
  }
  auto create = sym<decltype(&kf_create)>(lib, "kf_create");
  auto destroy = sym<decltype(&kf_destroy)>(lib, "kf_destroy");
  auto last_error = sym<decltype(&kf_last_error)>(lib, "kf_last_error");
  auto forward = sym<decltype(&kf_filter_forward)>(lib, "kf_filter_forward");
  auto backward = sym<decltype(&kf_filter_backward)>(lib, "kf_filter_backward");
  auto smoother = sym<decltype(&kf_smoother)>(lib, "kf_smoother");
  auto smoother_ff = sym<decltype(&kf_smoother_from_filter)>(lib, "kf_smoother_from_filter");
  auto ws_bytes = sym<decltype(&kf_workspace_bytes)>(lib, "kf_workspace_bytes");
  auto backward_tapes = sym<decltype(&kf_filter_backward_tapes)>(lib, "kf_filter_backward_tapes");
  auto set_option = sym<decltype(&kf_set_option)>(lib, "kf_set_option");
  auto get_option = sym<decltype(&kf_get_option)>(lib, "kf_get_option");
  auto abi = sym<decltype(&kf_abi_version)>(lib, "kf_abi_version");
  CHECK(abi() == KF_ABI_VERSION);

  std::vector<double> y;
  {
    FILE* f = fopen(argv[2], "r");
    CHECK(f != nullptr);
    char line[256];
    while (fgets(line, sizeof line, f)) {
      char* end;
      double v = strtod(line, &end);
      if (end != line) y.push_back(v);
    }
    fclose(f);
  }
  const int n = (int)y.size();
  CHECK(n == 100);

  kf_handle* h = nullptr;
  CHECK(create(&h, 0) == KF_OK);
  double* dy = to_dev(y);

  // ---- known answer 1: local level, every filter type -------------------------------------------
  {
    kf_dims dims = {1, n, 1, 1, 1, KF_FILTER_STANDARD, 0};
    double *a0 = to_dev({0.0}), *P0 = to_dev({1e7}), *T = to_dev({1.0}), *Z = to_dev({1.0}), *R = to_dev({1.0}),
           *H = to_dev({15099.0}), *Q = to_dev({1469.1}), *logp = to_dev({0.0});
    for (int ft = KF_FILTER_STANDARD; ft <= KF_FILTER_SINGLE; ++ft) {
      dims.filter = ft;
      int rc = forward(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, logp, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, 0, nullptr);
      if (rc != KF_OK) fprintf(stderr, "%s\n", last_error(h));
      CHECK(rc == KF_OK);
      CHECK(hipDeviceSynchronize() == hipSuccess);
      const double lp = to_host(logp, 1)[0];
      CHECK(fabs(lp - (-641.5855784594156)) < 1e-9);
    }
  }
  // ---- known answer 2: local linear trend, tapes, smoother, gradient ----------------------------------------
  {
    const int m = 2;
    kf_dims dims = {1, n, 1, m, 2, KF_FILTER_STANDARD, 0};
    std::vector<double> Hh = {0.8};
    double *a0 = to_dev({0.0, 0.0}), *P0 = to_dev({1e6, 0.0, 0.0, 1e6}), *T = to_dev({1.0, 1.0, 0.0, 1.0}),
           *Z = to_dev({1.0, 0.0}), *R = to_dev({1.0, 0.0, 0.0, 1.0}), *H = to_dev(Hh),
           *Q = to_dev({0.5, 0.0, 0.0, 0.01});
    double* logp = to_dev({0.0});
    double* ll = to_dev(std::vector<double>(n));
    double* fa = to_dev(std::vector<double>(n * m));
    double* pa = to_dev(std::vector<double>((n + 1) * m));
    double* fP = to_dev(std::vector<double>(n * m * m));
    double* pP = to_dev(std::vector<double>((n + 1) * m * m));
    int32_t* info = nullptr;
    CHECK(hipMalloc(&info, sizeof(int32_t)) == hipSuccess);
    size_t wsb = 0;
    CHECK(ws_bytes(h, &dims, &wsb) == KF_OK);
    void* ws = nullptr;
    if (wsb) CHECK(hipMalloc(&ws, wsb) == hipSuccess);
    int rc = forward(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, logp, ll, fa, pa, fP, pP, info, ws, wsb,
                     nullptr);
    if (rc != KF_OK) fprintf(stderr, "%s\n", last_error(h));
    CHECK(rc == KF_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    const double lp = to_host(logp, 1)[0];
    CHECK(fabs(lp - (-606737.514026642)) < 1e-10 * 606737.5);
    std::vector<double> hfa = to_host(fa, n * m);
    CHECK(fabs(hfa[(n - 1) * m] - 730.6659435031556) < 1e-6 && fabs(hfa[(n - 1) * m + 1] + 17.64853509699647) < 1e-6);
    std::vector<double> hll = to_host(ll, n);
    double s = 0.0;
    for (double v : hll) s += v;
    CHECK(fabs(s - lp) < 1e-6);

    double* sa = to_dev(std::vector<double>(n * m));
    double* sP = to_dev(std::vector<double>(n * m * m));
    CHECK(smoother(h, &dims, T, R, Q, fa, fP, sa, sP, ws, wsb, nullptr) == KF_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    std::vector<double> hsa = to_host(sa, n * m);
    CHECK(hsa[(n - 1) * m] == hfa[(n - 1) * m] && hsa[(n - 1) * m + 1] == hfa[(n - 1) * m + 1]);
    CHECK(fabs(hsa[0] - 1119.9055522554306) < 1e-5);  // smoothed state[0], SURVEY 8(c)
    // the same moments from the filter's predicted tapes (ABI 6): no R, Q, no workspace
    double* sa2 = to_dev(std::vector<double>(n * m));
    double* sP2 = to_dev(std::vector<double>(n * m * m));
    rc = smoother_ff(h, &dims, dy, T, Z, H, nullptr, pa, pP, sa2, sP2, nullptr, 0, nullptr);
    if (rc != KF_OK) fprintf(stderr, "%s\n", last_error(h));
    CHECK(rc == KF_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    std::vector<double> hsa2 = to_host(sa2, n * m);
    CHECK(fabs(hsa2[0] - 1119.9055522554306) < 1e-5);
    for (int e = 0; e < n * m; ++e) CHECK(fabs(hsa2[e] - hsa[e]) < 1e-7 * (1.0 + fabs(hsa[e])));

    double* g1 = to_dev({1.0});
    double* gH = to_dev({0.0});
    rc = backward(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, pa, pP, g1, nullptr, nullptr, nullptr, nullptr,
                  nullptr, nullptr, gH, nullptr, nullptr, nullptr, ws, wsb, nullptr);
    if (rc != KF_OK) fprintf(stderr, "%s\n", last_error(h));
    CHECK(rc == KF_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    const double dH = to_host(gH, 1)[0];
    double lpm[2];
    const double eps = 1e-5;
    for (int k = 0; k < 2; ++k) {
      std::vector<double> Hp = {0.8 + (k ? eps : -eps)};
      hipMemcpy(H, Hp.data(), sizeof(double), hipMemcpyHostToDevice);
      CHECK(forward(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, logp, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr, ws, wsb, nullptr) == KF_OK);
      CHECK(hipDeviceSynchronize() == hipSuccess);
      lpm[k] = to_host(logp, 1)[0];
    }
    const double fd = (lpm[1] - lpm[0]) / (2 * eps);
    CHECK(fabs(fd - dH) < 1e-4 * (1.0 + fabs(fd)));
    // ABI 7: a cotangent on ONE entry of the filtered-state tape -- d filtered_states[n-1][0] / dH against a central difference
    {
      hipMemcpy(H, Hh.data(), sizeof(double), hipMemcpyHostToDevice);
      CHECK(forward(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, logp, nullptr, fa, pa, fP, pP, nullptr, ws, wsb, nullptr) == KF_OK);
      std::vector<double> gfa(n * m, 0.0);
      gfa[(n - 1) * m] = 1.0;
      double* dgfa = to_dev(gfa);
      rc = backward_tapes(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, pa, pP, nullptr, nullptr, dgfa, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, gH, nullptr, nullptr, nullptr, ws, wsb, nullptr);
      if (rc != KF_OK) fprintf(stderr, "%s\n", last_error(h));
      CHECK(rc == KF_OK);
      CHECK(hipDeviceSynchronize() == hipSuccess);
      const double dHt = to_host(gH, 1)[0];
      double fam[2];
      for (int k = 0; k < 2; ++k) {
        std::vector<double> Hp = {0.8 + (k ? eps : -eps)};
        hipMemcpy(H, Hp.data(), sizeof(double), hipMemcpyHostToDevice);
        CHECK(forward(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, logp, nullptr, fa, nullptr, nullptr, nullptr, nullptr, ws, wsb,
                      nullptr) == KF_OK);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        fam[k] = to_host(fa, n * m)[(n - 1) * m];
      }
      const double fdt = (fam[1] - fam[0]) / (2 * eps);
      CHECK(fabs(fdt - dHt) < 1e-5 * (1.0 + fabs(fdt)));
      CHECK(backward_tapes(h, &dims, dy, a0, P0, T, Z, R, H, Q, nullptr, nullptr, pa, pP, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, gH, nullptr, nullptr, nullptr, ws, wsb, nullptr) == KF_ERR_ARG);
    }
    // ABI 7: path switches per handle
    CHECK(set_option(h, "KF_STEADY_STATE", "0") == KF_OK);
    CHECK(get_option(h, "KF_STEADY_STATE") != nullptr && !strcmp(get_option(h, "KF_STEADY_STATE"), "0"));
    CHECK(set_option(h, "KF_STEADY_STATE", nullptr) == KF_OK);
    CHECK(set_option(h, "KF_NOT_A_SWITCH", "1") == KF_ERR_ARG);
    // argument checking without a crash
    CHECK(forward(h, &dims, nullptr, a0, P0, T, Z, R, H, Q, nullptr, nullptr, logp, nullptr, nullptr, nullptr, nullptr,
                  nullptr, nullptr, nullptr, 0, nullptr) == KF_ERR_ARG);
    CHECK(strlen(last_error(h)) > 0);
  }
  destroy(h);
  printf("ABI_CONSUMER_OK\n");
  return 0;
}
