// kf_api.hip -- the extern "C" surface declared in include/kf.h: argument checks,
// kernel selection, launches.  No allocation, no synchronisation, no host copies here
// (safe to capture in a hipGraph).
#include <stdlib.h>

#include "kf_internal.h"
#include "kf_b4_split.h"

static thread_local char g_err[256] = "";

extern "C" int kf_abi_version(void) { return KF_ABI_VERSION; }

extern "C" int kf_create(kf_handle** out, int device) {
  if (!out) return KF_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    snprintf(g_err, sizeof(g_err), "kf_create: no HIP device (%s)", hipGetErrorString(e));
    return KF_ERR_HIP;
  }
  if (device < 0 || device >= ndev) {
    snprintf(g_err, sizeof(g_err), "kf_create: device %d out of range [0,%d)", device, ndev);
    return KF_ERR_ARG;
  }
  kf_handle* h = (kf_handle*)calloc(1, sizeof(kf_handle));
  if (!h) return KF_ERR_ARG;
  h->device = device;
  h->lanes_per_draw = 0;
  h->forward_kernel = KF_KERNEL_AUTO;
  const char* envk = kf_env_switch("KF_FORWARD_KERNEL");
  if (envk && *envk) {
    if (!strcmp(envk, "auto")) h->forward_kernel = KF_KERNEL_AUTO;
    else if (!strcmp(envk, "lds")) h->forward_kernel = KF_KERNEL_LDS;
    else if (!strcmp(envk, "row")) h->forward_kernel = KF_KERNEL_ROW;
    else if (!strcmp(envk, "lane")) h->forward_kernel = KF_KERNEL_LANE;
    else if (!strcmp(envk, "mfma")) h->forward_kernel = KF_KERNEL_MFMA;
    else if (!strcmp(envk, "b4")) h->forward_kernel = KF_KERNEL_B4;
    else {
      snprintf(g_err, sizeof(g_err), "kf_create: KF_FORWARD_KERNEL=%s is not one of auto|lds|row|lane|mfma|b4", envk);
      free(h);
      return KF_ERR_ARG;
    }
  }
  const char* env = kf_env_switch("KF_LANES_PER_DRAW");
  if (env) h->lanes_per_draw = atoi(env);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
  else h->n_cu = 256;
  h->err[0] = 0;
  *out = h;
  return KF_OK;
}

extern "C" void kf_destroy(kf_handle* h) {
  if (h && h->aux_ready) {
    KfDeviceGuard guard(h);
    (void)hipStreamSynchronize(h->aux_stream);
    (void)hipStreamSynchronize(h->aux_stream2);
    (void)hipStreamSynchronize(h->aux_stream3);
    for (int i = 0; i < KF_AUX_EVENTS; ++i) (void)hipEventDestroy(h->aux_ev[i]);
    (void)hipStreamDestroy(h->aux_stream);
    (void)hipStreamDestroy(h->aux_stream2);
    (void)hipStreamDestroy(h->aux_stream3);
  }
  free(h);
}

int kf_handle_aux(kf_handle* h) {
  if (h->aux_ready) return KF_OK;
  int lo = 0, hi = 0;
  KF_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = the numerically largest = lowest priority
  KF_HIP_CHECK(h, hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, lo));
  KF_HIP_CHECK(h, hipStreamCreateWithPriority(&h->aux_stream2, hipStreamNonBlocking, lo));
  KF_HIP_CHECK(h, hipStreamCreateWithPriority(&h->aux_stream3, hipStreamNonBlocking, hi));
  for (int i = 0; i < KF_AUX_EVENTS; ++i) KF_HIP_CHECK(h, hipEventCreateWithFlags(&h->aux_ev[i], hipEventDisableTiming));
  h->aux_ready = 1;
  return KF_OK;
}

extern "C" const char* kf_last_error(const kf_handle* h) { return h ? h->err : g_err; }

thread_local const kf_handle* kf_tls_handle = nullptr;

// every switch the library reads (kf_env_switch call sites; tests/test_abi.py checks this table against the sources)
static const char* const kf_option_names[] = {
    "KF_B4C_STARTS_LDS",
    "KF_B4Q_GROUP",
    "KF_B4Q_SPLIT",
    "KF_B4_SPLIT",
    "KF_BACKWARD_B4",
    "KF_BACKWARD_TV_MFMA",
    "KF_BIG_FUSE",
    "KF_BIG_SPARSE",
    "KF_BIG_TMASK",
    "KF_FORWARD_B4P",
    "KF_FORWARD_B4Q",
    "KF_FORWARD_B4TV",
    "KF_FORWARD_CHUNKED",
    "KF_FORWARD_KERNEL",
    "KF_FORWARD_WGL",
    "KF_FORWARD_WHITEN",
    "KF_LANES_PER_DRAW",
    "KF_LANE_PRECHUNK",
    "KF_SMOOTHER_B4",
    "KF_SMOOTHER_BF_LANE",
    "KF_SMOOTHER_BF_WG",
    "KF_SMOOTHER_BIG_GEMM",
    "KF_SMOOTHER_LANE",
    "KF_SMOOTHER_SPLIT",
    "KF_SMOOTHER_STEADY",
    "KF_SMOOTHER_TV_MFMA",
    "KF_STEADY_STATE",
    "KF_VG_SCHED_HINT",
    "KF_VG_STEADY",
    "KF_VG_STEADY_SCAN",
    "KF_VG_TILE_TAPE"};

const char* kf_handle_option(const kf_handle* h, const char* name) {
  for (int i = 0; i < h->n_options; ++i)
    if (!strcmp(h->options[i].name, name)) return h->options[i].value;
  return nullptr;
}

extern "C" int kf_set_option(kf_handle* h, const char* name, const char* value) {
  if (!h) return KF_ERR_ARG;
  if (!name) {
    KF_SET_ERR(h, "kf_set_option: name is NULL");
    return KF_ERR_ARG;
  }
  bool known = false;
  for (const char* nm : kf_option_names) known = known || !strcmp(nm, name);
  if (!known) {
    KF_SET_ERR(h, "kf_set_option: '%s' is not a switch of this library (INTEGRATION.md lists them)", name);
    return KF_ERR_ARG;
  }
  if (value && strlen(value) >= sizeof(h->options[0].value)) {
    KF_SET_ERR(h, "kf_set_option: value of %s is longer than %d characters", name, (int)sizeof(h->options[0].value) - 1);
    return KF_ERR_ARG;
  }
  int at = -1;
  for (int i = 0; i < h->n_options; ++i)
    if (!strcmp(h->options[i].name, name)) at = i;
  static const struct { const char* s; int k; } fwd_kernels[] = {{"auto", KF_KERNEL_AUTO}, {"lds", KF_KERNEL_LDS}, {"row", KF_KERNEL_ROW},
                                                                {"lane", KF_KERNEL_LANE}, {"mfma", KF_KERNEL_MFMA}, {"b4", KF_KERNEL_B4}};
  if (!value) {   // back to the environment
    if (at >= 0) h->options[at] = h->options[--h->n_options];
    if (!strcmp(name, "KF_FORWARD_KERNEL")) {   // (the selector kf_create would have chosen)
      h->forward_kernel = KF_KERNEL_AUTO;
      const char* e = getenv(name);
      for (const auto& k : fwd_kernels)
        if (e && !strcmp(e, k.s)) h->forward_kernel = k.k;
    }
    return KF_OK;
  }
  if (at < 0) {
    if (h->n_options == KF_MAX_OPTIONS) {
      KF_SET_ERR(h, "kf_set_option: more than %d options on one handle", KF_MAX_OPTIONS);
      return KF_ERR_ARG;
    }
    at = h->n_options++;
    snprintf(h->options[at].name, sizeof(h->options[at].name), "%s", name);
  }
  snprintf(h->options[at].value, sizeof(h->options[at].value), "%s", value);
  if (!strcmp(name, "KF_FORWARD_KERNEL")) {   // (read once, at kf_create: the handle's selector is what it sets)
    for (const auto& e : fwd_kernels)
      if (!strcmp(value, e.s)) h->forward_kernel = e.k;
  }
  return KF_OK;
}

extern "C" const char* kf_get_option(const kf_handle* h, const char* name) {
  if (!h || !name) return nullptr;
  const char* o = kf_handle_option(h, name);
  if (o) return *o ? o : nullptr;
  const char* e = getenv(name);
  return (e && *e) ? e : nullptr;
}

extern "C" int kf_set_lanes_per_draw(kf_handle* h, int lanes) {
  if (!h) return KF_ERR_ARG;
  if (lanes < 0 || lanes > 64) {
    KF_SET_ERR(h, "lanes_per_draw must be in [0, 64] (got %d)", lanes);
    return KF_ERR_ARG;
  }
  h->lanes_per_draw = lanes;
  return KF_OK;
}

extern "C" int kf_set_forward_kernel(kf_handle* h, int kernel) {
  if (!h) return KF_ERR_ARG;
  if (kernel != KF_KERNEL_AUTO && kernel != KF_KERNEL_LDS && kernel != KF_KERNEL_ROW &&
      kernel != KF_KERNEL_LANE && kernel != KF_KERNEL_MFMA && kernel != KF_KERNEL_B4) {
    KF_SET_ERR(h, "unknown forward kernel %d", kernel);
    return KF_ERR_ARG;
  }
  h->forward_kernel = kernel;
  return KF_OK;
}

// lanes per row for the row-split kernels: split the O(m^3) products over more lanes while
// the batch is too small to give every SIMD two wavefronts.
static int kf_pick_row_split(const kf_handle* h, int m, long B) {
  // Measured on MI355X (m = 10): with one row-owner wave per SIMD or more (B*m/64 waves >=
  // #SIMDs) the plain row kernel has the shortest instruction stream per draw; below that,
  // H = 3 lanes per row (two waves per SIMD at B = 4096) is ~1.4x faster.
  const long n_simd = (h ? h->n_cu : 256) * 4L;
  const long waves_h1 = (B * m + KF_WAVE - 1) / KF_WAVE;
  if (waves_h1 >= n_simd) return 1;
  switch (m) {
    case 10: return 3;
    case 8: return 2;
    case 6: return 3;
    case 4: return 2;
    default: return 1;
  }
}

static int kf_split_for(const kf_handle* h, int m, long B) {
  int H = 0;
  if (h->lanes_per_draw > 0 && h->lanes_per_draw % m == 0) H = h->lanes_per_draw / m;
  if (H == 0) H = kf_pick_row_split(h, m, B);
  const bool have = (m == 10 && (H == 2 || H == 3 || H == 5)) || (m == 8 && (H == 2 || H == 4)) ||
                    (m == 6 && H == 3) || (m == 4 && H == 2);
  return (m % 2 == 0 && have) ? H : 1;
}

// scalar-observation kernels (multivariate row kernel, MFMA tile kernel): eligible shapes
static bool kf_mp_ok(const kf_handle* h, int m, int p, int T, int filter, unsigned flags) {
  (void)T;
  if (flags & KF_FLAG_TV_ANY) return false;
  if (h->forward_kernel != KF_KERNEL_AUTO && h->forward_kernel != KF_KERNEL_ROW &&
      h->forward_kernel != KF_KERNEL_MFMA)
    return false;
  if (m > KF_MP_M_MAX || p > KF_MP_P_MAX || p < 1) return false;
  return p == 1 || filter == KF_FILTER_UNIVARIATE || (flags & KF_FLAG_H_DIAGONAL);
}

// the multivariate row kernel addresses its tapes with 31-bit offsets per block
// block-tile MFMA kernel: 16 < m <= 48, static matrices, scalar observations
static bool kf_nb_ok(const kf_handle* h, int m, int p, int r, int filter, unsigned flags) {
  if (flags & KF_FLAG_TV_ANY) return false;
  if (h->forward_kernel != KF_KERNEL_AUTO && h->forward_kernel != KF_KERNEL_MFMA) return false;
  if (!kf_forward_mfma_nb_ok(m, p, r)) return false;
  return p == 1 || filter == KF_FILTER_UNIVARIATE || (flags & KF_FLAG_H_DIAGONAL);
}

// workgroup-per-draw MFMA kernel: 48 < m <= 64, same eligibility
static bool kf_wg_ok(const kf_handle* h, int m, int p, int r, int filter, unsigned flags) {
  if (flags & KF_FLAG_TV_ANY) return false;
  if (h->forward_kernel != KF_KERNEL_AUTO && h->forward_kernel != KF_KERNEL_MFMA) return false;
  if (!kf_forward_mfma_wg_ok(m, p, r)) return false;
  return p == 1 || filter == KF_FILTER_UNIVARIATE || (flags & KF_FLAG_H_DIAGONAL);
}

// 4x4x4-block kernel (kf_forward_b4.hip): forced by KF_KERNEL_B4.  AUTO, measured on MI355X with 1000 steps
// (scripts/r02/kern_sweep.py; full tapes ; logp only, ms):
//   4096 draws  m = 10: 1.94 ; 1.12 against 2.10 ; 1.53 (16 x 16 tile kernel)   m = 14: 3.05 ; 1.68 against 3.29 ; 1.78
//               m = 6: 1.13 ; 0.75 against 1.26 ; 1.04 (row-split)   m = 4: 1.08 ; 0.76 against 1.62 ; 0.95 (lane)
//               m = 2: 0.77 ; 0.57 against 0.87 ; 0.50   m = 1: 0.76 ; 0.57 against 0.67 ; 0.36
//   512 draws   m <= 6: 0.43 - 0.64 ; 0.36 - 0.42 against 0.66 - 1.60 ; 0.36 - 1.15
//   32768 draws x 200 steps, m <= 8: the lane / row kernels win (64 / m draws per wavefront)
// -> one wavefront per draw pays as soon as the batch cannot fill the chip with packed wavefronts
static bool kf_b4_auto(const kf_handle* h, int m, long B) {
  // (m = 9, re-measured with the row kernel at 16384 / 65536 draws x 500 steps: 3.4 / 12.7 ms against 5.1 / 17.0 with every step,
  // 2.7 / 9.8 with the steady-state phase; logp only 2.1 / 7.5 against 2.6 / 7.5 -- scripts/r02/b4_auto_sweep.py)
  if (m >= 9) return true;
  const long n_simd = (h ? h->n_cu : 256) * 4L;
  if (m >= 7) return (B * m + KF_WAVE - 1) / KF_WAVE < 2 * n_simd;  // else the row kernel (as before)
  // (round 4, scripts/r04/fwd_dispatch_sweep.py, full tapes: past one round of wavefront slots the block kernel's time doubles while
  // the packed row kernel's barely moves -- m = 4 / 5 / 6 at 6144 draws x 300 steps 0.53 / 0.56 / 0.58 ms against 0.29 / 0.43 / 0.45, at
  // 8192 x 1000 m = 4: 2.00 against 0.96; log-likelihood only it is a tie at m = 5, 6 up to 8192 draws)
  if (m >= 4) return B <= 16L * (h ? h->n_cu : 256);
  // m <= 3: the lane kernel (4 .. 64 draws per wavefront, LDS-staged tapes; scripts/r02/lane_time.py, full ; logp ms):
  //   m = 2: 0.46 / 0.51 / 0.52 ; 0.33 - 0.35 at 512 / 2048 / 4096 draws against 0.47 / 0.54 / 0.71 ; 0.32 - 0.50 here
  //   m = 3: 0.76 / 0.84 / 0.92 against 0.58 / 0.69 / 0.98
  return m == 3 && B <= 2048;
}
// m = 4 past one round of wavefront slots: the packed row kernel, not the lane kernel (AUTO only; full tapes at 16384 / 32768 / 65536
// draws x 300 steps: 0.46 / 0.75 / 1.70 ms against 0.64 / 1.19 / 2.54, log-likelihood only 0.35 / 0.43 / 0.70 against 0.36 / 0.50 / 0.98)
static bool kf_row_over_lane(const kf_handle* h, int m, long B) {
  return h->forward_kernel == KF_KERNEL_AUTO && m == 4 && B > 16L * (h->n_cu > 0 ? h->n_cu : 256);
}
static bool kf_b4_selected(const kf_handle* h, int m, int p, int r, long B) {
  if (!kf_forward_b4_ok(m, p, r)) return false;
  if (h->forward_kernel == KF_KERNEL_B4) return true;
  return h->forward_kernel == KF_KERNEL_AUTO && kf_b4_auto(h, m, B);
}

// p > 1 scalar observations (the Univariate filter, or a diagonal H) on the block tiles: kf_forward_b4q.hip.
// Measured (scripts/r02/b4q_sweep.py, 500 steps, full tapes ; logp only, ms) against what AUTO picked before:
//   (p, m) = (5, 15)  512 / 4096 / 16384 draws: 1.04 ; 0.81 / 2.69 ; 2.25 / 10.0 ; 7.9 against 1.44 ; 1.25 / 3.19 ; 2.60 / 11.2 ; 9.1 (16 x 16 tile)
//   (8, 12): 1.37 ; 1.16 / 3.33 ; 3.08 / 12.0 ; 11.4 against 1.98 ; 1.83 / 5.19 ; 4.65 / 15.6 ; 14.8      (2, 11): 1.3x at every batch
//   (5, 10): 0.95 ; 0.80 / 1.97 ; 1.80 / 7.27 ; 6.59 against 1.75 ; 1.63 / 2.68 ; 2.31 / 8.72 ; 7.49 (multivariate row kernel)
//   m <= 8: 1.3 - 2.0x at 512 draws, a tie at 4096, 0.35 - 0.85x at 16384 (the row kernel packs 64 / m draws per wavefront)
// KF_FORWARD_B4Q=0 / 1 in the environment forces the choice (A/B measurements)
static bool kf_b4q_selected(const kf_handle* h, int m, int p, int r, int filter, unsigned flags, long B) {
  if (!kf_forward_b4q_ok(m, p, r) || (flags & KF_FLAG_TV_ANY)) return false;
  if (!(filter == KF_FILTER_UNIVARIATE || (flags & KF_FLAG_H_DIAGONAL))) return false;
  if (h->forward_kernel == KF_KERNEL_B4) return true;
  if (h->forward_kernel != KF_KERNEL_AUTO) return false;
  const char* e = kf_env_switch("KF_FORWARD_B4Q");
  if (e && *e) return atoi(e) != 0;
  return m >= 9 || B <= 2L * 4 * h->n_cu;
}

// KF_BACKWARD_B4=0 / 1 in the environment forces the choice (A/B measurements)
static bool kf_backward_b4_selected(const kf_handle* h, int m, int p, int r, long B) {
  if (!kf_backward_b4_ok(m, p, r)) return false;
  if (h->forward_kernel != KF_KERNEL_AUTO && h->forward_kernel != KF_KERNEL_B4 && h->forward_kernel != KF_KERNEL_MFMA)
    return false;
  const char* e = kf_env_switch("KF_BACKWARD_B4");  // (read per call: tests switch it)
  if (e && e[0]) return e[0] != '0';
  // measured (scripts/r02/bwd_sweep.py, 1000 steps, ms): 4096 draws m = 5 / 6 / 8 / 10 / 14: 1.72 / 1.78 / 2.46 / 2.56 / 3.69 against
  // 2.54 / 2.30 / 2.63 / 3.20 / 4.15 (round-1 choice); m = 2 / 4: 1.22 / 1.71 against 0.89 / 1.53, but 0.60 / 0.69 against
  // 0.70 / 1.32 at 512 draws (one wavefront per draw fills the chip sooner)
  return m >= 5 || B <= 1024;
}

static bool kf_mp_offsets_ok(int m, int T) { return 16.0 * ((double)T + 1.0) * m * m < 2147483648.0; }

static void kf_mp_pick(const kf_handle* h, int m, long B, int* Mo, int* Ho) {
  int M = m + (m & 1);
  if (M < 4) M = 4;
  // compiled lanes-per-row choices per M: {narrow, wide}
  int h_narrow, h_wide;
  switch (M) {
    case 16: case 14: h_narrow = 2; h_wide = 4; break;
    case 12: case 10: h_narrow = 2; h_wide = 3; break;
    case 8: h_narrow = 2; h_wide = 4; break;
    case 6: h_narrow = 1; h_wide = 3; break;
    default: h_narrow = 1; h_wide = 2; break;
  }
  int H = 0;
  if (h->lanes_per_draw > 0 && h->lanes_per_draw % M == 0) {
    const int want = h->lanes_per_draw / M;
    if (want == h_narrow || want == h_wide) H = want;
  }
  if (H == 0) {
    // narrow rows while they still give every SIMD two wavefronts
    const long n_simd = (h ? h->n_cu : 256) * 4L;
    const long dpw = KF_WAVE / (M * h_narrow);
    const long waves = (B + dpw - 1) / dpw;
    H = (waves >= 2 * n_simd) ? h_narrow : h_wide;
  }
  *Mo = M;
  *Ho = H;
}

// AUTO: accumulator-layout (MFMA) kernel instead of the row-owner kernels?  Measured on MI355X
// (scripts/gpu_kernel_sweep.py): one wavefront per draw costs the same whatever m <= 16 is, the
// row kernels pack 64/m draws per wavefront but pay LDS operand traffic ~ m^3.
//   p == 1: m <= 6 row kernels; m = 7..9 MFMA while the batch is small; m >= 10 MFMA
//   p  > 1: m >= 11 MFMA, below that the multivariate row kernel
static bool kf_auto_mfma(const kf_handle* h, int m, int p, long B) {
  if (p > 1) return m >= 11;
  if (m >= 10) return true;
  if (m <= 6) return false;
  const long n_simd = (h ? h->n_cu : 256) * 4L;
  return (B * m + KF_WAVE - 1) / KF_WAVE < 2 * n_simd;
}

static int kf_launch_forward_mp(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  int M, H;
  kf_mp_pick(h, a.m, a.B, &M, &H);
#define KF_TRY_MP(M_, H_) \
  if (M == M_ && H == H_) return kf_launch_forward_rowsplit_mp_m##M_##_h##H_(h, a, stream);
  KF_TRY_MP(16, 2) KF_TRY_MP(16, 4) KF_TRY_MP(14, 2) KF_TRY_MP(14, 4) KF_TRY_MP(12, 2) KF_TRY_MP(12, 3)
  KF_TRY_MP(10, 2) KF_TRY_MP(10, 3) KF_TRY_MP(8, 2) KF_TRY_MP(8, 4) KF_TRY_MP(6, 1) KF_TRY_MP(6, 3)
  KF_TRY_MP(4, 1) KF_TRY_MP(4, 2)
#undef KF_TRY_MP
  KF_SET_ERR(h, "forward(rowsplit_mp): no instantiation for M=%d H=%d", M, H);
  return KF_ERR_UNSUPPORTED;
}

static bool kf_whiten_shape_ok(const kf_handle* h, int m, int p, int r, int T, int filter, unsigned flags, long B);
static size_t kf_whiten_total_bytes(const kf_handle* h, const kf_dims* d);
// kf_filter_value_and_grad on a dense-H model by whitening (round 5): the shapes whose diagonal-H gradient runs on the adjoints that can
// carry sum_t vb_t y~_t' (kf_backward_mfma.hip, kf_backward_steady.hip: m <= 16, p <= 8)
static bool kf_vg_whiten_ok(const kf_handle* h, const kf_dims* d) {
  return d->T >= 2 && d->m <= KF_MP_M_MAX && d->p <= KF_MP_P_MAX &&
         kf_whiten_shape_ok(h, d->m, d->p, d->r, d->T, d->filter, d->flags, d->B);
}

extern "C" const char* kf_last_forward_kernel(const kf_handle* h) {
  return (h && h->last_forward_kernel) ? h->last_forward_kernel : "";
}

extern "C" const char* kf_forward_kernel_name(const kf_handle* h, const kf_dims* d) {
  return kf_forward_kernel_name_ws(h, d, (size_t)-1);
}

extern "C" const char* kf_forward_kernel_name_ws(const kf_handle* h, const kf_dims* d, size_t workspace_bytes) {
  KfTlsHandle kf_tls_guard_(h);
  if (!h || !d) return "";
  if (kf_whiten_shape_ok(h, d->m, d->p, d->r, d->T, d->filter, d->flags, d->B) && workspace_bytes >= kf_whiten_total_bytes(h, d)) {
    kf_dims d2 = *d;  // dense H, whitened: the diagonal-H kernel of the shape (behind kf_whiten_kernel)
    d2.flags |= KF_FLAG_H_DIAGONAL | KF_FLAG_Y_PER_DRAW;
    return kf_forward_kernel_name_ws(h, &d2, workspace_bytes);
  }
  const bool row_ok = d->p == 1 && d->m <= KF_ROW_M_MAX;
  if (d->flags & KF_FLAG_TV_ANY) {
    const char* e = kf_env_switch("KF_FORWARD_B4TV");
    if ((h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_B4) && kf_forward_b4tv_ok(d->m, d->p, d->r, d->flags) &&
        !(e && e[0] == '0'))
      return "kf_fwd_b4tv_kernel";
    return "kf_fwd_lds_kernel";
  }
  if (kf_b4_selected(h, d->m, d->p, d->r, d->B)) return "kf_fwd_b4_kernel";
  if (kf_b4q_selected(h, d->m, d->p, d->r, d->filter, d->flags, d->B)) return "kf_fwd_b4q_kernel";
  if (kf_nb_ok(h, d->m, d->p, d->r, d->filter, d->flags)) return "kf_fwd_mfma_nb_kernel";
  if (kf_wg_ok(h, d->m, d->p, d->r, d->filter, d->flags)) return "kf_fwd_mfma_wg_kernel";
  // (48 < m that still fits LDS: the workspace path is 5x faster, kf_launch_forward takes it when
  // the caller passed the workspace kf_workspace_bytes asks for)
  // (mirrors kf_launch_forward: without the workspace the launcher stays on the LDS kernel while it fits)
  if (!kf_forward_lds_fits(d->m, d->p) ||
      (d->m > KF_NB_M_MAX && d->B <= 65535 && workspace_bytes >= kf_forward_big_workspace_bytes(d->B, d->m, d->p)))
    return "big_gemm_kernel";
  if (h->forward_kernel == KF_KERNEL_MFMA)
    return kf_mp_ok(h, d->m, d->p, d->T, d->filter, d->flags) ? "kf_fwd_mfma_kernel" : "";
  if (d->p == 1 && d->m <= KF_LANE_M_MAX && !kf_row_over_lane(h, d->m, d->B) &&
      (h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_LANE))
    return "kf_fwd_lane_kernel";
  const bool mp_ok = kf_mp_ok(h, d->m, d->p, d->T, d->filter, d->flags);
  if (h->forward_kernel == KF_KERNEL_AUTO && mp_ok && kf_auto_mfma(h, d->m, d->p, d->B))
    return "kf_fwd_mfma_kernel";
  if (row_ok && h->forward_kernel != KF_KERNEL_LDS && h->forward_kernel != KF_KERNEL_LANE)
    return kf_split_for(h, d->m, d->B) > 1 ? "kf_fwd_rowsplit_kernel" : "kf_fwd_row_kernel";
  if (mp_ok) return kf_mp_offsets_ok(d->m, d->T) ? "kf_fwd_rowsplit_mp_kernel" : "kf_fwd_mfma_kernel";
  return "kf_fwd_lds_kernel";
}

// Dense H with p > 1 (Standard / Cholesky) on the scalar-observation kernels by whitening (kf_forward_whiten.hip): eligible when
// one of those kernels takes the shape once H is diagonal, the dense (LDS) kernel fits as the fallback, and no kernel was forced.
// KF_FORWARD_WHITEN=0 in the environment keeps the dense kernel.
static bool kf_whiten_shape_ok(const kf_handle* h, int m, int p, int r, int T, int filter, unsigned flags, long B) {
  if (p < 2 || p > 8 || (flags & (KF_FLAG_TV_ANY | KF_FLAG_H_DIAGONAL))) return false;
  if (filter != KF_FILTER_STANDARD && filter != KF_FILTER_CHOLESKY) return false;
  if (h->forward_kernel != KF_KERNEL_AUTO || !kf_forward_lds_fits(m, p)) return false;
  const char* e = kf_env_switch("KF_FORWARD_WHITEN");
  if (e && e[0] == '0') return false;
  const unsigned fd = flags | KF_FLAG_H_DIAGONAL;
  return kf_b4q_selected(h, m, p, r, filter, fd, B) || kf_nb_ok(h, m, p, r, filter, fd) || kf_wg_ok(h, m, p, r, filter, fd) ||
         kf_mp_ok(h, m, p, T, filter, fd);
}
static size_t kf_whiten_total_bytes(const kf_handle* h, const kf_dims* d) {
  kf_dims d2 = *d;
  d2.flags |= KF_FLAG_H_DIAGONAL | KF_FLAG_Y_PER_DRAW;
  size_t inner = 0;
  (void)kf_workspace_bytes_for(h, &d2, KF_WS_FORWARD, &inner);
  return kf_forward_whiten_workspace_bytes(d->B, d->T, d->p, d->m) + inner + 256;
}

int kf_launch_forward(kf_handle* h, const KfFwdArgs& a_in, void* workspace, size_t workspace_bytes,
                      hipStream_t stream) {
  KfFwdArgs a = a_in;
  if (kf_whiten_shape_ok(h, a.m, a.p, a.r, a.T, a.filter, a.flags, a.B) && workspace) {
    kf_dims d0;
    d0.B = a.B; d0.T = a.T; d0.p = a.p; d0.m = a.m; d0.r = a.r; d0.filter = a.filter; d0.flags = a.flags;
    if (workspace_bytes >= kf_whiten_total_bytes(h, &d0))
      return kf_launch_forward_whitened(h, a, workspace, workspace_bytes, stream, kf_launch_forward);
  }
  {
    const char* e = kf_env_experiment("KF_SS_ENTRY_TOL");  // experiments: the entrywise tolerance of the steady-state test (default 1e-12)
    if (e && *e && atof(e) > 0.0) a.ss_tol2 = atof(e) * atof(e);
  }
  if (a.flags & KF_FLAG_TV_ANY) {
    if (((a.flags & KF_FLAG_TV_C) && !a.c) || ((a.flags & KF_FLAG_TV_D) && !a.d)) {
      KF_SET_ERR(h, "time-varying c / d flagged but the pointer is NULL");
      return KF_ERR_ARG;
    }
    // scalar observation, m <= 10, R and Q static: the block kernel with per-step slices (kf_fwd_b4tv_kernel); KF_FORWARD_B4TV=0 in
    // the environment keeps the LDS kernel
    if ((h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_B4) && kf_forward_b4tv_ok(a.m, a.p, a.r, a.flags)) {
      const char* e = kf_env_switch("KF_FORWARD_B4TV");
      if (!(e && e[0] == '0')) return kf_launch_forward_b4tv(h, a, stream);
    }
    if (!kf_forward_lds_fits(a.m, a.p)) {
      KF_SET_ERR(h, "time-varying system matrices need the LDS-resident kernel (m=%d p=%d too large)", a.m, a.p);
      return KF_ERR_UNSUPPORTED;
    }
    return kf_launch_forward_lds(h, a, stream);
  }
  if (kf_b4_selected(h, a.m, a.p, a.r, a.B)) return kf_launch_forward_b4(h, a, workspace, workspace_bytes, stream);
  if (kf_b4q_selected(h, a.m, a.p, a.r, a.filter, a.flags, a.B)) return kf_launch_forward_b4q(h, a, workspace, workspace_bytes, stream);
  if (h->forward_kernel == KF_KERNEL_B4) {
    KF_SET_ERR(h, "forward(b4): needs p == 1, m <= 14, static matrices (got p=%d m=%d)", a.p, a.m);
    return KF_ERR_UNSUPPORTED;
  }
  // measured (MI355X, 4096 draws x 1000 steps): m = 17..40 took 25..670 ms on the LDS kernel
  if (kf_nb_ok(h, a.m, a.p, a.r, a.filter, a.flags)) return kf_launch_forward_mfma_nb(h, a, stream);
  if (kf_wg_ok(h, a.m, a.p, a.r, a.filter, a.flags)) return kf_launch_forward_mfma_wg(h, a, stream);
  // 64 < m <= 96, scalar observation, no covariance tapes (the log-likelihood-only call): P stays in LDS, one workgroup per draw
  // (kf_forward_wgl.hip; measured at 4096 x 1000: m = 70 / 80 / 96 in 0.18 / 0.18 / 0.27 s against 0.79 / 0.82 / ~0.9 s on the
  // workspace path, scripts/r03/wgl_check.py); KF_FORWARD_WGL=0 keeps the latter
  if (h->forward_kernel == KF_KERNEL_AUTO && a.filter != KF_FILTER_UNIVARIATE &&
      kf_forward_wgl_ok(a.m, a.p, a.flags, a.filt_P || a.pred_P) && workspace &&
      workspace_bytes >= kf_forward_wgl_workspace_bytes(a.B, a.m)) {
    const char* e = kf_env_switch("KF_FORWARD_WGL");
    if (!(e && e[0] == '0')) return kf_launch_forward_wgl(h, a, workspace, workspace_bytes, stream);
  }
  if (!kf_forward_lds_fits(a.m, a.p)) return kf_launch_forward_big(h, a, workspace, workspace_bytes, stream);
  // 48 < m <= ~60 still fits LDS, but one workgroup per CU at m^3 LDS reads per step: measured
  // (4096 draws x 1000 steps) 2.2 s at m = 49, 3.9 s at m = 60 against 0.62 s at m = 64 on the
  // workspace path -- taken whenever the caller provided the workspace
  if (a.m > KF_NB_M_MAX && a.B <= 65535 && workspace &&
      workspace_bytes >= kf_forward_big_workspace_bytes(a.B, a.m, a.p))
    return kf_launch_forward_big(h, a, workspace, workspace_bytes, stream);
  const bool row_ok = a.p == 1 && a.m <= KF_ROW_M_MAX;
  const bool lane_ok = a.p == 1 && a.m <= KF_LANE_M_MAX;
  const bool mp_ok = kf_mp_ok(h, a.m, a.p, a.T, a.filter, a.flags);
  if (h->forward_kernel == KF_KERNEL_MFMA) {
    if (!mp_ok) {
      KF_SET_ERR(h, "forward(mfma): needs m <= %d, p <= %d, static matrices and p == 1 / the Univariate "
                 "filter / KF_FLAG_H_DIAGONAL (got p=%d m=%d)", KF_MP_M_MAX, KF_MP_P_MAX, a.p, a.m);
      return KF_ERR_UNSUPPORTED;
    }
    return kf_launch_forward_mfma(h, a, stream);
  }
  if (h->forward_kernel == KF_KERNEL_LANE && !lane_ok) {
    KF_SET_ERR(h, "forward(lane): needs p == 1 and m <= %d (got p=%d m=%d)", KF_LANE_M_MAX, a.p, a.m);
    return KF_ERR_UNSUPPORTED;
  }
  if (lane_ok && !kf_row_over_lane(h, a.m, a.B) && (h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_LANE)) {
    switch (a.m) {
      case 1: return kf_launch_forward_lane_m1(h, a, stream);
      case 2: return kf_launch_forward_lane_m2(h, a, stream);
      case 3: return kf_launch_forward_lane_m3(h, a, stream);
      default: return kf_launch_forward_lane_m4(h, a, stream);
    }
  }
  if (h->forward_kernel == KF_KERNEL_AUTO && mp_ok && kf_auto_mfma(h, a.m, a.p, a.B))
    return kf_launch_forward_mfma(h, a, stream);
  if (h->forward_kernel == KF_KERNEL_ROW && !row_ok && !mp_ok) {
    KF_SET_ERR(h, "forward(row): needs m <= %d with p == 1, or m <= %d, p <= %d with the Univariate "
               "filter / KF_FLAG_H_DIAGONAL (got p=%d m=%d)", KF_ROW_M_MAX, KF_MP_M_MAX, KF_MP_P_MAX, a.p, a.m);
    return KF_ERR_UNSUPPORTED;
  }
  if (row_ok && h->forward_kernel != KF_KERNEL_LDS && h->forward_kernel != KF_KERNEL_LANE) {
    if (a.m % 2 == 0) {
      // lanes per row H: explicit (lanes_per_draw = m*H) or chosen from the batch size;
      // H == 1 is the plain row-owner kernel below
      const int H = kf_split_for(h, a.m, a.B);
#define KF_TRY_SPLIT(M_, H_) \
  if (a.m == M_ && H == H_) return kf_launch_forward_rowsplit_m##M_##_h##H_(h, a, stream);
      KF_TRY_SPLIT(10, 2) KF_TRY_SPLIT(10, 3) KF_TRY_SPLIT(10, 5)
      KF_TRY_SPLIT(8, 2) KF_TRY_SPLIT(8, 4)
      KF_TRY_SPLIT(6, 3)
      KF_TRY_SPLIT(4, 2)
#undef KF_TRY_SPLIT
    }
    switch (a.m) {
      case 1: return kf_launch_forward_row_m1(h, a, stream);
      case 2: return kf_launch_forward_row_m2(h, a, stream);
      case 3: return kf_launch_forward_row_m3(h, a, stream);
      case 4: return kf_launch_forward_row_m4(h, a, stream);
      case 5: return kf_launch_forward_row_m5(h, a, stream);
      case 6: return kf_launch_forward_row_m6(h, a, stream);
      case 7: return kf_launch_forward_row_m7(h, a, stream);
      case 8: return kf_launch_forward_row_m8(h, a, stream);
      case 9: return kf_launch_forward_row_m9(h, a, stream);
      default: return kf_launch_forward_row_m10(h, a, stream);
    }
  }
  if (mp_ok) return kf_mp_offsets_ok(a.m, a.T) ? kf_launch_forward_mp(h, a, stream) : kf_launch_forward_mfma(h, a, stream);
  return kf_launch_forward_lds(h, a, stream);
}

int kf_pick_lanes(const kf_handle* h, int m, long B) {
  if (h && (h->lanes_per_draw == 4 || h->lanes_per_draw == 8 || h->lanes_per_draw == 16 ||
            h->lanes_per_draw == 32 || h->lanes_per_draw == 64))
    return h->lanes_per_draw;
  int want = (m * m + 1) / 2;  // ~2 matrix elements per lane
  int G = 4;
  while (G < want && G < KF_WAVE) G *= 2;
  // keep >= 2 wavefronts per SIMD (256 CUs x 4 SIMDs) when the batch is small
  long n_simd = (h ? h->n_cu : 256) * 4L;
  while (G < KF_WAVE && B * G / KF_WAVE < 2 * n_simd) G *= 2;
  return G;
}

int kf_launch_forward_lds(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  if (a.p == 1) return kf_launch_forward_lds_p1(h, a, stream);
  if (a.p == 2) return kf_launch_forward_lds_p2(h, a, stream);
  if (a.p <= 4) return kf_launch_forward_lds_p4(h, a, stream);
  if (a.p <= 8) return kf_launch_forward_lds_p8(h, a, stream);
  return kf_launch_forward_lds_p0(h, a, stream);
}

int kf_launch_backward_lds(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  if (a.p == 1) return kf_launch_backward_lds_p1(h, a, stream);
  if (a.p == 2) return kf_launch_backward_lds_p2(h, a, stream);
  if (a.p <= 4) return kf_launch_backward_lds_p4(h, a, stream);
  if (a.p <= 8) return kf_launch_backward_lds_p8(h, a, stream);
  return kf_launch_backward_lds_p0(h, a, stream);
}

int kf_launch_backward(kf_handle* h, const KfBwdArgs& a, void* workspace, size_t workspace_bytes,
                       hipStream_t stream) {
  // cotangents of the tape outputs (kf_filter_backward_tapes): the LDS-resident adjoint is the one that carries them
  if (a.g_filt_a || a.g_pred_a || a.g_filt_P || a.g_pred_P) {
    // static matrices, m <= 16, scalar observations: the accumulator-layout adjoint's TP instantiation (kf_backward_mfma.hip;
    // KF_FORWARD_KERNEL=lds keeps the LDS adjoint, for A/B tests)
    if (!(a.flags & KF_FLAG_TV_ANY) && !a.steps && !a.accumulate && a.m <= KF_MP_M_MAX && a.p <= KF_MP_P_MAX &&
        (a.p == 1 || a.filter == KF_FILTER_UNIVARIATE || (a.flags & KF_FLAG_H_DIAGONAL)) &&
        (h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_MFMA || h->forward_kernel == KF_KERNEL_B4))
      return kf_launch_backward_mfma(h, a, stream);
    if (!kf_backward_lds_fits(a.m, a.p)) {
      KF_SET_ERR(h, "kf_filter_backward_tapes: cotangents of the tape outputs need the LDS-resident adjoint (m=%d p=%d does "
                 "not fit %d B of LDS)", a.m, a.p, KF_LDS_BYTES_MAX);
      return KF_ERR_UNSUPPORTED;
    }
    return kf_launch_backward_lds(h, a, stream);
  }
  // time-varying system matrices: the LDS-resident adjoint reloads slice t per step and writes gradient tapes
  if (a.flags & KF_FLAG_TV_ANY) {
    // round 5: T / Z / H / c / d time-varying (R, Q static), m <= 16, scalar observations: the accumulator-layout adjoint with
    // per-step slices and gradient tapes (kf_bwd_mfma_kernel<.., TV>); KF_BACKWARD_TV_MFMA=0 in the environment keeps the LDS adjoint
    if (!(a.flags & (KF_FLAG_TV_R | KF_FLAG_TV_Q)) && !a.steps && !a.accumulate && a.m <= KF_MP_M_MAX && a.p <= KF_MP_P_MAX &&
        (a.p == 1 || a.filter == KF_FILTER_UNIVARIATE || (a.flags & KF_FLAG_H_DIAGONAL)) &&
        (h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_MFMA || h->forward_kernel == KF_KERNEL_B4)) {
      const char* e = kf_env_switch("KF_BACKWARD_TV_MFMA");
      if (!(e && e[0] == '0')) return kf_launch_backward_mfma(h, a, stream);
    }
    if (!kf_backward_lds_fits(a.m, a.p)) {
      KF_SET_ERR(h, "kf_filter_backward: time-varying system matrices need the LDS-resident adjoint (m=%d p=%d does "
                 "not fit %d B of LDS)", a.m, a.p, KF_LDS_BYTES_MAX);
      return KF_ERR_UNSUPPORTED;
    }
    return kf_launch_backward_lds(h, a, stream);
  }
  // per-draw step counts (the steady-state adjoint's hand-over, kf_filter_value_and_grad): kf_backward_mfma.hip is the one adjoint
  // that honours them -- whatever forward kernel was forced on the handle; any other shape is refused, never swept in full
  if (a.steps) {
    const bool ok = a.m <= KF_MP_M_MAX && a.p <= KF_MP_P_MAX &&
                    (a.p == 1 || a.filter == KF_FILTER_UNIVARIATE || (a.flags & KF_FLAG_H_DIAGONAL));
    if (!ok) {
      KF_SET_ERR(h, "backward: per-draw step counts need the accumulator-layout adjoint (m <= %d, p <= %d, scalar observations; "
                 "got p=%d m=%d)", KF_MP_M_MAX, KF_MP_P_MAX, a.p, a.m);
      return KF_ERR_UNSUPPORTED;
    }
    return kf_launch_backward_mfma(h, a, stream);
  }
  // scalar observation, m <= 14: adjoint of the augmented recursion on 4x4x4 block MFMAs (kf_backward_b4.hip); needs
  // B (m + m^2) doubles of workspace for the hand-over of its last step (else the 16x16 tile adjoint below)
  if (kf_backward_b4_selected(h, a.m, a.p, a.r, a.B) && workspace &&
      workspace_bytes >= kf_backward_b4_workspace_bytes(a.B, a.m))
    return kf_launch_backward_b4(h, a, workspace, workspace_bytes, stream);
  // 16 < m <= 48, scalar observations: block-tile adjoint (kf_backward_mfma_nb.hip); measured at
  // 4096 draws x 1000 steps the LDS adjoint took 144 ms (m = 20) .. 2.6 s (m = 40)
  if ((h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_MFMA) &&
      kf_backward_mfma_nb_ok(a.m, a.p, a.r) &&
      (a.p == 1 || a.filter == KF_FILTER_UNIVARIATE || (a.flags & KF_FLAG_H_DIAGONAL)))
    return kf_launch_backward_mfma_nb(h, a, stream);
  // 48 < m <= 64: four wavefronts per draw (kf_backward_mfma_wg.hip)
  if ((h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_MFMA) &&
      kf_backward_mfma_wg_ok(a.m, a.p, a.r) &&
      (a.p == 1 || a.filter == KF_FILTER_UNIVARIATE || (a.flags & KF_FLAG_H_DIAGONAL)))
    return kf_launch_backward_mfma_wg(h, a, stream);
  // tiles too large for LDS (m > ~45): workspace + MFMA-GEMM adjoint, scalar observations only
  if (!kf_backward_lds_fits(a.m, a.p)) return kf_launch_backward_big(h, a, workspace, workspace_bytes, stream);
  // accumulator-layout adjoint (kf_backward_mfma.hip): scalar observations only, i.e. p == 1, the
  // Univariate filter, or a diagonal H asserted by the caller (then g_H carries the diagonal and
  // g_P0 the symmetric part)
  const bool mfma_ok = a.m <= KF_MP_M_MAX && a.p <= KF_MP_P_MAX &&
                       (a.p == 1 || a.filter == KF_FILTER_UNIVARIATE || (a.flags & KF_FLAG_H_DIAGONAL));
  if (h->forward_kernel == KF_KERNEL_MFMA) {
    if (!mfma_ok) {
      KF_SET_ERR(h, "backward(mfma): needs m <= %d, p <= %d and p == 1 / the Univariate filter / "
                 "KF_FLAG_H_DIAGONAL (got p=%d m=%d)", KF_MP_M_MAX, KF_MP_P_MAX, a.p, a.m);
      return KF_ERR_UNSUPPORTED;
    }
    return kf_launch_backward_mfma(h, a, stream);
  }
  // measured (MI355X, scripts/probes/bwd_sweep.py): m = 10, B = 4096: 3.2 ms vs 6.1 ms (row-split
  // adjoint); m = 8: 2.7 vs 4.4 ms; m = 6: 2.6 vs 2.4 ms (7.3 vs 17 ms at B = 32768) -> even m <= 6
  // stays on the row-split adjoint; C4 (p = 5, m = 15, B = 16384): 44 ms vs 396 ms (LDS adjoint)
  if (h->forward_kernel == KF_KERNEL_AUTO && mfma_ok && (a.p > 1 || a.m >= 7 || (a.m & 1)))
    return kf_launch_backward_mfma(h, a, stream);
  // p == 1, even m <= 10: register-resident row-split adjoint (KF_FORWARD_KERNEL=lds forces the
  // generic LDS-resident adjoint, e.g. for A/B tests)
  if (a.p == 1 && a.m <= KF_ROW_M_MAX && a.m % 2 == 0 && h->forward_kernel != KF_KERNEL_LDS) {
    int H = 0;
    if (h->lanes_per_draw > 0 && h->lanes_per_draw % a.m == 0) H = h->lanes_per_draw / a.m;
    if (H == 0) H = 1;  // measured at m = 10, B = 4096: H = 1 6.1 ms, H = 3 7.0 ms (LDS kernel 14.8 ms)
#define KF_TRY_BSPLIT(M_, H_) \
  if (a.m == M_ && H == H_) return kf_launch_backward_rowsplit_m##M_##_h##H_(h, a, stream);
    KF_TRY_BSPLIT(10, 1)
    KF_TRY_BSPLIT(10, 2)
    KF_TRY_BSPLIT(10, 3)
    KF_TRY_BSPLIT(10, 5)
    KF_TRY_BSPLIT(8, 1)
    KF_TRY_BSPLIT(8, 2)
    KF_TRY_BSPLIT(8, 4)
    KF_TRY_BSPLIT(6, 1)
    KF_TRY_BSPLIT(6, 3)
    KF_TRY_BSPLIT(4, 1)
    KF_TRY_BSPLIT(4, 2)
    KF_TRY_BSPLIT(2, 1)
#undef KF_TRY_BSPLIT
  }
  return kf_launch_backward_lds(h, a, stream);
}

static int check_dims(kf_handle* h, const kf_dims* d, bool need_p) {
  if (!h) return KF_ERR_ARG;
  if (!d) {
    KF_SET_ERR(h, "dims is NULL");
    return KF_ERR_ARG;
  }
  if (d->B < 1 || d->T < 1 || d->m < 1 || d->r < 1 || (need_p && d->p < 1)) {
    KF_SET_ERR(h, "bad dims: B=%ld T=%d p=%d m=%d r=%d (all must be >= 1)", (long)d->B, d->T, d->p,
               d->m, d->r);
    return KF_ERR_ARG;
  }
  if (need_p) {
    if (d->filter < KF_FILTER_STANDARD || d->filter > KF_FILTER_SINGLE) {
      KF_SET_ERR(h, "unknown filter %d", d->filter);
      return KF_ERR_ARG;
    }
    if (d->filter == KF_FILTER_SINGLE && d->p != 1) {
      // assert_data_is_1d, pymc_statespace/filters/kalman_filter.py:19,:329
      KF_SET_ERR(h, "UnivariateTimeSeries filter requires data be at most 1-dimensional");
      return KF_ERR_ARG;
    }
    if (d->p > 255 || d->m > 32767) {
      KF_SET_ERR(h, "p=%d m=%d out of range", d->p, d->m);
      return KF_ERR_ARG;
    }
  }
  return KF_OK;
}

// kf_filter_value_and_grad's workspace: [ones B | pred_a B (T+1) m | pred_P B (T+1) m^2 | what the passes need], 64-byte pieces
static size_t kf_vg_round(size_t x) { return (x + 63) & ~(size_t)63; }
static size_t kf_vg_tape_bytes(const kf_dims* d) {
  const size_t B = (size_t)d->B, n1 = (size_t)d->T + 1, m = (size_t)d->m;
  const size_t compact = kf_vg_round(8 * B * n1 * m) + kf_vg_round(8 * B * n1 * m * m);
  const size_t n4 = 4 * ((m + 2 + 3) / 4);
  const size_t tile = kf_vg_round(8 * B * n1 * n4 * n4);  // the block kernels' own tape (p = 1, m <= 14)
  return kf_vg_round(8 * B) + (m + 2 <= 16 && tile > compact ? tile : compact) + 64;
}
// value-and-gradient through the 4x4x4-block kernels' tile tape: both kernels selected, static, a series to sweep
static bool kf_vg_tile_path(const kf_handle* h, const kf_dims* d) {
  if ((d->flags & KF_FLAG_TV_ANY) || d->T < 2) return false;
  const char* e = kf_env_switch("KF_VG_TILE_TAPE");  // =0: compact tapes (A/B measurements)
  if (e && e[0] == '0') return false;
  if (kf_b4_selected(h, d->m, d->p, d->r, d->B) && kf_backward_b4_selected(h, d->m, d->p, d->r, d->B)) return true;
  // 5 <= m <= 8 at batches where the FORWARD alone is faster on the packed row kernels (kf_b4_auto): the call as a whole is not --
  // measured (scripts/r04/vg_rowshapes.py, 500 steps, ms, row forward + row-split adjoint ; block kernels): m = 6 at 16384 / 32768
  // draws 4.19 ; 1.81 / 8.55 ; 3.48, m = 8 6.04 ; 2.52 / 12.0 ; 4.89 (m <= 4 stays with the lane kernel: 1.67 ; 1.85 at 16384 draws)
  return h->forward_kernel == KF_KERNEL_AUTO && d->m >= 5 && kf_forward_b4_ok(d->m, d->p, d->r) &&
         kf_backward_b4_selected(h, d->m, d->p, d->r, d->B);
}

// value-and-gradient with the steady-state adjoint (kf_backward_steady.hip): the forward kernel must report where its
// steady-state phase starts (the multi-observation block kernel does) and the exact adjoint behind it must take per-draw step
// counts (kf_backward_mfma.hip: m <= 16, scalar observations).  KF_VG_STEADY=0 in the environment: the exact adjoint at every step
static bool kf_vg_steady_path(const kf_handle* h, const kf_dims* d) {
  if ((d->flags & (KF_FLAG_TV_ANY | KF_FLAG_EXACT_RECURSION)) || d->T < 64) return false;
  const char* e = kf_env_switch("KF_VG_STEADY");
  if (e && e[0] == '0') return false;
  const char* sse = kf_env_switch("KF_STEADY_STATE");
  if (sse && *sse && atoi(sse) == 0) return false;
  // (m = 8 beyond 2048 draws: the FORWARD alone is faster on the packed multi-observation row kernel, the call is not -- p = 3,
  // 16384 x 300: 8.84 ms against 6.65 on the block kernels, scripts/r04/vg_dispatch_sweep.py)
  const char* fe = kf_env_switch("KF_FORWARD_B4Q");
  const bool for_the_call = h->forward_kernel == KF_KERNEL_AUTO && kf_forward_b4q_ok(d->m, d->p, d->r) &&
                            (d->filter == KF_FILTER_UNIVARIATE || (d->flags & KF_FLAG_H_DIAGONAL)) && !(fe && *fe && atoi(fe) == 0);
  if (!kf_b4q_selected(h, d->m, d->p, d->r, d->filter, d->flags, d->B) && !for_the_call) return false;
  if (!(d->p >= 3 && d->p <= 5 && d->m >= 8)) return false;  // (the shapes kf_forward_b4q.hip's steady-state phase covers)
  return kf_backward_steady_ok(d->m, d->p, d->r) && d->m <= KF_MP_M_MAX;
}

// value-and-gradient behind the LANE kernel (p == 1, m <= 4 where AUTO does not take the block kernels: C2) with the steady-state
// adjoint: the lane kernel reports where each draw's steady-state phase starts and writes every slot of the compact tapes;
// kf_bwd_steady_scan_kernel takes the steps behind that point (a_t and the converged covariance from the tapes), kf_dare_bwd_kernel
// the Riccati adjoint, kf_bwd_b4_kernel (compact tapes, per-draw step counts, snake order) the steps before it.  C2 (ARMA(1,1),
// 4096 x 1000): 1.12 -> 0.59 ms.  KF_VG_STEADY=0 / KF_FLAG_EXACT_RECURSION: the full sweep as before; =1 / KF_FLAG_STEADY_STATE: any batch.
static bool kf_vg_lane_steady_path(const kf_handle* h, const kf_dims* d) {
  // (m <= 3: at m = 4 the lane kernel is AUTO's choice only beyond 4096 draws, where the packed row kernels + the full row-split sweep
  // are faster -- 8192 x 1000: 2.48 ms against 2.79 on this path)
  if ((d->flags & (KF_FLAG_TV_ANY | KF_FLAG_EXACT_RECURSION)) || d->T < 64 || d->p != 1 || d->m > 3) return false;
  if (d->B > 0x7fffffffL) return false;
  if (h->forward_kernel != KF_KERNEL_AUTO && h->forward_kernel != KF_KERNEL_LANE) return false;
  if (kf_b4_selected(h, d->m, d->p, d->r, d->B)) return false;   // (the tile-tape path)
  const char* e = kf_env_switch("KF_VG_STEADY");
  if (e && e[0] == '0') return false;
  const char* sse = kf_env_switch("KF_STEADY_STATE");
  if (sse && *sse && atoi(sse) == 0) return false;
  if (!(kf_backward_steady_scan_ok(d->m, d->p) && kf_backward_b4_ok(d->m, d->p, d->r))) return false;
  if ((d->flags & KF_FLAG_STEADY_STATE) || (e && e[0] == '1') || (sse && atoi(sse) != 0)) return true;
  // AUTO: up to two rounds of the chip's wavefront slots -- the sweep before the convergence point runs one wavefront per draw, the
  // full sweep it replaces packs 64 / m^2 draws into one.  Measured (ARMA(1,1) x 1000 steps, ms, this path ; the full sweep):
  // 512 draws 0.49 ; 0.98, 4096 (C2) 0.59 ; 1.31 (1.12 with the round-3 default), 16384 1.77 ; 1.56; m = 1 / 3 at 4096: 0.98 ; 2.14, 1.66 ; 2.60
  return d->B <= 32L * (h->n_cu > 0 ? h->n_cu : 256);
}

// value-and-gradient through the tile tape WITH the steady-state adjoint (p == 1, m <= 10: the block kernel's tapes go through its LDS
// rings): the forward kernel reports each draw's convergence point ts, leaves the tile tape of steps < ts and a_t of steps >= ts;
// kf_bwd_steady_kernel + kf_dare_bwd_kernel take the steps behind ts, kf_bwd_b4_kernel (per-draw step counts) the steps before it.
// KF_VG_STEADY=0 / KF_STEADY_STATE=0 / KF_FLAG_EXACT_RECURSION: the exact adjoint at every step
static bool kf_vg_tile_steady_path(const kf_handle* h, const kf_dims* d) {
  if (!kf_vg_tile_path(h, d)) return false;
  if ((d->flags & KF_FLAG_EXACT_RECURSION) || d->T < 64 || d->m > 10) return false;
  const char* e = kf_env_switch("KF_VG_STEADY");
  if (e && e[0] == '0') return false;
  const char* sse = kf_env_switch("KF_STEADY_STATE");
  if (sse && atoi(sse) == 0) return false;
  if (!kf_backward_steady_ok(d->m, d->p, d->r)) return false;
  if ((d->flags & KF_FLAG_STEADY_STATE) || (e && e[0] == '1') || (sse && atoi(sse) != 0)) return true;
  // AUTO: where it wins.  A launch lasts as long as its most loaded SIMD: below one round of the chip's wavefront slots (4 per
  // SIMD) that is the slowest draw's chain -- a draw whose covariance never converges gains nothing and the steady-state kernels are
  // more links -- measured (scripts/r04/vg_time.py sweep, structural m = 10, ms, steady ; exact -- profiles/r04/vg_sweep_*.txt):
  // 512 draws x 1000 1.54 ; 1.15, 2048 x 1000 1.71 ; 1.56, 4096 x 500 1.41 ; 1.41, 4096 x 1000 2.15 ; 2.78, 4096 x 2000 4.04 ; 5.61,
  // 8192 x 500 2.61 ; 2.78, 8192 x 1000 3.55 ; 5.49, 8192 x 2000 (C3) 5.66 ; 11.0
  const long slots = 16L * (h->n_cu > 0 ? h->n_cu : 256);
  return (d->B >= slots && d->T >= 768) || d->B >= 2 * slots;
}

extern "C" int kf_workspace_bytes_for(const kf_handle* h, const kf_dims* dims, int which, size_t* bytes) {
  KfTlsHandle kf_tls_guard_(h);
  if (!h || !dims || !bytes) return KF_ERR_ARG;
  if (which == KF_WS_SMOOTHER_FROM_FILTER) {
    *bytes = kf_smoother_bf_big_ok(dims->m, dims->p, dims->flags & ~(unsigned)(KF_FLAG_EXACT_RECURSION | KF_FLAG_STEADY_STATE))
                 ? kf_smoother_bf_big_workspace_bytes(dims->B, dims->m) : 0;
    return KF_OK;
  }
  if (which & KF_WS_VALUE_AND_GRAD) {
    size_t inner = 0;
    const int rc = kf_workspace_bytes_for(h, dims, KF_WS_FORWARD | KF_WS_BACKWARD, &inner);
    if (rc) return rc;
    *bytes = kf_vg_tape_bytes(dims) + inner;
    if (kf_vg_tile_steady_path(h, dims)) {
      const size_t B = (size_t)dims->B, m = (size_t)dims->m, n1 = (size_t)dims->T + 1;
      *bytes += kf_vg_round(8 * B * n1 * m) + 3 * kf_vg_round(4 * B) + kf_vg_round(64) + kf_vg_round(8 * B * m) + 3 * kf_vg_round(8 * B * m * m);
    }
    if (kf_vg_steady_path(h, dims)) {
      const size_t B = (size_t)dims->B, m = (size_t)dims->m;
      *bytes += kf_vg_round(4 * B) + kf_vg_round(8 * B * m) + 2 * kf_vg_round(8 * B * m * m);
    }
    if (kf_vg_lane_steady_path(h, dims)) {
      const size_t B = (size_t)dims->B, m = (size_t)dims->m;
      *bytes += 2 * kf_vg_round(4 * B) + kf_vg_round(8 * B * m) + 2 * kf_vg_round(8 * B * m * m) +
                kf_backward_b4_workspace_bytes(dims->B, dims->m) + 64;
    }
    if (kf_vg_whiten_ok(h, dims)) {  // dense H: [flag | whitened inputs, L^-1, whitened gradients | the diagonal-H call's own need]
      kf_dims d2 = *dims;
      d2.flags |= KF_FLAG_H_DIAGONAL | KF_FLAG_Y_PER_DRAW;
      size_t inner2 = 0;
      const int rc2 = kf_workspace_bytes_for(h, &d2, KF_WS_VALUE_AND_GRAD, &inner2);
      if (rc2) return rc2;
      KfWhitenGrad W;
      const size_t wneed = kf_whiten_grad_carve(nullptr, dims->B, dims->T, dims->p, dims->m, &W) + inner2 + 512;
      *bytes += 512;   // (the dense redo's tapes start behind the flag)
      if (wneed > *bytes) *bytes = wneed;
    }
    return KF_OK;
  }
  *bytes = 0;  // the LDS- and register-resident kernels keep all scratch on chip
  const int m = dims->m, p = dims->p, r = dims->r;
  // forward, dense H with p > 1: the whitened inputs (one series per draw) + what the diagonal-H kernel needs itself
  if ((which & KF_WS_FORWARD) && dims->T >= 1 && kf_whiten_shape_ok(h, m, p, r, dims->T, dims->filter, dims->flags, dims->B))
    *bytes = kf_whiten_total_bytes(h, dims);
  const bool tile_kernels = h->forward_kernel == KF_KERNEL_AUTO || h->forward_kernel == KF_KERNEL_MFMA;
  const bool scalar_obs = p == 1 || dims->filter == KF_FILTER_UNIVARIATE || (dims->flags & KF_FLAG_H_DIAGONAL);
  // forward: HBM-workspace path for tiles beyond LDS and, when no on-chip kernel takes the shape, for m > 48
  if ((which & KF_WS_FORWARD) && m >= 1 && p >= 1 && !(dims->flags & KF_FLAG_TV_ANY) &&
      (!kf_forward_lds_fits(m, p) || m > KF_NB_M_MAX) &&
      !kf_nb_ok(h, m, p, r, dims->filter, dims->flags) && !kf_wg_ok(h, m, p, r, dims->filter, dims->flags))
    *bytes = kf_forward_big_workspace_bytes(dims->B, m, p);
  // forward, block kernel with its steady-state phase: hand-over buffers of the split pipeline (kf_b4_split.h); without them the
  // launcher keeps the steady-state phase inside the recursion kernel
  if ((which & KF_WS_FORWARD) && kf_b4_selected(h, m, p, r, dims->B) &&
      kf_forward_b4_split_ok(h, dims->B, dims->T, p, m, r, dims->flags)) {
    const size_t sw = kf_b4_split_workspace_bytes(dims->B, m, dims->T);
    if (sw > *bytes) *bytes = sw;
  }
  // forward, block kernel on a small batch: the time-chunked pipeline's buffers (kf_forward_b4c.hip); without them the sequential kernel runs
  if ((which & KF_WS_FORWARD) && dims->T >= 1 && kf_b4_selected(h, m, p, r, dims->B) &&
      kf_forward_b4c_shape_ok(h, dims->B, dims->T, p, m, r, dims->flags)) {
    int cs = 0, cn = 0;
    kf_forward_b4c_plan(h, dims->B, dims->T, &cs, &cn);
    B4cWs Wc;
    const size_t cw = kf_forward_b4c_carve(nullptr, dims->B, dims->T, m, cn, &Wc);
    if (cw > *bytes) *bytes = cw;
  }
  // forward, multi-observation block kernel with its steady-state phase: hand-over buffers + per-draw constants of the split phase
  if ((which & KF_WS_FORWARD) && !(dims->flags & (KF_FLAG_TV_ANY | KF_FLAG_EXACT_RECURSION)) && p >= 3 && p <= 5 && m >= 8 && m <= 15 &&
      dims->T >= 64 && (kf_b4q_selected(h, m, p, r, dims->filter, dims->flags, dims->B) ||
                        ((which & KF_WS_BACKWARD) && kf_vg_steady_path(h, dims)))) {
    const size_t sw = kf_b4q_split_workspace_bytes(dims->B);
    if (sw > *bytes) *bytes = sw;
  }
  if ((which & KF_WS_SMOOTHER) && m >= 1 && m <= KF_WG_M_MAX) {  // flags of the register-tile smoothers
    size_t sm = kf_smoother_mfma_workspace_bytes(dims->B);
    if (kf_smoother_b4_ok(m, r) && m <= 10) sm += kf_smoother_b4_split_workspace_bytes(dims->B, m);  // kf_smoother_b4_steady.hip
    if (sm > *bytes) *bytes = sm;
  }
  if ((which & KF_WS_BACKWARD) && m >= 1 && p >= 1 && dims->T >= 1 && !kf_backward_lds_fits(m, p) &&
      !(tile_kernels && scalar_obs && (kf_backward_mfma_nb_ok(m, p, r) || kf_backward_mfma_wg_ok(m, p, r)))) {
    const size_t bw = kf_backward_big_workspace_bytes(dims->B, m, p, dims->T, r);
    if (bw > *bytes) *bytes = bw;
  }
  if ((which & KF_WS_BACKWARD) && dims->T >= 2 && !(dims->flags & KF_FLAG_TV_ANY) && kf_backward_b4_selected(h, m, p, r, dims->B)) {
    const size_t bw = kf_backward_b4_workspace_bytes(dims->B, m);
    if (bw > *bytes) *bytes = bw;
  }
  if ((which & KF_WS_SMOOTHER) && m >= 1 && !kf_smoother_lds_fits(m) &&
      !(kf_smoother_mfma_wg_ok(m, r) && h->forward_kernel != KF_KERNEL_LDS)) {
    const size_t sm = kf_smoother_big_workspace_bytes(dims->B, m);
    if (sm > *bytes) *bytes = sm;
  }
  return KF_OK;
}

extern "C" int kf_workspace_bytes(const kf_handle* h, const kf_dims* dims, size_t* bytes) {
  return kf_workspace_bytes_for(h, dims, KF_WS_ALL, bytes);
}

extern "C" int kf_filter_forward(kf_handle* h, const kf_dims* dims, const double* y, const double* a0,
                                 const double* P0, const double* T, const double* Z, const double* R,
                                 const double* H, const double* Q, const double* c, const double* d,
                                 double* logp, double* ll_obs, double* filt_a, double* pred_a,
                                 double* filt_P, double* pred_P, int32_t* info, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  int rc = check_dims(h, dims, true);
  if (rc) return rc;
  if (!y || !a0 || !P0 || !T || !Z || !R || !H || !Q || !logp) {
    KF_SET_ERR(h, "kf_filter_forward: y, a0, P0, T, Z, R, H, Q and logp must be non-NULL");
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  KfFwdArgs a;
  a.B = dims->B; a.T = dims->T; a.p = dims->p; a.m = dims->m; a.r = dims->r;
  a.filter = dims->filter; a.flags = dims->flags;
  a.y = y; a.a0 = a0; a.P0 = P0; a.Tm = T; a.Z = Z; a.R = R; a.H = H; a.Q = Q; a.c = c; a.d = d;
  a.logp = logp; a.ll_obs = ll_obs; a.filt_a = filt_a; a.pred_a = pred_a; a.filt_P = filt_P;
  a.pred_P = pred_P; a.info = info; a.tile_tape = nullptr;
  return kf_launch_forward(h, a, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int kf_filter_backward(kf_handle* h, const kf_dims* dims, const double* y, const double* a0,
                                  const double* P0, const double* T, const double* Z, const double* R,
                                  const double* H, const double* Q, const double* c, const double* d,
                                  const double* pred_a, const double* pred_P, const double* g_logp,
                                  const double* g_ll_obs, double* g_a0, double* g_P0, double* g_T,
                                  double* g_Z, double* g_R, double* g_H, double* g_Q, double* g_c,
                                  double* g_d, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_dims(h, dims, true);
  if (rc) return rc;
  if (!y || !a0 || !P0 || !T || !Z || !R || !H || !Q || !pred_a || !pred_P) {
    KF_SET_ERR(h, "kf_filter_backward: inputs and the predicted tapes must be non-NULL");
    return KF_ERR_ARG;
  }
  if (!g_logp && !g_ll_obs) {
    KF_SET_ERR(h, "kf_filter_backward: g_logp and g_ll_obs are both NULL");
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  KfBwdArgs a;
  a.B = dims->B; a.T = dims->T; a.p = dims->p; a.m = dims->m; a.r = dims->r;
  a.filter = dims->filter; a.flags = dims->flags;
  a.y = y; a.a0 = a0; a.P0 = P0; a.Tm = T; a.Z = Z; a.R = R; a.H = H; a.Q = Q; a.c = c; a.d = d;
  a.pred_a = pred_a; a.pred_P = pred_P; a.g_logp = g_logp; a.g_ll_obs = g_ll_obs;
  a.g_a0 = g_a0; a.g_P0 = g_P0; a.g_T = g_T; a.g_Z = g_Z; a.g_R = g_R; a.g_H = g_H; a.g_Q = g_Q;
  a.g_c = g_c; a.g_d = g_d;
  a.tape_T = dims->T; a.y_T = dims->T; a.tile_tape = nullptr; a.init_ab = nullptr; a.init_Pb = nullptr; a.accumulate = 0;
  // (internal: the whitened inner call of kf_filter_value_and_grad on a dense-H model asks the scalar-observation adjoint for Gamma~ too)
  if (h->vg_gyy && a.p > 1 && (a.flags & KF_FLAG_H_DIAGONAL) && a.m <= KF_MP_M_MAX && a.p <= KF_MP_P_MAX) a.g_yy = h->vg_gyy;
  return kf_launch_backward(h, a, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int kf_filter_backward_tapes(kf_handle* h, const kf_dims* dims, const double* y, const double* a0,
                                        const double* P0, const double* T, const double* Z, const double* R,
                                        const double* H, const double* Q, const double* c, const double* d,
                                        const double* pred_a, const double* pred_P, const double* g_logp,
                                        const double* g_ll_obs, const double* g_filt_a, const double* g_pred_a,
                                        const double* g_filt_P, const double* g_pred_P, double* g_a0, double* g_P0,
                                        double* g_T, double* g_Z, double* g_R, double* g_H, double* g_Q, double* g_c,
                                        double* g_d, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_dims(h, dims, true);
  if (rc) return rc;
  if (!y || !a0 || !P0 || !T || !Z || !R || !H || !Q || !pred_a || !pred_P) {
    KF_SET_ERR(h, "kf_filter_backward_tapes: inputs and the predicted tapes must be non-NULL");
    return KF_ERR_ARG;
  }
  if (!g_logp && !g_ll_obs && !g_filt_a && !g_pred_a && !g_filt_P && !g_pred_P) {
    KF_SET_ERR(h, "kf_filter_backward_tapes: every cotangent is NULL");
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  KfBwdArgs a;
  a.B = dims->B; a.T = dims->T; a.p = dims->p; a.m = dims->m; a.r = dims->r;
  a.filter = dims->filter; a.flags = dims->flags;
  a.y = y; a.a0 = a0; a.P0 = P0; a.Tm = T; a.Z = Z; a.R = R; a.H = H; a.Q = Q; a.c = c; a.d = d;
  a.pred_a = pred_a; a.pred_P = pred_P; a.g_logp = g_logp; a.g_ll_obs = g_ll_obs;
  a.g_filt_a = g_filt_a; a.g_pred_a = g_pred_a; a.g_filt_P = g_filt_P; a.g_pred_P = g_pred_P;
  a.g_a0 = g_a0; a.g_P0 = g_P0; a.g_T = g_T; a.g_Z = g_Z; a.g_R = g_R; a.g_H = g_H; a.g_Q = g_Q;
  a.g_c = g_c; a.g_d = g_d;
  a.tape_T = dims->T; a.y_T = dims->T; a.tile_tape = nullptr; a.init_ab = nullptr; a.init_Pb = nullptr; a.accumulate = 0;
  return kf_launch_backward(h, a, workspace, workspace_bytes, (hipStream_t)stream);
}

namespace {
__global__ void kf_fill_kernel(double* x, long n, double v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

// Launch order of draws whose sweeps have very different lengths (the steady-state adjoint and the exact adjoint behind the block
// kernel: ts[b] steps of one, n - ts[b] of the other).  One wavefront per draw, four resident per SIMD: a launch lasts as long as its
// most loaded SIMD, and with draws in their natural order that is a SIMD that happens to hold two slow draws (measured at the
// headline shape: the exact adjoint over 49 % of the steps took 82 % of the full sweep's time).  Measured placement on MI355X
// (scripts/probes/wg_placement.hip: 4096 one-wave workgroups, 10 KB of LDS each): workgroups i, i + S, i + 2 S, i + 3 S share SIMD
// number i, S = 4 x CUs.  So: rank the draws by ts, longest first (counting sort on 2048 buckets), and deal them out in snake order
// -- round 0 left to right, round 1 right to left, ... -- so that every SIMD gets one draw of each quarter and about the same
// total; draws beyond 4 S keep descending order (the dispatcher hands them to whichever SIMD frees a slot first: longest first).
// The order only decides WHERE a draw runs: results do not depend on it.
constexpr int KF_ORD_BINS = 2048;
//
// The FORWARD launch of a value-and-gradient call is ordered the same way from the convergence points of the PREVIOUS call that used the
// same workspace (hint != NULL: the ts slot still holds them; hdr = {magic, B, n, m} says so -- a sampler's successive calls move the
// parameters a little, the convergence points with them).  check != 0: a header that does not match (first call, another shape, a
// workspace someone else wrote to) gives the identity order; whatever the slot holds, the list is a permutation.  check == 0: the
// header is (re)written for the next call.
constexpr int KF_ORD_MAGIC = 0x6b665f74;
__global__ __launch_bounds__(1024) void kf_vg_order_kernel(const int32_t* __restrict__ ts, int n, int B, int S, int32_t* __restrict__ list,
                                                           int32_t* __restrict__ hdr, int check, int m) {
  __shared__ int hist[KF_ORD_BINS];   // bucket j = KF_ORD_BINS - 1 - key: ascending j = descending ts
  const int tid = threadIdx.x;
  if (hdr) {
    if (check) {
      if (hdr[0] != KF_ORD_MAGIC || hdr[1] != B || hdr[2] != n || hdr[3] != m) {
        for (int b = tid; b < B; b += 1024) list[b] = b;
        return;
      }
    } else if (tid < 4) {
      hdr[tid] = tid == 0 ? KF_ORD_MAGIC : tid == 1 ? B : tid == 2 ? n : m;
    }
  }
  for (int j = tid; j < KF_ORD_BINS; j += 1024) hist[j] = 0;
  __syncthreads();
  auto bucket = [&](int v) {
    v = v < 0 ? 0 : (v > n ? n : v);
    const int key = (int)(((long)v * (KF_ORD_BINS - 1)) / (n > 0 ? n : 1));
    return KF_ORD_BINS - 1 - key;
  };
  for (int b = tid; b < B; b += 1024) atomicAdd(&hist[bucket(ts[b])], 1);
  __syncthreads();
  if (tid < 64) {  // exclusive prefix over the buckets: lane l owns buckets [32 l, 32 l + 32)
    int loc = 0;
    for (int k = 0; k < 32; ++k) loc += hist[32 * tid + k];
    int incl = loc;
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if (tid >= off) incl += up;
    }
    int run = incl - loc;
    for (int k = 0; k < 32; ++k) {
      const int c = hist[32 * tid + k];
      hist[32 * tid + k] = run;
      run += c;
    }
  }
  __syncthreads();
  const int R = 4 * S;
  for (int b = tid; b < B; b += 1024) {
    const int r = atomicAdd(&hist[bucket(ts[b])], 1);   // rank, longest first (order inside a bucket: as the atomics fall)
    int pos = r;
    if (r < R) {
      const int round = r / S, col = r - round * S;
      const int len = (B - round * S) < S ? (B - round * S) : S;
      pos = round * S + ((round & 1) ? (len - 1 - col) : col);
    }
    list[pos] = b;
  }
}
}  // namespace

int kf_launch_order(kf_handle* h, const int32_t* keys, int n, long B, int32_t* list, hipStream_t stream) {
  hipLaunchKernelGGL(kf_vg_order_kernel, dim3(1), dim3(1024), 0, stream, keys, n, (int)B, 4 * (h->n_cu > 0 ? h->n_cu : 256), list,
                     (int32_t*)nullptr, 0, 0);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

extern "C" int kf_filter_value_and_grad(kf_handle* h, const kf_dims* dims, const double* y, const double* a0,
                                        const double* P0, const double* T, const double* Z, const double* R,
                                        const double* H, const double* Q, const double* c, const double* d,
                                        const double* g_logp, double* logp, int32_t* info, double* g_a0, double* g_P0,
                                        double* g_T, double* g_Z, double* g_R, double* g_H, double* g_Q, double* g_c,
                                        double* g_d, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_dims(h, dims, true);
  if (rc) return rc;
  size_t need = 0;
  rc = kf_workspace_bytes_for(h, dims, KF_WS_VALUE_AND_GRAD, &need);
  if (rc) return rc;
  if (!workspace || workspace_bytes < need) {
    KF_SET_ERR(h, "kf_filter_value_and_grad: needs %zu bytes of workspace (got %zu)", need, workspace_bytes);
    return KF_ERR_ARG;
  }
  const size_t B = (size_t)dims->B, n1 = (size_t)dims->T + 1, m = (size_t)dims->m;
  char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
  double* ones = reinterpret_cast<double*>(w);
  w += kf_vg_round(8 * B);
  double* pred_a = reinterpret_cast<double*>(w);
  w += kf_vg_round(8 * B * n1 * m);
  double* pred_P = reinterpret_cast<double*>(w);
  w += kf_vg_round(8 * B * n1 * m * m);
  const size_t inner = workspace_bytes - (size_t)(w - reinterpret_cast<char*>(workspace));
  const bool tile = kf_vg_tile_path(h, dims);
  if (!y || !a0 || !P0 || !T || !Z || !R || !H || !Q || !logp) {
    KF_SET_ERR(h, "kf_filter_value_and_grad: y, a0, P0, T, Z, R, H, Q and logp must be non-NULL");
    return KF_ERR_ARG;
  }
  if (kf_vg_whiten_ok(h, dims)) {
    // Dense H, p > 1 (round 5; VERDICT round 4, missing #4: C4 with a dense H took 415 ms on the LDS adjoint against 14.3 ms with a
    // diagonal one).  Whiten per draw (kf_forward_whiten.hip), run THIS entry point on the whitened model with H* = I flagged diagonal --
    // its adjoints also leave Gamma~ = sum_t vb_t y~_t' --, and map the gradients back (kf_unwhiten_grad_kernel: the full p x p g_H
    // from the invariance of the joint likelihood).  A partially missing row or an H that is not positive definite raises the device
    // flag: the dense LDS kernels behind, launched with run_if = flag, then redo the batch (they exit at once otherwise).
    KF_ON_DEVICE(h);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    KfWhitenGrad W;
    const size_t used = kf_whiten_grad_carve(base, dims->B, dims->T, dims->p, dims->m, &W);
    const size_t slack = (size_t)(base - reinterpret_cast<char*>(workspace));
    KfFwdArgs f;
    f.B = dims->B; f.T = dims->T; f.p = dims->p; f.m = dims->m; f.r = dims->r; f.filter = dims->filter; f.flags = dims->flags;
    f.y = y; f.a0 = a0; f.P0 = P0; f.Tm = T; f.Z = Z; f.R = R; f.H = H; f.Q = Q; f.c = c; f.d = d;
    f.logp = logp; f.ll_obs = nullptr; f.filt_a = nullptr; f.pred_a = nullptr; f.filt_P = nullptr; f.pred_P = nullptr;
    f.info = info; f.tile_tape = nullptr;
    rc = kf_launch_whiten_for_grad(h, f, W, (hipStream_t)stream);
    if (rc) return rc;
    kf_dims d2 = *dims;
    d2.flags |= KF_FLAG_H_DIAGONAL | KF_FLAG_Y_PER_DRAW;
    h->vg_gyy = W.gyy;
    rc = kf_filter_value_and_grad(h, &d2, W.ys, a0, P0, T, W.Zs, R, W.Hs, Q, c, W.ds, g_logp, logp, info, g_a0, g_P0, g_T, W.gZs,
                                  g_R, W.gHs, g_Q, g_c, W.gds, base + used, workspace_bytes - slack - used, stream);
    h->vg_gyy = nullptr;
    if (rc) return rc;
    rc = kf_launch_unwhiten_grad(h, dims->B, dims->p, dims->m, W, g_logp, logp, g_Z, g_H, g_d, (hipStream_t)stream);
    if (rc) return rc;
    // the dense redo (flag set): [flag 256 B | ones | pred_a | pred_P]
    char* w2 = base + 256;
    double* ones2 = reinterpret_cast<double*>(w2); w2 += kf_vg_round(8 * B);
    double* pa2 = reinterpret_cast<double*>(w2); w2 += kf_vg_round(8 * B * n1 * m);
    double* pP2 = reinterpret_cast<double*>(w2);
    KfFwdArgs fd = f;
    fd.pred_a = pa2; fd.pred_P = pP2; fd.run_if = W.flag;
    rc = kf_launch_forward_lds(h, fd, (hipStream_t)stream);
    if (rc) return rc;
    KfBwdArgs ad;
    ad.B = dims->B; ad.T = dims->T; ad.p = dims->p; ad.m = dims->m; ad.r = dims->r; ad.filter = dims->filter; ad.flags = dims->flags;
    ad.y = y; ad.a0 = a0; ad.P0 = P0; ad.Tm = T; ad.Z = Z; ad.R = R; ad.H = H; ad.Q = Q; ad.c = c; ad.d = d;
    ad.pred_a = pa2; ad.pred_P = pP2; ad.g_logp = g_logp; ad.g_ll_obs = nullptr;
    ad.g_a0 = g_a0; ad.g_P0 = g_P0; ad.g_T = g_T; ad.g_Z = g_Z; ad.g_R = g_R; ad.g_H = g_H; ad.g_Q = g_Q; ad.g_c = g_c; ad.g_d = g_d;
    ad.tape_T = dims->T; ad.y_T = dims->T; ad.tile_tape = nullptr; ad.init_ab = nullptr; ad.init_Pb = nullptr; ad.accumulate = 0;
    ad.run_if = W.flag;
    if (!g_logp) {   // (ones2 is read by the redo only, behind the kernels that used the region before)
      hipLaunchKernelGGL(kf_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ones2, (long)B, 1.0);
      KF_HIP_CHECK(h, hipGetLastError());
      ad.g_logp = ones2;
    }
    return kf_launch_backward_lds(h, ad, (hipStream_t)stream);
  }
  if (tile) {
    // [ones | tile tape | inner]: the forward kernel leaves the predicted tile of every step, the adjoint reads it back
    w = reinterpret_cast<char*>(pred_a);
    const bool tsteady = kf_vg_tile_steady_path(h, dims);
    int32_t *ts = nullptr, *order = nullptr, *order_f = nullptr, *hdr = nullptr;
    if (tsteady) {  // [ones | ts B | header | order B | forward order B | tile tape | ...]: ahead of the tape, at offsets that depend on B only
      ts = reinterpret_cast<int32_t*>(w); w += kf_vg_round(4 * B);
      hdr = reinterpret_cast<int32_t*>(w); w += kf_vg_round(64);
      order = reinterpret_cast<int32_t*>(w); w += kf_vg_round(4 * B);
      order_f = reinterpret_cast<int32_t*>(w); w += kf_vg_round(4 * B);
    }
    double* tape = reinterpret_cast<double*>(w);
    const size_t n4 = 4 * ((m + 2 + 3) / 4);
    w += kf_vg_round(8 * B * n1 * n4 * n4);
    const size_t inner_t = workspace_bytes - (size_t)(w - reinterpret_cast<char*>(workspace));
    KF_ON_DEVICE(h);
    KfFwdArgs f;
    f.B = dims->B; f.T = dims->T; f.p = dims->p; f.m = dims->m; f.r = dims->r; f.filter = dims->filter; f.flags = dims->flags;
    f.y = y; f.a0 = a0; f.P0 = P0; f.Tm = T; f.Z = Z; f.R = R; f.H = H; f.Q = Q; f.c = c; f.d = d;
    f.logp = logp; f.ll_obs = nullptr; f.filt_a = nullptr; f.pred_a = nullptr; f.filt_P = nullptr; f.pred_P = nullptr;
    f.info = info; f.tile_tape = tape;
    double *s_pa = nullptr, *s_ab = nullptr, *s_Pb = nullptr, *s_P = nullptr, *s_Pc = nullptr;
    if (tsteady) {  // [.. | pred_a B (T+1) m | lam_ts B m | P_b B m^2 | P B m^2 | P as the forward left it B m^2 | inner]
      s_pa = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * n1 * m);
      s_ab = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * m);
      s_Pb = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * m * m);
      s_P = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * m * m);
      s_Pc = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * m * m);
      f.pred_a = s_pa; f.hs_ts = ts; f.vg_P = s_Pc;
      f.vg_stop = kf_backward_steady_scan_ok(dims->m, dims->p) ? 1 : 0;   // the steps behind ts: forward AND backward in the scan kernel
      // placement of the forward launch from the previous call's convergence points (kf_vg_order_kernel; KF_VG_SCHED_HINT=0: natural order)
      const char* sh = kf_env_switch("KF_VG_SCHED_HINT");
      if (dims->B <= 0x7fffffffL && !(sh && sh[0] == '0')) {
        hipLaunchKernelGGL(kf_vg_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ts, (int)dims->T, (int)dims->B,
                           4 * (h->n_cu > 0 ? h->n_cu : 256), order_f, hdr, 1, (int)dims->m);
        KF_HIP_CHECK(h, hipGetLastError());
        f.order = order_f;
      }
    }
    const size_t inner_ts = workspace_bytes - (size_t)(w - reinterpret_cast<char*>(workspace));
    rc = kf_launch_forward_b4(h, f, nullptr, 0, (hipStream_t)stream);
    if (rc) return rc;
    if (!g_logp) {
      hipLaunchKernelGGL(kf_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ones,
                         (long)B, 1.0);
      KF_HIP_CHECK(h, hipGetLastError());
      g_logp = ones;
    }
    if (tsteady) {
      KfBwdArgs a;
      a.B = dims->B; a.T = dims->T; a.p = dims->p; a.m = dims->m; a.r = dims->r; a.filter = dims->filter; a.flags = dims->flags;
      a.y = y; a.a0 = a0; a.P0 = P0; a.Tm = T; a.Z = Z; a.R = R; a.H = H; a.Q = Q; a.c = c; a.d = d;
      a.pred_a = s_pa; a.pred_P = nullptr; a.g_logp = g_logp; a.g_ll_obs = nullptr;
      a.g_a0 = g_a0; a.g_P0 = g_P0; a.g_T = g_T; a.g_Z = g_Z; a.g_R = g_R; a.g_H = g_H; a.g_Q = g_Q; a.g_c = g_c; a.g_d = g_d;
      a.tape_T = dims->T; a.y_T = dims->T; a.tile_tape = nullptr; a.init_ab = nullptr; a.init_Pb = nullptr; a.accumulate = 0;
      if (dims->B <= 0x7fffffffL) {  // the launch order of the two sweeps (see kf_vg_order_kernel)
        hipLaunchKernelGGL(kf_vg_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ts, (int)dims->T, (int)dims->B,
                           4 * (h->n_cu > 0 ? h->n_cu : 256), order, hdr, 0, (int)dims->m);
        KF_HIP_CHECK(h, hipGetLastError());
        a.list = order;
      }
      if (f.vg_stop) {
        a.fwd_pred_a = s_pa; a.fwd_logp = logp; a.fwd_info = info;
      }
      rc = kf_launch_backward_steady(h, a, ts, s_ab, s_Pb, s_P, (hipStream_t)stream, s_Pc);  // writes every gradient (zeros where it has none)
      if (rc) return rc;
      rc = kf_launch_dare_backward(h, dims->B, dims->m, dims->p, dims->r, T, Z, R, H, Q, s_P, s_Pb, g_T, g_Z, g_R, g_H, g_Q,
                                   /*accumulate=*/1, nullptr, (hipStream_t)stream, /*h_diag_only=*/1);
      if (rc) return rc;
      a.pred_a = nullptr; a.tile_tape = tape; a.steps = ts; a.init_ab = s_ab; a.accumulate = 1;
      return kf_launch_backward_b4(h, a, w, inner_ts, (hipStream_t)stream);
    }
    KfBwdArgs a;
    a.B = dims->B; a.T = dims->T; a.p = dims->p; a.m = dims->m; a.r = dims->r; a.filter = dims->filter; a.flags = dims->flags;
    a.y = y; a.a0 = a0; a.P0 = P0; a.Tm = T; a.Z = Z; a.R = R; a.H = H; a.Q = Q; a.c = c; a.d = d;
    a.pred_a = nullptr; a.pred_P = nullptr; a.g_logp = g_logp; a.g_ll_obs = nullptr;
    a.g_a0 = g_a0; a.g_P0 = g_P0; a.g_T = g_T; a.g_Z = g_Z; a.g_R = g_R; a.g_H = g_H; a.g_Q = g_Q; a.g_c = g_c; a.g_d = g_d;
    a.tape_T = dims->T; a.y_T = dims->T; a.tile_tape = tape; a.init_ab = nullptr; a.init_Pb = nullptr; a.accumulate = 0;
    return kf_launch_backward_b4(h, a, w, inner_t, (hipStream_t)stream);
  }
  if (kf_vg_steady_path(h, dims)) {
    // Forward kernel with a steady-state phase (the multi-observation block kernel: C4): it reports where the phase starts, the
    // steady-state adjoint (kf_backward_steady.hip: a vector recursion + one Riccati adjoint per draw) takes the steps behind that
    // point, the exact adjoint the steps before it.  [.. | ts B | lam_ts B m | P_b B m^2 | P B m^2 | inner]
    KF_ON_DEVICE(h);
    int32_t* ts = reinterpret_cast<int32_t*>(w);
    w += kf_vg_round(4 * B);
    double* s_ab = reinterpret_cast<double*>(w);
    w += kf_vg_round(8 * B * m);
    double* s_Pb = reinterpret_cast<double*>(w);
    w += kf_vg_round(8 * B * m * m);
    double* s_P = reinterpret_cast<double*>(w);
    w += kf_vg_round(8 * B * m * m);
    const size_t inner_s = workspace_bytes - (size_t)(w - reinterpret_cast<char*>(workspace));
    KfFwdArgs f;
    f.B = dims->B; f.T = dims->T; f.p = dims->p; f.m = dims->m; f.r = dims->r; f.filter = dims->filter; f.flags = dims->flags;
    f.y = y; f.a0 = a0; f.P0 = P0; f.Tm = T; f.Z = Z; f.R = R; f.H = H; f.Q = Q; f.c = c; f.d = d;
    f.logp = logp; f.ll_obs = nullptr; f.filt_a = nullptr; f.pred_a = pred_a; f.filt_P = nullptr; f.pred_P = pred_P;
    f.info = info; f.tile_tape = nullptr; f.hs_ts = ts; f.skip_steady_cov = 1;
    rc = kf_launch_forward_b4q(h, f, w, inner_s, (hipStream_t)stream);
    if (rc) return rc;
    if (!g_logp) {
      hipLaunchKernelGGL(kf_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ones,
                         (long)B, 1.0);
      KF_HIP_CHECK(h, hipGetLastError());
      g_logp = ones;
    }
    KfBwdArgs a;
    a.B = dims->B; a.T = dims->T; a.p = dims->p; a.m = dims->m; a.r = dims->r; a.filter = dims->filter; a.flags = dims->flags;
    a.y = y; a.a0 = a0; a.P0 = P0; a.Tm = T; a.Z = Z; a.R = R; a.H = H; a.Q = Q; a.c = c; a.d = d;
    a.pred_a = pred_a; a.pred_P = pred_P; a.g_logp = g_logp; a.g_ll_obs = nullptr;
    a.g_a0 = g_a0; a.g_P0 = g_P0; a.g_T = g_T; a.g_Z = g_Z; a.g_R = g_R; a.g_H = g_H; a.g_Q = g_Q; a.g_c = g_c; a.g_d = g_d;
    a.tape_T = dims->T; a.y_T = dims->T; a.tile_tape = nullptr; a.init_ab = nullptr; a.init_Pb = nullptr; a.accumulate = 0;
    a.g_yy = h->vg_gyy;   // (the whitened call of a dense-H model: both adjoints below add their share)
    rc = kf_launch_backward_steady(h, a, ts, s_ab, s_Pb, s_P, (hipStream_t)stream);   // writes every gradient (zeros where it has none)
    if (rc) return rc;
    rc = kf_launch_dare_backward(h, dims->B, dims->m, dims->p, dims->r, T, Z, R, H, Q, s_P, s_Pb, g_T, g_Z, g_R, g_H, g_Q,
                                 /*accumulate=*/1, nullptr, (hipStream_t)stream, /*h_diag_only=*/1);
    if (rc) return rc;
    a.steps = ts;
    a.init_ab = s_ab;
    a.accumulate = 1;
    return kf_launch_backward(h, a, w, inner_s, (hipStream_t)stream);
  }
  if (kf_vg_lane_steady_path(h, dims)) {
    // [.. | ts B | order B | lam_ts B m | P_b B m^2 | P B m^2 | the block adjoint's hand-over | inner]
    KF_ON_DEVICE(h);
    int32_t* ts = reinterpret_cast<int32_t*>(w); w += kf_vg_round(4 * B);
    int32_t* order = reinterpret_cast<int32_t*>(w); w += kf_vg_round(4 * B);
    double* s_ab = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * m);
    double* s_Pb = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * m * m);
    double* s_P = reinterpret_cast<double*>(w); w += kf_vg_round(8 * B * m * m);
    const size_t inner_l = workspace_bytes - (size_t)(w - reinterpret_cast<char*>(workspace));
    KfFwdArgs f;
    f.B = dims->B; f.T = dims->T; f.p = dims->p; f.m = dims->m; f.r = dims->r; f.filter = dims->filter; f.flags = dims->flags;
    f.y = y; f.a0 = a0; f.P0 = P0; f.Tm = T; f.Z = Z; f.R = R; f.H = H; f.Q = Q; f.c = c; f.d = d;
    f.logp = logp; f.ll_obs = nullptr; f.filt_a = nullptr; f.pred_a = pred_a; f.filt_P = nullptr; f.pred_P = pred_P;
    f.info = info; f.tile_tape = nullptr; f.hs_ts = ts;
    switch (dims->m) {
      case 1: rc = kf_launch_forward_lane_m1(h, f, (hipStream_t)stream); break;
      case 2: rc = kf_launch_forward_lane_m2(h, f, (hipStream_t)stream); break;
      case 3: rc = kf_launch_forward_lane_m3(h, f, (hipStream_t)stream); break;
      default: rc = kf_launch_forward_lane_m4(h, f, (hipStream_t)stream); break;
    }
    if (rc) return rc;
    if (!g_logp) {
      hipLaunchKernelGGL(kf_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ones,
                         (long)B, 1.0);
      KF_HIP_CHECK(h, hipGetLastError());
      g_logp = ones;
    }
    KfBwdArgs a;
    a.B = dims->B; a.T = dims->T; a.p = dims->p; a.m = dims->m; a.r = dims->r; a.filter = dims->filter; a.flags = dims->flags;
    a.y = y; a.a0 = a0; a.P0 = P0; a.Tm = T; a.Z = Z; a.R = R; a.H = H; a.Q = Q; a.c = c; a.d = d;
    a.pred_a = pred_a; a.pred_P = pred_P; a.g_logp = g_logp; a.g_ll_obs = nullptr;
    a.g_a0 = g_a0; a.g_P0 = g_P0; a.g_T = g_T; a.g_Z = g_Z; a.g_R = g_R; a.g_H = g_H; a.g_Q = g_Q; a.g_c = g_c; a.g_d = g_d;
    a.tape_T = dims->T; a.y_T = dims->T; a.tile_tape = nullptr; a.init_ab = nullptr; a.init_Pb = nullptr; a.accumulate = 0;
    hipLaunchKernelGGL(kf_vg_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ts, (int)dims->T, (int)dims->B,
                       4 * (h->n_cu > 0 ? h->n_cu : 256), order, (int32_t*)nullptr, 0, 0);
    KF_HIP_CHECK(h, hipGetLastError());
    a.list = order;
    rc = kf_launch_backward_steady(h, a, ts, s_ab, s_Pb, s_P, (hipStream_t)stream);   // writes every gradient (zeros where it has none)
    if (rc) return rc;
    rc = kf_launch_dare_backward(h, dims->B, dims->m, dims->p, dims->r, T, Z, R, H, Q, s_P, s_Pb, g_T, g_Z, g_R, g_H, g_Q,
                                 /*accumulate=*/1, nullptr, (hipStream_t)stream, /*h_diag_only=*/1);
    if (rc) return rc;
    a.steps = ts; a.init_ab = s_ab; a.accumulate = 1;
    return kf_launch_backward_b4(h, a, w, inner_l, (hipStream_t)stream);
  }
  rc = kf_filter_forward(h, dims, y, a0, P0, T, Z, R, H, Q, c, d, logp, nullptr, nullptr, pred_a, nullptr, pred_P, info,
                         w, inner, stream);
  if (rc) return rc;
  if (!g_logp) {
    KF_ON_DEVICE(h);
    hipLaunchKernelGGL(kf_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ones,
                       (long)B, 1.0);
    KF_HIP_CHECK(h, hipGetLastError());
    g_logp = ones;
  }
  return kf_filter_backward(h, dims, y, a0, P0, T, Z, R, H, Q, c, d, pred_a, pred_P, g_logp, nullptr, g_a0, g_P0, g_T,
                            g_Z, g_R, g_H, g_Q, g_c, g_d, w, inner, stream);
}

extern "C" int kf_smoother(kf_handle* h, const kf_dims* dims, const double* T, const double* R,
                           const double* Q, const double* filt_a, const double* filt_P,
                           double* smooth_a, double* smooth_P, void* workspace,
                           size_t workspace_bytes, void* stream) {
  int rc = check_dims(h, dims, false);
  if (rc) return rc;
  if (!T || !R || !Q || !filt_a || !filt_P || !smooth_a || !smooth_P) {
    KF_SET_ERR(h, "kf_smoother: all pointers must be non-NULL");
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  KfSmoothArgs a;
  a.B = dims->B; a.T = dims->T; a.m = dims->m; a.r = dims->r;
  a.flags = dims->flags & (KF_FLAG_TV_T | KF_FLAG_TV_R | KF_FLAG_TV_Q);
  a.Tm = T; a.R = R; a.Q = Q; a.filt_a = filt_a; a.filt_P = filt_P;
  a.smooth_a = smooth_a; a.smooth_P = smooth_P;
  a.only_flagged = nullptr;
  a.exact = (dims->flags & KF_FLAG_EXACT_RECURSION) ? 1 : 0;
  {
    const char* e = kf_env_switch("KF_SMOOTHER_STEADY");  // =0: as KF_FLAG_EXACT_RECURSION (A/B measurements)
    if (e && e[0] == '0') a.exact = 1;
  }
  if (a.flags) {
    a.flags |= dims->flags & KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING;
    // time-varying T / R / Q (kalman_smoother.py:17-54,62-77).  Round 5: 5 <= m <= 16 on the register-tile elimination kernel with
    // per-step slices (kf_smooth_mfma_kernel<.., TV>; draws it flags go to the LDS kernel as for static models);
    // KF_SMOOTHER_TV_MFMA=0 in the environment keeps the LDS kernel, which reloads the slices per step
    if (a.m >= 5 && a.m <= KF_MP_M_MAX && h->forward_kernel != KF_KERNEL_LDS && workspace &&
        workspace_bytes >= kf_smoother_mfma_workspace_bytes(a.B)) {
      const char* e = kf_env_switch("KF_SMOOTHER_TV_MFMA");
      if (!(e && e[0] == '0')) return kf_launch_smoother_mfma(h, a, workspace, workspace_bytes, (hipStream_t)stream);
    }
    if (!kf_smoother_lds_fits(a.m)) {
      KF_SET_ERR(h, "kf_smoother: time-varying system matrices need the LDS-resident kernel (m=%d too large)", a.m);
      return KF_ERR_UNSUPPORTED;
    }
    return kf_launch_smoother_lds(h, a, (hipStream_t)stream);
  }
  // m <= 4 (C1, C2): one lane per draw, pseudo-inverse in registers (kf_smoother_lane.hip); KF_SMOOTHER_LANE=0 keeps the LDS kernel
  if (kf_smoother_lane_ok(a.m) && h->forward_kernel != KF_KERNEL_LDS) {
    const char* e = kf_env_switch("KF_SMOOTHER_LANE");
    if (!(e && e[0] == '0')) return kf_launch_smoother_lane(h, a, (hipStream_t)stream);
  }
  // 5 <= m <= 15: block-MFMA smoother with warm-started Newton-Schulz inverses (kf_smoother_b4.hip); KF_SMOOTHER_B4=0
  // in the environment keeps the elimination kernel below (A/B measurements)
  if (kf_smoother_b4_ok(a.m, a.r) && h->forward_kernel != KF_KERNEL_LDS && workspace &&
      workspace_bytes >= kf_smoother_mfma_workspace_bytes(a.B)) {
    // measured (scripts/r02/smooth_sweep.py, 1000 steps, ms, against the elimination kernel): 4096 draws m = 5 / 7 / 8 / 10 / 11:
    // 2.3 / 2.5 / 4.1 / 4.8 / 5.1 against 3.8 / 4.5 / 5.3 / 6.6 / 7.1; m = 12 .. 15 (4 x 4 blocks: three wavefronts per SIMD,
    // two rounds at 4096 draws) 9.3 - 10.3 against 7.4 - 10.3, but 3.0 - 3.2 against 3.6 - 4.6 at 512 draws and 4.0 against 7.1
    // on the C4 shard (m = 15, 2048 draws).  Re-measured at the end of round 2 (scripts/r02/smooth_sweep_m12.py, smooth_c4.py; 500 steps,
    // dense random models, m = 12 / 13 / 14 / 15): 4096 draws 4.4 / 4.0 / 4.3 / 4.4 against 4.0 / 4.8 / 5.0 / 5.4, 8192 draws 6.5 / 7.2 / 7.3 /
    // 7.9 against 7.1 / 8.7 / 9.1 / 9.7, 16384 draws 11.9 / 12.5 / 13.3 / 16.5 against 13.3 / 16.5 / 16.9 / 17.9; C4 (VARMA, m = 15, 1000
    // steps) 16384 / 8192 / 4096 draws: 22.1 / 12.2 / 6.9 against 35.5 / 18.6 / 10.0 -- the one shape left to the elimination kernel is
    // m = 12 between 3072 and 8192 draws
    const char* e = kf_env_switch("KF_SMOOTHER_B4");
    const long wave3 = 3L * 4 * (h->n_cu > 0 ? h->n_cu : 256);
    // round 5 (scripts/r05/smooth_dispatch_rn.py, profiles/r05/smoother_dispatch_rank.txt): with rank-one state noise from m = 6 on,
    // and with rank-two noise on short series (m >= 8, <= 512 steps), the warm-started iteration needs more steps than the
    // elimination costs (4096 draws x 300 steps, m = 8, r = 2: 2.3 against 1.6 ms; r = 1: 14.4 against 11.3)
    const bool low_rank = (a.r == 1 && a.m >= 6) || (a.r == 2 && a.m >= 8 && a.T <= 512);
    const bool pick = (e && e[0]) ? e[0] != '0' : ((a.m != 12 || a.B <= wave3 || a.B >= 8192) && !low_rank);
    if (pick) return kf_launch_smoother_b4(h, a, workspace, workspace_bytes, (hipStream_t)stream);
  }
  // 5 <= m <= 16: register-tile smoother when the caller provided the (tiny) flag workspace
  // (measured: m = 10, B = 4096: 6.5 vs 17.1 ms; C4 shard m = 15: 6.1 vs 25.7 ms; m = 2 stays on the
  // LDS kernel, 3.4 ms)
  if (a.m >= 5 && a.m <= KF_MP_M_MAX && h->forward_kernel != KF_KERNEL_LDS && workspace &&
      workspace_bytes >= kf_smoother_mfma_workspace_bytes(a.B))
    return kf_launch_smoother_mfma(h, a, workspace, workspace_bytes, (hipStream_t)stream);
  // 16 < m <= 47: block-tile smoother (measured at 4096 draws x 1000 steps: the LDS smoother took
  // 260 ms at m = 20, 4.9 s at m = 40)
  if (kf_smoother_mfma_nb_ok(a.m, a.r) && h->forward_kernel != KF_KERNEL_LDS && workspace &&
      workspace_bytes >= kf_smoother_mfma_workspace_bytes(a.B))
    return kf_launch_smoother_mfma_nb(h, a, workspace, workspace_bytes, (hipStream_t)stream);
  // 47 < m <= 64: four wavefronts per draw (measured: 1.2 - 1.6 s per 1000 steps at 4096 draws on the
  // workspace path)
  if (kf_smoother_mfma_wg_ok(a.m, a.r) && h->forward_kernel != KF_KERNEL_LDS)
    return kf_launch_smoother_mfma_wg(h, a, workspace, workspace_bytes, (hipStream_t)stream);
  if (!kf_smoother_lds_fits(a.m))
    return kf_launch_smoother_big(h, a, workspace, workspace_bytes, (hipStream_t)stream);
  return kf_launch_smoother_lds(h, a, (hipStream_t)stream);
}

extern "C" int kf_smoother_from_filter(kf_handle* h, const kf_dims* dims, const double* y, const double* T, const double* Z,
                                       const double* H, const double* d, const double* pred_a, const double* pred_P,
                                       double* smooth_a, double* smooth_P, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_dims(h, dims, true);
  if (rc) return rc;
  if (!y || !T || !Z || !H || !pred_a || !pred_P || !smooth_a || !smooth_P) {
    KF_SET_ERR(h, "kf_smoother_from_filter: all pointers but d must be non-NULL");
    return KF_ERR_ARG;
  }
  const unsigned fl = dims->flags & ~(unsigned)(KF_FLAG_EXACT_RECURSION | KF_FLAG_STEADY_STATE);
  const bool big = kf_smoother_bf_big_ok(dims->m, dims->p, fl);
  if (!big && !kf_smoother_bf_ok(dims->m, dims->p, fl)) {
    KF_SET_ERR(h, "kf_smoother_from_filter: needs p == 1, m <= 1024 and static matrices (p=%d m=%d flags=0x%x); use kf_smoother",
               dims->p, dims->m, dims->flags);
    return KF_ERR_UNSUPPORTED;
  }
  KF_ON_DEVICE(h);
  KfSmoothBfArgs a;
  a.B = dims->B; a.T = dims->T; a.m = dims->m; a.filter = dims->filter;
  a.flags = dims->flags & KF_FLAG_Y_PER_DRAW;
  a.exact = (dims->flags & KF_FLAG_EXACT_RECURSION) ? 1 : 0;
  {
    const char* e = kf_env_switch("KF_SMOOTHER_STEADY");  // =0: as KF_FLAG_EXACT_RECURSION (A/B measurements), like kf_smoother
    if (e && e[0] == '0') a.exact = 1;
  }
  a.y = y; a.Tm = T; a.Z = Z; a.H = H; a.d = d; a.pred_a = pred_a; a.pred_P = pred_P;
  a.smooth_a = smooth_a; a.smooth_P = smooth_P;
  // 16 <= m <= 79: the recursion on chip, 16 x 16 fp64 MFMA tiles, one workgroup per draw (round 6, kf_smoother_bf_wg.hip);
  // KF_SMOOTHER_BF_WG=0 keeps the HBM-workspace path (A/B measurements)
  if (kf_smoother_bf_wg_ok(a.m, dims->p, fl)) {
    const char* e = kf_env_switch("KF_SMOOTHER_BF_WG");
    if (!(e && e[0] == '0')) return kf_launch_smoother_bf_wg(h, a, (hipStream_t)stream);
  }
  if (big) return kf_launch_smoother_bf_big(h, a, workspace, workspace_bytes, (hipStream_t)stream);
  // m <= 4 (C1, C2): one lane per draw (kf_smoother_bf_lane.hip); KF_SMOOTHER_BF_LANE=0 keeps the block-tile kernel (A/B measurements)
  if (kf_smoother_bf_lane_ok(a.m, dims->p, fl) && h->forward_kernel != KF_KERNEL_B4) {
    const char* e = kf_env_switch("KF_SMOOTHER_BF_LANE");
    if (!(e && e[0] == '0')) return kf_launch_smoother_bf_lane(h, a, (hipStream_t)stream);
  }
  return kf_launch_smoother_bf(h, a, (hipStream_t)stream);
}

extern "C" int kf_assemble(kf_handle* h, int64_t B, int32_t n_theta, const double* theta, int64_t total,
                           const double* tmpl, int32_t n_map, const int32_t* dst, const int32_t* src,
                           double* out, void* stream) {
  if (!h) return KF_ERR_ARG;
  if (B < 1 || n_theta < 0 || total < 1 || n_map < 0 || !tmpl || !out || (n_map > 0 && (!theta || !dst || !src))) {
    KF_SET_ERR(h, "kf_assemble: bad arguments (B=%ld n_theta=%d total=%ld n_map=%d)", (long)B, n_theta,
               (long)total, n_map);
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  return kf_launch_assemble(h, B, n_theta, theta, total, tmpl, n_map, dst, src, out, (hipStream_t)stream);
}

extern "C" int kf_lyapunov(kf_handle* h, int64_t B, int32_t m, int32_t r, const double* T, const double* R,
                           const double* Q, const double* S, double* P, int32_t* info, void* stream) {
  if (!h) return KF_ERR_ARG;
  if (B < 1 || m < 1 || !T || !P || (!S && (!R || !Q || r < 1))) {
    KF_SET_ERR(h, "kf_lyapunov: need T, P and either S or (R, Q, r >= 1)");
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  return kf_launch_lyapunov(h, B, m, r, T, R, Q, S, P, info, (hipStream_t)stream);
}

extern "C" int kf_conditional_simulation(kf_handle* h, int64_t B, int32_t n, int32_t k, int32_t n_sims,
                                         const double* mu, const double* cov, const double* z,
                                         double jitter, double* out, int32_t* info, void* stream) {
  if (!h) return KF_ERR_ARG;
  if (B < 1 || n < 1 || k < 1 || n_sims < 1 || !mu || !cov || !z || !out || !(jitter >= 0.0)) {
    KF_SET_ERR(h, "kf_conditional_simulation: bad arguments (B=%ld n=%d k=%d n_sims=%d)", (long)B, n, k,
               n_sims);
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  return kf_launch_simulate(h, B, n, k, n_sims, mu, cov, z, jitter, out, info, (hipStream_t)stream);
}

extern "C" int kf_dare(kf_handle* h, int64_t B, int32_t m, int32_t p, int32_t r, const double* T, const double* Z,
                       const double* R, const double* H, const double* Q, const double* P_init, double* P,
                       int32_t* info, int32_t max_iter, double tol, void* stream) {
  if (!h) return KF_ERR_ARG;
  if (B < 1 || m < 1 || p < 1 || r < 1 || !T || !Z || !R || !H || !Q || !P || max_iter < 1 || !(tol >= 0.0)) {
    KF_SET_ERR(h, "kf_dare: bad arguments (B=%ld m=%d p=%d r=%d max_iter=%d)", (long)B, m, p, r, max_iter);
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  return kf_launch_dare(h, B, m, p, r, T, Z, R, H, Q, P_init, P, info, max_iter, tol, (hipStream_t)stream);
}

extern "C" int kf_dare_backward(kf_handle* h, int64_t B, int32_t m, int32_t p, int32_t r, const double* T,
                                const double* Z, const double* R, const double* H, const double* Q, const double* P,
                                const double* g_P, double* g_T, double* g_Z, double* g_R, double* g_H, double* g_Q,
                                int32_t accumulate, int32_t* info, void* stream) {
  if (!h) return KF_ERR_ARG;
  if (B < 1 || m < 1 || p < 1 || r < 1 || !T || !Z || !R || !H || !Q || !P || !g_P) {
    KF_SET_ERR(h, "kf_dare_backward: bad arguments (B=%ld m=%d p=%d r=%d)", (long)B, m, p, r);
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  return kf_launch_dare_backward(h, B, m, p, r, T, Z, R, H, Q, P, g_P, g_T, g_Z, g_R, g_H, g_Q, accumulate, info,
                                 (hipStream_t)stream);
}

extern "C" int kf_simulate_statespace(kf_handle* h, int64_t B, int32_t n_sims, int32_t n, int32_t p, int32_t m,
                                      int32_t r, const double* T, const double* Z, const double* R, const double* H,
                                      const double* Q, const double* x0, const double* z_state, const double* z_obs,
                                      double* states, double* obs, int32_t* info, void* stream) {
  if (!h) return KF_ERR_ARG;
  if (B < 1 || n_sims < 1 || n < 1 || p < 1 || m < 1 || r < 1 || !T || !Z || !R || !H || !Q || !z_state || !z_obs ||
      !states || !obs) {
    KF_SET_ERR(h, "kf_simulate_statespace: bad arguments (B=%ld n_sims=%d n=%d p=%d m=%d r=%d)", (long)B, n_sims, n,
               p, m, r);
    return KF_ERR_ARG;
  }
  KF_ON_DEVICE(h);
  return kf_launch_simulate_statespace(h, B, n_sims, n, p, m, r, T, Z, R, H, Q, x0, z_state, z_obs, states, obs, info,
                                       (hipStream_t)stream);
}
