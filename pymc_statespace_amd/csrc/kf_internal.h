// kf_internal.h -- shared declarations of the HIP implementation behind include/kf.h.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/kf.h"

#define KF_WAVE 64
#define KF_LDS_BYTES_MAX (160 * 1024)
#define KF_LOG_2PI 1.8378770664093454835606594728112  // log(2*pi), MVN_CONST kalman_filter.py:16

// hipFuncSetAttribute is per device: remember it per (kernel instantiation, device), not per process
struct KfAttrOnce {
  unsigned long long done = 0;
  bool need(int dev) const { return dev < 0 || dev >= 64 || !((done >> dev) & 1ull); }
  void mark(int dev) {
    if (dev >= 0 && dev < 64) done |= 1ull << dev;
  }
};

// Environment switches (table: INTEGRATION.md).  Two classes, so that nothing in the environment can make an entry point return
// incomplete results or change a documented tolerance:
//   kf_env_switch      documented selectors between COMPLETE, parity-tested paths (KF_STEADY_STATE, KF_B4_SPLIT, KF_FORWARD_B4P,
//                      KF_VG_STEADY, ...): read per call (tests flip them), each value's results are held to parity by
//                      tests/test_gpu_switches.py
//   kf_env_experiment  tuning knobs of the measurement scripts (chunk sizes, workgroup counts, the convergence tolerance): compiled in
//                      only with -DKF_EXPERIMENTS (make XFLAGS=-DKF_EXPERIMENTS); a product build ignores them
// Round 6: a switch set on the HANDLE (kf_set_option) wins over the environment -- a server that holds several handles (models,
// tenants) configures each of them and never touches the process environment; the entry point's KfDeviceGuard publishes its handle
// to the thread for the duration of the call, so the ~30 call sites below the API need no handle argument.
struct kf_handle;
extern thread_local const kf_handle* kf_tls_handle;
const char* kf_handle_option(const kf_handle* h, const char* name);   // NULL: not set on the handle
inline const char* kf_env_switch(const char* name) {
  if (kf_tls_handle) {
    const char* o = kf_handle_option(kf_tls_handle, name);
    if (o) return *o ? o : nullptr;
  }
  const char* e = getenv(name);
  return (e && *e) ? e : nullptr;
}
inline const char* kf_env_experiment(const char* name) {
#ifdef KF_EXPERIMENTS
  return kf_env_switch(name);
#else
  (void)name;
  return nullptr;
#endif
}

#define KF_AUX_EVENTS 38  // 0 .. 33: the split pipelines' own; 34 .. 36: KfAuxJoinGuard
#define KF_MAX_OPTIONS 40
struct KfOption {
  char name[32];
  char value[24];
};
struct kf_handle {
  int device;
  int n_options;                       // kf_set_option: switches set on this handle (kf_env_switch reads them before the environment)
  KfOption options[KF_MAX_OPTIONS];
  int lanes_per_draw;  // 0 = auto
  int forward_kernel;  // KF_KERNEL_AUTO / _LDS / _ROW / _LANE
  int n_cu;
  char err[512];
  // second stream of the split forward pipeline (kf_forward_b4.hip): the steady kernels run beside the recursion chunks.
  // Created on first use (kf_handle_aux), forked from / joined to the caller's stream with events -- the call stays
  // asynchronous on the caller's stream and never synchronises the host
  int aux_ready;
  hipStream_t aux_stream;   // covariance fill
  hipStream_t aux_stream2;  // mean recursion
  hipStream_t aux_stream3;  // pair kernel: the tail of the full recursion
  hipEvent_t aux_ev[KF_AUX_EVENTS];
  // internal (kf_filter_value_and_grad on a dense-H model, kf_api.hip): while the whitened inner call runs, where its adjoints leave
  // sum_t vb_t y~_t' (KfBwdArgs::g_yy); NULL otherwise
  double* vg_gyy;
  // name of the kernel that carried the recursion of the last kf_filter_forward on the HBM-workspace path (m > 64), where the choice
  // depends on the data (T sparse or not); NULL before the first such call.  kf_last_forward_kernel()
  const char* last_forward_kernel;
  // verdict of the last read-back on that path (kf_forward_big.hip): a call that is being CAPTURED into a hipGraph cannot read back, and
  // takes the sparse chain when the last un-captured call on the same (T pointer, B, m) did -- under a device-side guard (the CSR
  // build's ok word): replayed on a T that no longer fits, every draw comes back NaN with info 0x7fffff02 instead of a wrong number
  const double* big_cache_T;
  long big_cache_B;
  int big_cache_m;
  int big_cache_sparse;
};
int kf_handle_aux(kf_handle* h);  // KF_OK once aux_stream / aux_ev exist (current device = the handle's)
// Scope guard of a launcher that forks work onto the handle's streams: whichever way the function is left -- an error return in the
// middle of the chunk loop included -- the caller's stream ends up behind everything the three streams were given, so the caller may
// reuse or free its buffers and an active hipGraph capture is left with every fork joined.  The success path makes
// its own, finer joins and dismisses the guard.
struct KfAuxJoinGuard {
  kf_handle* h;
  hipStream_t stream;
  bool armed;
  KfAuxJoinGuard(kf_handle* h_, hipStream_t s) : h(h_), stream(s), armed(true) {}
  void dismiss() { armed = false; }
  ~KfAuxJoinGuard() {
    if (!armed || !h || !h->aux_ready) return;
    hipStream_t aux[3] = {h->aux_stream, h->aux_stream2, h->aux_stream3};
    for (int i = 0; i < 3; ++i) {
      hipEvent_t ev = h->aux_ev[KF_AUX_EVENTS - 4 + i];
      if (hipEventRecord(ev, aux[i]) == hipSuccess) (void)hipStreamWaitEvent(stream, ev, 0);
    }
  }
  KfAuxJoinGuard(const KfAuxJoinGuard&) = delete;
  KfAuxJoinGuard& operator=(const KfAuxJoinGuard&) = delete;
};

// Arguments of the LDS-resident forward kernel (one struct, passed by value).
struct KfFwdArgs {
  long B;
  int T, p, m, r;
  int filter;
  unsigned flags;
  const double* y;
  const double* a0;
  const double* P0;
  const double* Tm;
  const double* Z;
  const double* R;
  const double* H;
  const double* Q;
  const double* c;
  const double* d;
  double* logp;
  double* ll_obs;
  double* filt_a;
  double* pred_a;
  double* filt_P;
  double* pred_P;
  int32_t* info;
  // internal (kf_filter_value_and_grad): the 4x4x4-block kernel's own tape, [B][T+1][N4][N4] row-major (N4 = 4 ceil((m + 2) / 4)):
  // slot t < T the filtered augmented tile of step t with u_hat in row JA and 1 / sqrt(F) at [JA][JA], slot T the predicted
  // tile of step 1; read back by kf_backward_b4.hip.  NULL for every public call
  double* tile_tape;
  // internal (split pipeline of kf_forward_b4.hip, see kf_forward_b4_steady.hip): this launch runs steps [t_begin, t_end) of the
  // draws list[0 .. *count) (list == NULL: draw = workgroup index, count ignored) and leaves each draw's tile, partial
  // log-likelihood and status in the hand-over buffers; all NULL / 0 for every other kernel
  int t_begin = 0, t_end = 0;
  const int32_t* list = nullptr;
  const int32_t* count = nullptr;
  double* hs_const = nullptr;   // [B][B4_HC_STRIDE] the first chunk's per-draw constants in register layout (T blocks, C_ext)
  double* hs_tile = nullptr;    // [B][NB * 64] register image of the predicted tile at the hand-over step
  double* hs_logp = nullptr;    // [B] log-likelihood of the steps done so far
  int32_t* hs_info = nullptr;   // [B]
  int32_t* hs_ts = nullptr;     // [B] -1: recursion continues in the next chunk, >= 0: converged, steady phase starts at this step, -2: finished
  double ss_tol2 = 1e-24;       // (square of) the entrywise tolerance of the steady-state test, 1e-12 of sqrt(P_ii P_jj)
  int skip_steady_cov = 0;      // internal (kf_filter_value_and_grad): the steady-state phase leaves the constant covariance tapes unwritten
  int32_t* hs_tfree = nullptr;  // [B] first step after the last missing observation (written by the first chunk)
  const int32_t* run_if = nullptr;  // internal (kf_forward_whiten.hip): the LDS kernel exits at once unless *run_if != 0
  double* vg_P = nullptr;       // internal (kf_filter_value_and_grad, block kernel + steady-state adjoint): [B][m][m] the predicted covariance at hs_ts[b]
  const int32_t* order = nullptr;  // internal, with vg_P: workgroup i runs draw order[i] (kf_vg_order_kernel, kf_api.hip: placement only)
  int vg_stop = 0;              // internal, with vg_P: the kernel ends at hs_ts[b] (logp / info up to there, a_ts in pred_a) -- the steps behind it are
                                // kf_bwd_steady_scan_kernel's, forward and backward (kf_backward_steady.hip)
  // internal (round 6, the time-chunked pipeline: kf_forward_b4c.hip): workgroup i of the block kernel's chunk instantiation runs chunk
  // j = i % ck_nc of draw i / ck_nc -- steps [j ck_s + 1, (j + 1) ck_s + 1) (chunk 0: from step 0) -- from the predicted covariance
  // ck_start[b][j] (chunk 0: P0) and, j > 0, the predicted mean c ("zero start": the draw's true mean joins in the mean pass); it writes
  // the covariance tapes and leaves the gain, (F, -2 v / sqrt F) and the chunk's last filtered mean for the mean pass
  int lane_pre = 0;   // kf_forward_lane.hip: the first 256 steps as sixteen 16-step chunks on a slot's sixteen lanes (KF_LANE_PRECHUNK)
  int ck_nc = 0, ck_s = 0;
  const double* ck_start = nullptr;  // [B][ck_nc][m][m]
  double* ck_K = nullptr;            // [B][T][m]      K_t = P_t Z' / F_t (0 on a missing step: never here)
  double* ck_fv = nullptr;           // [B][T][2]      {F_t, -2 v_t / sqrt(F_t)} of the zero-start recursion
  double* ck_zend = nullptr;         // [B][ck_nc][m]  filtered mean of the chunk's last step (zero start)
  int32_t* ck_flag = nullptr;        // [1] set non-zero by any kernel of the pipeline that cannot vouch for its result: the batch is then
                                     // redone by the sequential kernel launched behind (run_if = ck_flag)
  double ck_tol = 1e-12;             // a chunk's last predicted covariance against the next chunk's start, relative to the largest entry
};

struct KfBwdArgs {
  long B;
  int T, p, m, r;
  int filter;
  unsigned flags;
  const double* y;
  const double* a0;
  const double* P0;
  const double* Tm;
  const double* Z;
  const double* R;
  const double* H;
  const double* Q;
  const double* c;
  const double* d;
  const double* pred_a;
  const double* pred_P;
  const double* g_logp;
  const double* g_ll_obs;
  // cotangents of the filter's tape outputs (kf_filter_backward_tapes; NULL = 0): filtered / predicted states [B][T][m] /
  // [B][T+1][m], filtered / predicted covariances [B][T][m][m] / [B][T+1][m][m].  Honoured by the LDS-resident adjoint only
  // (kf_backward_lds.hip): kf_launch_backward routes there when one of them is set
  const double* g_filt_a = nullptr;
  const double* g_pred_a = nullptr;
  const double* g_filt_P = nullptr;
  const double* g_pred_P = nullptr;
  double* g_a0;
  double* g_P0;
  double* g_T;
  double* g_Z;
  double* g_R;
  double* g_H;
  double* g_Q;
  double* g_c;
  double* g_d;
  // continuation of a reverse sweep another kernel started (kf_backward_b4.hip hands its last step to
  // kf_backward_mfma.hip): T counts the steps to run, tape_T is the series length the tapes / y / g_ll_obs are strided by,
  // (init_ab, init_Pb) [B][m], [B][m][m] are the adjoints of the predicted moments after those steps (NULL = 0), and
  // accumulate != 0 adds to g_T, g_Z, g_R, g_H, g_Q, g_c, g_d instead of overwriting them
  int tape_T;
  int y_T;  // series length y / g_ll_obs are strided by (= tape_T except in the tile-tape hand-over)
  const double* tile_tape;  // internal: kf_forward_b4.hip's tile tape instead of (pred_a, pred_P), see KfFwdArgs
  const double* init_ab;
  const double* init_Pb;
  int accumulate;
  // internal (round 5, the gradient of a dense-H model by whitening, kf_api.hip): [B][p][p], sum over the steps of vb_t y~_t' -- the
  // adjoint of the innovation times the observation (y -+ d) -- written (accumulate: added to) by the scalar-observation adjoints
  // that can carry it (kf_backward_mfma.hip, kf_backward_steady.hip); NULL for every public call
  double* g_yy = nullptr;
  // internal: the LDS adjoint exits at once unless *run_if != 0 (the dense redo behind a whitened call, like KfFwdArgs::run_if)
  const int32_t* run_if = nullptr;
  // internal (steady-state adjoint, kf_backward_steady.hip): per-draw number of steps to sweep instead of T (NULL: T for every draw)
  const int32_t* steps = nullptr;
  // internal: workgroup i processes draw list[i] (NULL: draw i) -- a launch whose draws need very different step counts is ordered
  // so that the wavefronts sharing a SIMD carry about the same total (kf_vg_order_kernel, kf_api.hip)
  const int32_t* list = nullptr;
  // internal (kf_bwd_steady_scan_kernel): when fwd_logp is set the kernel first runs the FORWARD mean recursion of the steps behind ts[b]
  // (a_ts is in fwd_pred_a's slot ts; a_t of the later steps is written there, pred_a must alias it), adds their log-likelihood to
  // fwd_logp[b] and marks fwd_info[b]
  double* fwd_pred_a = nullptr;
  double* fwd_logp = nullptr;
  int32_t* fwd_info = nullptr;
};

struct KfSmoothArgs {
  long B;
  int T, m, r;
  unsigned flags;  // KF_FLAG_TV_T / _R / _Q: that matrix is [B][T][.][.], step t uses slice t (LDS kernel only)
  const double* Tm;
  const double* R;
  const double* Q;
  const double* filt_a;
  const double* filt_P;
  double* smooth_a;
  double* smooth_P;
  const int32_t* only_flagged;  // LDS kernel: when set, only draws with a non-zero flag are processed
  int exact;                    // KF_FLAG_EXACT_RECURSION: every step in full (kf_smoother_b4.hip reuses a converged filter's gain otherwise)
};

// kf_smoother_from_filter (kf_smoother_bf.hip): the smoothed moments from the filter's predicted tapes, no inverse
struct KfSmoothBfArgs {
  long B;
  int T, m;
  int filter;      // KF_FILTER_SINGLE flips the sign of d (kalman_filter.py:335)
  unsigned flags;  // KF_FLAG_Y_PER_DRAW
  int exact;       // KF_FLAG_EXACT_RECURSION: every step in full (no steady phase)
  const double* y;
  const double* Tm;
  const double* Z;
  const double* H;
  const double* d;       // nullable
  const double* pred_a;  // [B][T+1][m]
  const double* pred_P;  // [B][T+1][m][m]
  double* smooth_a;
  double* smooth_P;
};
bool kf_smoother_bf_ok(int m, int p, unsigned flags);
int kf_launch_smoother_bf(kf_handle* h, const KfSmoothBfArgs& a, hipStream_t stream);
// one lane per draw for m <= 4 (kf_smoother_bf_lane.hip)
bool kf_smoother_bf_lane_ok(int m, int p, unsigned flags);
int kf_launch_smoother_bf_lane(kf_handle* h, const KfSmoothBfArgs& a, hipStream_t stream);
// 16 <= m <= 79 on 16 x 16 fp64 MFMA tiles, one workgroup per draw (kf_smoother_bf_wg.hip)
bool kf_smoother_bf_wg_ok(int m, int p, unsigned flags);
int kf_launch_smoother_bf_wg(kf_handle* h, const KfSmoothBfArgs& a, hipStream_t stream);
// the same recursion on the HBM workspace for 15 < m <= 1024 (kf_smoother_bf_big.hip)
bool kf_smoother_bf_big_ok(int m, int p, unsigned flags);
size_t kf_smoother_bf_big_workspace_bytes(long B, int m);
int kf_launch_smoother_bf_big(kf_handle* h, const KfSmoothBfArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream);

// per-draw LDS footprints (in doubles) -- shared by host launchers and kernels
__host__ __device__ inline int kf_fwd_region_doubles(int m, int p) {
  int mm = m * m, pm = p * m, pp = p * p;
  int n = 5 * mm      // Pa, Pb, W, T, RQR
          + 4 * pm    // Z, M, K, KF
          + 3 * pp    // H, F, L
          + 3 * m     // a, af, c
          + 5 * p + 4;  // d, v, ym, obs, w (+3 scalars of the big-p path)
  return n | 1;       // odd stride: neighbouring draws start on different LDS banks
}
__host__ __device__ inline int kf_pair_table_doubles(int m) {
  int npair = m * (m + 1) / 2;
  return (npair + 1) / 2 + 1;
}

// one object per KF_PMAX (kf_forward_lds.hip compiled 5x); kf_api.hip dispatches on p
int kf_launch_forward_lds_p0(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lds_p1(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lds_p2(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lds_p4(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lds_p8(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lds(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
// register-resident row-owner kernels, one object per compile-time k_states (kf_forward_row.hip)
#define KF_ROW_M_MAX 10
#define KF_DECL_ROW(M) int kf_launch_forward_row_m##M(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
KF_DECL_ROW(1) KF_DECL_ROW(2) KF_DECL_ROW(3) KF_DECL_ROW(4) KF_DECL_ROW(5)
KF_DECL_ROW(6) KF_DECL_ROW(7) KF_DECL_ROW(8) KF_DECL_ROW(9) KF_DECL_ROW(10)
// row-split kernels (kf_forward_rowsplit.hip), one object per (even k_states, lanes per row)
int kf_launch_forward_rowsplit_m10_h2(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_rowsplit_m10_h3(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_rowsplit_m10_h5(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_rowsplit_m8_h2(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_rowsplit_m8_h4(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_rowsplit_m6_h3(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_rowsplit_m4_h2(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
// multivariate row-split kernels (kf_forward_rowsplit_mp.hip): runtime p <= 8, m <= padded even M
#define KF_MP_M_MAX 16
#define KF_MP_P_MAX 8
#define KF_DECL_MP(M, H) \
  int kf_launch_forward_rowsplit_mp_m##M##_h##H(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
KF_DECL_MP(16, 2) KF_DECL_MP(16, 4) KF_DECL_MP(14, 2) KF_DECL_MP(14, 4) KF_DECL_MP(12, 2) KF_DECL_MP(12, 3)
KF_DECL_MP(10, 2) KF_DECL_MP(10, 3) KF_DECL_MP(8, 2) KF_DECL_MP(8, 4) KF_DECL_MP(6, 1) KF_DECL_MP(6, 3)
KF_DECL_MP(4, 1) KF_DECL_MP(4, 2)
// accumulator-layout kernel (kf_forward_mfma.hip): one wavefront per draw, m <= 16, p <= 8
int kf_launch_forward_mfma(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
// 4x4x4-block kernel (kf_forward_b4.hip): v_mfma_f64_4x4x4_4b_f64, p == 1, m <= 14, mean carried in the tile
bool kf_forward_b4_ok(int m, int p, int r);
int kf_launch_forward_b4(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream);
// time-varying T / Z / H / c / d on the block tiles (kf_forward_b4.hip, kf_fwd_b4tv_kernel)
bool kf_forward_b4tv_ok(int m, int p, int r, unsigned flags);
int kf_launch_forward_b4tv(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
// time-chunked pipeline for small batches (kf_forward_b4c.hip, round 6): chunk-start covariances by doubling + the mean pass
struct B4cWs {
  double* start;   // [B][nc][m][m] predicted covariance chunk j starts from (slot 0 unused)
  double* psi;     // [B][nc][m][m] Psi_k: a_f((k + 1) s) = Psi_k a_f(k s) + (zero-start response), k >= 1
  double* K;       // [B][T][m]
  double* fv;      // [B][T][2]
  double* zend;    // [B][nc][m]
  int32_t* flag;   // [16]
};
bool kf_forward_b4c_shape_ok(const kf_handle* h, long B, int T, int p, int m, int r, unsigned flags);
void kf_forward_b4c_plan(const kf_handle* h, long B, int T, int* s, int* nc);
size_t kf_forward_b4c_carve(void* base, long B, int T, int m, int nc, B4cWs* W);
int kf_launch_b4c_starts(kf_handle* h, const KfFwdArgs& a, const B4cWs& W, int s, int nc, hipStream_t stream);
int kf_launch_b4c_means(kf_handle* h, const KfFwdArgs& a, const B4cWs& W, int s, int nc, hipStream_t stream);
// recursion + tape-writer wavefront per draw (kf_forward_b4p.hip)
bool kf_forward_b4p_ok(int m, int p, int r);
int kf_launch_forward_b4p(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
// several scalar observations per step on the same 4x4x4 block tiles (kf_forward_b4q.hip)
bool kf_forward_b4q_ok(int m, int p, int r);
int kf_launch_forward_b4q(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t kf_b4q_split_workspace_bytes(long B);
// block-tile variant (kf_forward_mfma_nb.hip): 16 < m <= 48 as 2 x 2 / 3 x 3 blocks of 16 x 16
#define KF_NB_M_MAX 48
bool kf_forward_mfma_nb_ok(int m, int p, int r);
int kf_launch_forward_mfma_nb(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
// workgroup-per-draw variant (kf_forward_mfma_wg.hip): 48 < m <= 64, four wavefronts per draw
#define KF_WG_M_MAX 64
bool kf_forward_mfma_wg_ok(int m, int p, int r);
int kf_launch_forward_mfma_wg(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
struct KfBwdArgs;
bool kf_backward_mfma_wg_ok(int m, int p, int r);
int kf_launch_backward_mfma_wg(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
bool kf_backward_mfma_nb_ok(int m, int p, int r);
int kf_launch_backward_mfma_nb(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
// block-MFMA smoother, Newton-Schulz inverses (kf_smoother_b4.hip): 5 <= m <= 15
bool kf_smoother_b4_ok(int m, int r);
int kf_launch_smoother_b4(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream);
// adjoint of the augmented recursion on 4x4x4 block MFMAs (kf_backward_b4.hip): p == 1, m + 2 <= 16
bool kf_backward_b4_ok(int m, int p, int r);
size_t kf_backward_b4_workspace_bytes(long B, int m);
int kf_launch_backward_b4(kf_handle* h, const KfBwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream);
int kf_launch_backward_mfma(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
// steady-state adjoint (kf_backward_steady.hip): the steps a forward kernel ran in its steady-state phase
bool kf_backward_steady_ok(int m, int p, int r);
bool kf_backward_steady_scan_ok(int m, int p);   // the time-parallel kernel takes the shape (and KF_VG_STEADY_SCAN=0 is not set)
size_t kf_backward_steady_workspace_bytes(long B, int m);
int kf_launch_backward_steady(kf_handle* h, const KfBwdArgs& a, const int32_t* ts, double* out_ab, double* out_Pb, double* out_P,
                              hipStream_t stream, const double* P_conv = nullptr);
// lane-per-draw kernels for tiny m (kf_forward_lane.hip)
#define KF_LANE_M_MAX 4
int kf_launch_forward_lane_m1(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lane_m2(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lane_m3(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward_lane_m4(kf_handle* h, const KfFwdArgs& a, hipStream_t stream);
int kf_launch_forward(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes,
                      hipStream_t stream);
// large-m path (kf_forward_big.hip): covariance in a caller-provided HBM workspace, MFMA GEMMs
// kf_forward_wgl.hip: 64 < m <= 96, p == 1, static matrices, no covariance tapes -- P resident in LDS, one workgroup per draw
bool kf_forward_wgl_ok(int m, int p, unsigned flags, bool cov_tapes);
size_t kf_forward_wgl_workspace_bytes(long B, int m);
int kf_launch_forward_wgl(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t kf_forward_big_workspace_bytes(long B, int m, int p);
// kf_forward_big.hip's 128 x 64-tile fp64-MFMA product for m x m operands (see there)
int kf_big_gemm_launch(hipStream_t st, bool trans_b, bool add_e, bool upper_mirror, const double* A, long sA, const double* Bm,
                       long sB, const double* E, long sE, double* C, long sC, int m, long draws);
int kf_big_gemm_launch_x(hipStream_t st, bool trans_b, bool add_e, const double* A, long sA, const double* Bm, long sB, const double* E,
                         long sE, double* C, long sC, int m, long draws, const uint32_t* tmask, int tmask_on_b, int neg_acc,
                         const double* u1, const double* v1, const double* u2, const double* v2, bool upper_mirror);
int kf_big_tmask_launch(hipStream_t st, const double* T, int m, long B, uint32_t* tmask);
// CSR copy of a batch of per-draw m x m matrices (kf_forward_big.hip, round 5): at most cap = 4 m entries per draw; rows with more than 32
// non-zeros are kept out of it as "dense rows" (drow, at most 4 per draw).  ok[0] stays non-zero while every draw fits, ok[1] = the
// largest number of dense rows of a draw (both on the device: read them back before trusting the arrays).
struct KfBspCsr {
  int* rowptr;   // [B][m + 1]
  int* col;      // [B][cap]
  double* val;   // [B][cap]
  int* drow;     // [B][4]
  int* nd;       // [B]
  int* ok;       // [2]
  int cap;
};
size_t kf_bsp_csr_bytes(long B, int m);
int kf_bsp_csr_build(hipStream_t st, const double* Tm, int m, long B, void* mem, KfBspCsr* c);   // carves `mem`, enqueues the build
// out = A X A' + u1 v1' + u2 v2' per draw: A = the CSR matrices (NO dense rows: ok[1] == 0), X dense [B][m][m], out another buffer
// (wout, or NULL: the product A X itself, [B][m][m])
int kf_bsp_sandwich_launch(hipStream_t st, const KfBspCsr& c, const double* X, double* out, int m, long B, const double* u1,
                           const double* v1, const double* u2, const double* v2, double* wout = nullptr);
// block map of T by columns (word per 64-column block, bit per 32-row chunk): for A T with T the untransposed B operand, tmask_on_b = 1
int kf_big_tmask_cols_launch(hipStream_t st, const double* T, int m, long B, uint32_t* tmask);
int kf_launch_forward_big(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes,
                          hipStream_t stream);
// dense H with p > 1 on the scalar-observation kernels by whitening (kf_forward_whiten.hip)
size_t kf_forward_whiten_workspace_bytes(long B, int T, int p, int m);
// the gradient of a dense-H model through the same whitening (round 5): whitened inputs + L^-1 + the whitened gradients' scratch
struct KfWhitenGrad {
  double *ys, *Zs, *ds, *Hs, *logdet, *Li, *gZs, *gds, *gHs, *gyy;   // Li [B][p][p]; gZs [B][p][m]; gds [B][p]; gHs, gyy [B][p][p]
  int32_t *nobs, *flag;
};
size_t kf_whiten_grad_carve(void* base, long B, int T, int p, int m, KfWhitenGrad* W);
int kf_launch_whiten_for_grad(kf_handle* h, const KfFwdArgs& a, const KfWhitenGrad& W, hipStream_t stream);
int kf_launch_unwhiten_grad(kf_handle* h, long B, int p, int m, const KfWhitenGrad& W, const double* g_logp, double* logp,
                            double* g_Z, double* g_H, double* g_d, hipStream_t stream);
int kf_launch_forward_whitened(kf_handle* h, const KfFwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t stream,
                               int (*inner)(kf_handle*, const KfFwdArgs&, void*, size_t, hipStream_t));
inline bool kf_forward_lds_fits(int m, int p) {
  size_t lds = sizeof(double) * ((size_t)kf_pair_table_doubles(m) + (size_t)kf_fwd_region_doubles(m, p));
  return lds <= KF_LDS_BYTES_MAX;
}
int kf_launch_backward_lds_p0(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_lds_p1(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_lds_p2(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_lds_p4(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_lds_p8(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
// register-resident backward kernels (kf_backward_rowsplit.hip), p == 1, even m <= 10
int kf_launch_backward_rowsplit_m10_h1(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m10_h2(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m10_h3(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m10_h5(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m8_h1(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m8_h2(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m8_h4(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m6_h1(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m6_h3(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m4_h1(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m4_h2(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_rowsplit_m2_h1(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_mfma(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward_lds(kf_handle* h, const KfBwdArgs& a, hipStream_t stream);
int kf_launch_backward(kf_handle* h, const KfBwdArgs& a, void* workspace, size_t workspace_bytes,
                       hipStream_t stream);
int kf_launch_smoother_lds(kf_handle* h, const KfSmoothArgs& a, hipStream_t stream);
// lane-per-draw smoother with the pseudo-inverse in registers (kf_smoother_lane.hip): m <= 4, static matrices
bool kf_smoother_lane_ok(int m);
int kf_launch_smoother_lane(kf_handle* h, const KfSmoothArgs& a, hipStream_t stream);
// per-draw LDS footprint of kf_smooth_lds_kernel with all 64 lanes on one draw (kf_smoother_lds.hip)
inline bool kf_smoother_lds_fits(int m) {
  const int M2 = (m + 1) / 2;
  return sizeof(double) * (size_t)((9 * m * m + 6 * m + 4 * M2 + 2) | 1) <= KF_LDS_BYTES_MAX;
}
// sub-block fp64-MFMA GEMM and transpose of kf_smoother_big.hip, shared with kf_backward_big.hip:
// C (M x N, ldc) = epi(A (M x K, lda) * op(B)), epi 0: store, 1: E + acc, 2: E - acc; batched over B
void kf_sb_gemm(hipStream_t st, long B, bool trans_b, int epi, const double* A, long sA, int lda, const double* Bm,
                long sB, int ldb, const double* E, long sE, int lde, double* C, long sC, int ldc, int M, int N, int K);
void kf_sb_transpose(hipStream_t st, long B, const double* src, long sS, double* dst, long sD, int m);
// large-m adjoint (kf_backward_big.hip): scalar observations, HBM workspace
inline bool kf_backward_lds_fits(int m, int p) {
  const size_t lds = sizeof(double) * ((size_t)kf_pair_table_doubles(m) +
                                       (size_t)((9 * m * m + 10 * p * m + 10 * p * p + 6 * m + 6 * p + 2) | 1));
  return lds <= KF_LDS_BYTES_MAX;
}
size_t kf_backward_big_workspace_bytes(long B, int m, int p, int n, int r);
int kf_launch_backward_big(kf_handle* h, const KfBwdArgs& a, void* workspace, size_t workspace_bytes,
                           hipStream_t stream);
// register-tile smoother (kf_smoother_mfma.hip): m <= 16, needs 4 B + 64 bytes of workspace (flags)
size_t kf_smoother_mfma_workspace_bytes(long B);
// launch order of one-wavefront-per-draw kernels whose draws need very different times (kf_vg_order_kernel, kf_api.hip): list = the
// draws by descending key (0 .. n), dealt out in snake order over the SIMDs; B <= INT_MAX.  Placement only.
int kf_launch_order(kf_handle* h, const int32_t* keys, int n, long B, int32_t* list, hipStream_t stream);
// the smoother behind the filter's convergence point, time-parallel (kf_smoother_b4_steady.hip): 5 <= m <= 10, static matrices
bool kf_smoother_b4_split_ok(const KfSmoothArgs& a);
size_t kf_smoother_b4_split_workspace_bytes(long B, int m);
int kf_launch_smoother_b4_steady(kf_handle* h, const KfSmoothArgs& a, void* ws, int32_t** start, int32_t** list, hipStream_t stream);
int kf_launch_smoother_mfma(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                            hipStream_t stream);
// block-tile smoother (kf_smoother_mfma_nb.hip): 16 < m <= 47, same flag workspace
bool kf_smoother_mfma_nb_ok(int m, int r);
int kf_launch_smoother_mfma_nb(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                               hipStream_t stream);
// workgroup-per-draw smoother (kf_smoother_mfma_wg.hip): 47 < m <= 64, no workspace
bool kf_smoother_mfma_wg_ok(int m, int r);
int kf_launch_smoother_mfma_wg(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                               hipStream_t stream);
int kf_launch_smoother_pinv_wide(kf_handle* h, const KfSmoothArgs& a, const int32_t* flags, hipStream_t stream);
// large-m smoother (kf_smoother_big.hip): HBM workspace, blocked Cholesky, fp64-MFMA GEMMs
size_t kf_smoother_big_workspace_bytes(long B, int m);
int kf_launch_smoother_big(kf_handle* h, const KfSmoothArgs& a, void* workspace, size_t workspace_bytes,
                           hipStream_t stream);
int kf_pick_lanes(const kf_handle* h, int m, long B);
int kf_launch_lyapunov(kf_handle* h, long B, int m, int r, const double* T, const double* R,
                       const double* Q, const double* S, double* P, int32_t* info, hipStream_t stream);
int kf_launch_simulate(kf_handle* h, long B, int n, int k, int n_sims, const double* mu,
                       const double* cov, const double* z, double jitter, double* out, int32_t* info,
                       hipStream_t stream);
int kf_launch_dare(kf_handle* h, long B, int m, int p, int r, const double* T, const double* Z, const double* R,
                   const double* H, const double* Q, const double* Pinit, double* P, int32_t* info, int max_iter,
                   double tol, hipStream_t stream);
int kf_launch_dare_backward(kf_handle* h, long B, int m, int p, int r, const double* T, const double* Z, const double* R,
                            const double* H, const double* Q, const double* P, const double* g_P, double* g_T,
                            double* g_Z, double* g_R, double* g_H, double* g_Q, int accumulate, int32_t* info,
                            hipStream_t stream, int h_diag_only = 0);
int kf_launch_simulate_statespace(kf_handle* h, long B, int n_sims, int n, int p, int m, int r, const double* T,
                                  const double* Z, const double* R, const double* H, const double* Q,
                                  const double* x0, const double* zs, const double* zo, double* states, double* obs,
                                  int32_t* info, hipStream_t stream);
int kf_launch_assemble(kf_handle* h, long B, int n_theta, const double* theta, long total,
                       const double* tmpl, int n_map, const int32_t* dst, const int32_t* src,
                       double* out, hipStream_t stream);

// Every extern "C" entry point runs on the handle's device whatever device is current in the calling
// thread (launches and hipFuncSetAttribute act on the CURRENT device; a NULL stream is the current
// device's default stream), and restores the caller's device on return.
struct KfDeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  const kf_handle* prev_tls = nullptr;
  explicit KfDeviceGuard(const kf_handle* h) {
    prev_tls = kf_tls_handle;
    kf_tls_handle = h;
    if (!h) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != h->device) {
      err = hipSetDevice(h->device);
      switched = err == hipSuccess;
    }
  }
  ~KfDeviceGuard() {
    kf_tls_handle = prev_tls;
    if (switched) (void)hipSetDevice(prev);
  }
  KfDeviceGuard(const KfDeviceGuard&) = delete;
  KfDeviceGuard& operator=(const KfDeviceGuard&) = delete;
};
// (entry points that only ANSWER questions about dispatch -- kf_forward_kernel_name, kf_workspace_bytes_for -- read the same switches
// without touching the device: they publish their handle's options the same way)
struct KfTlsHandle {
  const kf_handle* prev;
  explicit KfTlsHandle(const kf_handle* h) : prev(kf_tls_handle) { kf_tls_handle = h; }
  ~KfTlsHandle() { kf_tls_handle = prev; }
  KfTlsHandle(const KfTlsHandle&) = delete;
  KfTlsHandle& operator=(const KfTlsHandle&) = delete;
};
#define KF_ON_DEVICE(h)                                                                        \
  KfDeviceGuard kf_dev_guard_(h);                                                              \
  if (kf_dev_guard_.err != hipSuccess) {                                                       \
    KF_SET_ERR(h, "cannot make device %d current: %s", (h)->device, hipGetErrorString(kf_dev_guard_.err)); \
    return KF_ERR_HIP;                                                                         \
  }

#define KF_SET_ERR(h, ...)                                 \
  do {                                                     \
    if (h) snprintf((h)->err, sizeof((h)->err), __VA_ARGS__); \
  } while (0)

#define KF_HIP_CHECK(h, expr)                                                     \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      KF_SET_ERR(h, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return KF_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)
