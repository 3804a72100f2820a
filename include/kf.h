/* kf.h -- C-ABI of the MI355X-native batched Kalman filter / RTS smoother.
 *
 * This is the drop-in boundary for the one hot path of
 * jessegrabowski/pymc_statespace: everything `pymc_statespace.filters` computes
 * numerically.  The reference has NO native code and NO FFI for this path (it is a
 * symbolic PyTensor scan), so each entry point cites the reference Python
 * interface it replaces; the binding a maintainer adds on the reference side is
 * the ctypes stub shown in INTEGRATION.md.
 *
 *   kf_filter_forward   <- BaseFilter.build_graph + kalman_step + update variants +
 *                          _postprocess_scan_results
 *                          (pymc_statespace/filters/kalman_filter.py:126-163,195-252,
 *                           255-351,444-505,165-193), evaluated for B draws at once
 *   kf_filter_backward  <- pytensor.grad through that scan (implicit in the reference:
 *                          README.md:6; core/statespace.py:174 registers the Potential
 *                          NUTS differentiates)
 *   kf_smoother         <- KalmanSmoother.build_graph
 *                          (pymc_statespace/filters/kalman_smoother.py:56-104)
 *
 * Conventions
 *   - Plain C, no torch / HIP types in signatures.  `stream` is a hipStream_t passed
 *     as void* (NULL = the null stream).  All calls are asynchronous on `stream` (one exception, a
 *     4- / 8-byte read-back on the m > 64 workspace paths of the filter, its gradient and kf_smoother_from_filter: kf_last_forward_kernel below).
 *   - Every data pointer is a DEVICE pointer to fp64 (the reference computes in
 *     float64: kalman_filter.py:16).  The caller owns all buffers; the library
 *     never allocates in a call, never frees and never keeps a pointer.
 *   - Layouts are the reference's per-draw row-major layouts with ONE extra leading
 *     axis B (the batch of parameter draws), draw-major:
 *         y   [T][p]            shared by all draws   (or [B][T][p], KF_FLAG_Y_PER_DRAW)
 *         a0  [B][m]     P0 [B][m][m]   T [B][m][m]   Z [B][p][m]
 *         R   [B][m][r]  H  [B][p][p]   Q [B][r][r]   c [B][m]   d [B][p]
 *     outputs (each tape pointer may be NULL = do not store):
 *         logp   [B]            ll_obs [B][T]
 *         filt_a [B][T][m]      pred_a [B][T+1][m]     (pred_a[:,0] = a0)
 *         filt_P [B][T][m][m]   pred_P [B][T+1][m][m]  (pred_P[:,0] = P0)
 *         info   [B]  int32     0 = ok; else ((t+1) << 8) | k : at time step t the
 *                               leading minor k of F was not positive definite
 *   - NaN in y marks a missing observation (kalman_filter.py:195-213).
 *   - Return value: 0 ok; < 0 argument / HIP error (text via kf_last_error).
 *     Numerical failures are reported per draw through `info`, like LAPACK.
 *   - Threading: re-entrant per handle; one handle per (host thread, device).
 *   - Devices: every entry point makes the handle's device current for the duration of the
 *     call and restores the caller's current device on return; buffers and `stream` must
 *     belong to the handle's device (NULL = that device's null stream).
 */
#ifndef KF_H_
#define KF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#define KF_API __attribute__((visibility("default")))
#else
#define KF_API
#endif

#define KF_ABI_VERSION 7 /* 7: kf_filter_backward_tapes, kf_set_option / kf_get_option.  6: kf_smoother_from_filter. \
                          * 5: KF_FLAG_STEADY_STATE; the block kernel's steady-state phase off by default below 1536 steps; kf_workspace_bytes_for(KF_WS_FORWARD) covers the split pipeline. \
                          * 2: KF_FLAG_H_DIAGONAL, KF_KERNEL_MFMA, kf_dare, kf_simulate_statespace, workspace use by smoother/backward; \
                            3: kf_workspace_bytes_for, kf_forward_kernel_name_ws; every entry point runs on the handle's device; \
                            4: kf_dare_backward, time-varying matrices in kf_filter_backward, kf_filter_value_and_grad */

typedef struct kf_handle kf_handle;

/* filter variants: FILTER_FACTORY keys of pymc_statespace/core/statespace.py:25-31 */
enum {
  KF_FILTER_STANDARD = 0,   /* StandardFilter         kalman_filter.py:255-284 */
  KF_FILTER_CHOLESKY = 1,   /* CholeskyFilter         kalman_filter.py:287-318 */
  KF_FILTER_UNIVARIATE = 2, /* UnivariateFilter       kalman_filter.py:444-505 */
  KF_FILTER_SINGLE = 3      /* SingleTimeseriesFilter kalman_filter.py:321-351 (p == 1) */
};

/* flags */
enum {
  /* log(2 pi) is added n_observed times per step (textbook / statsmodels).  When
   * clear, the reference's own counts are used: Standard/Single once per step
   * (kalman_filter.py:281,:349), Cholesky p times (:314), Univariate once per
   * non-degenerate scalar update (:503).  Identical for p == 1. */
  KF_FLAG_MVN_CONST_EXACT = 1u << 0,
  /* y is [B][T][p] instead of the shared [T][p] */
  KF_FLAG_Y_PER_DRAW = 1u << 1,
  /* Time-varying system matrices: the flagged matrix is [B][T][rows][cols] (per draw: time on
   * axis 0, the reference's 3-D convention, filters/utilities.py:1-20) and step t uses slice t
   * for both its update and its predict (kalman_filter.py:231-252).  kf_filter_forward (all seven)
   * and kf_smoother (T, R, Q: the step that smooths t predicts from t with slice t); LDS-resident
   * kernels only (m <~ 47).  kf_filter_backward: all seven as well; the gradient of a flagged matrix is a tape of
   * the same [B][T][rows][cols] shape. */
  KF_FLAG_TV_T = 1u << 4,
  KF_FLAG_TV_Z = 1u << 5,
  KF_FLAG_TV_R = 1u << 6,
  KF_FLAG_TV_H = 1u << 7,
  KF_FLAG_TV_Q = 1u << 8,
  KF_FLAG_TV_C = 1u << 9,
  KF_FLAG_TV_D = 1u << 10,
  KF_FLAG_TV_ANY = 0x7f0u,
  /* The caller asserts that every draw's H is diagonal (e.g. VARMAX measurement error,
   * models/VARMAX.py:80-83).  Standard/CholeskyFilter with p > 1 and m <= 48 may then process the
   * p observations of a step one at a time in the register-resident kernels -- algebraically
   * the same filter (results agree to rounding).  Off-diagonal entries are ignored when set.
   * Round 4: WITHOUT the flag a dense H (Standard / Cholesky, 2 <= p <= 8, a shape one of those kernels takes) is whitened per draw --
   * H = L L', y* = L^-1 y, Z* = L^-1 Z, d* = L^-1 d, log det L subtracted per observed step: the same state moments and
   * log-likelihood to rounding (kf_forward_whiten.hip) -- given the workspace kf_workspace_bytes_for asks for (B T p doubles for the
   * per-draw series); a partially missing row or an H that is not positive definite makes the dense LDS kernel redo the batch.
   * The gradient of a dense-H model: kf_filter_value_and_grad whitens too (round 5, m <= 16: see there); kf_filter_backward on the caller's
   * own tapes still runs the LDS adjoint (entrywise g_H). */
  KF_FLAG_H_DIAGONAL = 1u << 11,
  /* Forward filter: never switch to the steady-state phase.  The kernels for static models -- the lane kernel (p == 1, m <= 4),
   * the block kernel (p == 1, m <= 10) and the multi-observation block kernel (3 <= p <= 5, m >= 8) -- can test every 8 (lane
   * kernel: 16) steps whether the predicted covariance has stopped moving and from then on advance only the mean with the
   * converged gain, writing the two constant covariances to the tapes -- what statsmodels' filter does by default
   * (`tolerance`).  The test (round 3): no missing observation ahead, series of >= 64 steps, and on TWO consecutive checks the
   * one-step change of every entry is <= 1e-15 (lane kernel 3e-16) of the largest entry AND <= 1e-12 of the entry's own scale
   * sqrt(P_ii P_jj) -- the second condition keeps a large (diffuse) block from hiding a small block that still moves.
   * The reference recomputes every step; measured differences on the BASELINE workloads: <= 7e-14 relative on logp, <= 5e-13
   * absolute on the covariance tapes.
   * DEFAULT: the lane kernel and the multi-observation block kernel use the phase; the block kernel (p == 1, 4 < m <= 10:
   * the north-star shape) uses it only for series of >= 1536 steps (at T = 1000 the recursion at every step is the faster
   * kernel, so the default there IS the reference's recursion) -- then, given the workspace kf_workspace_bytes asks for, as a
   * split pipeline: recursion launches over compacted lists of the draws still converging, the converged draws' constant
   * covariance tapes and time-parallel mean recursion from separate kernels on a second stream (kf_forward_b4_steady.hip).
   * KF_STEADY_STATE=0 in the environment has the same effect as this flag, KF_STEADY_STATE=1 that of KF_FLAG_STEADY_STATE.
   * kf_smoother honours the flag too: by default the block smoother (5 <= m <= 15, kf_smoother_b4.hip) and the lane smoother (m <= 2)
   * reuse the last step's gain X = P_hat^-1 T P_f wherever the FILTERED covariance on the tape is bit for bit the next step's (the
   * constant tapes a steady-state forward writes) instead of inverting the same matrix again; with the flag the prediction covariance
   * is inverted at every step, as kalman_smoother.py:87-97 does.  KF_SMOOTHER_STEADY=0 in the environment has the same effect.
   * Series of >= 1536 steps, 5 <= m <= 10 (round 4, kf_smoother_b4_steady.hip): the steps from the last one down to the first whose
   * filtered covariance is the last step's within 1e-14 of its largest entry -- every covariance of that range is read and compared --
   * are smoothed with one gain: the mean by a prefix scan over time, the covariance recursed to its fixpoint and written as a
   * constant; <= 6e-13 of a draw's largest entry away from every step in full on the BASELINE workloads.  Not with this flag
   * (KF_SMOOTHER_SPLIT=0 / 1 in the environment: never / for any series length).
   * Round 6: the flag also keeps a SMALL batch on the sequential kernel -- without it kf_filter_forward runs p == 1, m <= 10, static models
   * of at most 3/4 wavefront per SIMD (768 draws on an MI355X; a full round, 1024, from 1536 steps on) and >= 256 steps through the
   * time-chunked pipeline described at kf_filter_forward: the same recursion at every step, but each chunk of the series starts from a
   * covariance computed by doubling instead of from its predecessor's last step (<= 1e-13 relative apart, not bit for bit). */
  KF_FLAG_EXACT_RECURSION = 1u << 12,
  /* Forward filter: use the steady-state phase wherever the selected kernel has one, whatever the series length (the block
   * kernel's default depends on it, see above).  KF_FLAG_EXACT_RECURSION wins when both are set. */
  KF_FLAG_STEADY_STATE = 1u << 13,
  /* kf_smoother with time-varying T / R / Q: the step that smooths t uses slice t + 1 -- the pairing of the reference's own scan
   * (kalman_smoother.py:62-77 reverses the n-long matrix sequences next to the (n-1)-long filtered tapes).  Default (flag clear):
   * slice t, the pairing of the forward filter that produced the tapes. */
  KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING = 1u << 14
};

/* error codes */
enum {
  KF_OK = 0,
  KF_ERR_ARG = -1,       /* bad dimension / NULL pointer / unsupported combination */
  KF_ERR_HIP = -2,       /* a HIP runtime call failed */
  KF_ERR_WORKSPACE = -3, /* workspace too small */
  KF_ERR_UNSUPPORTED = -4
};

typedef struct kf_dims {
  int64_t B;      /* draws */
  int32_t T;      /* time steps (n_obs) */
  int32_t p;      /* k_endog */
  int32_t m;      /* k_states */
  int32_t r;      /* k_posdef */
  int32_t filter; /* KF_FILTER_* */
  uint32_t flags; /* KF_FLAG_* */
} kf_dims;

KF_API int kf_abi_version(void);

/* One handle per (process, device): holds the device id, kernel tuning and the last
 * error text.  `device` is the HIP ordinal. */
KF_API int kf_create(kf_handle** out, int device);
KF_API void kf_destroy(kf_handle* h);
KF_API const char* kf_last_error(const kf_handle* h);

/* Path switches per HANDLE (round 6).  Every documented KF_* selector between complete, parity-tested paths (INTEGRATION.md,
 * "Environment switches": KF_STEADY_STATE, KF_FORWARD_CHUNKED, KF_SMOOTHER_BF_WG, ...) can be set on a handle; the value then wins
 * over the process environment for every call made through THAT handle (value "" = behave as if the variable were unset, whatever
 * the environment says; value NULL = remove the override, the environment counts again).  Names the library does not read are
 * refused (KF_ERR_ARG), values longer than 23 characters too.  kf_get_option: the value a call through this handle would see
 * (handle first, then the environment; NULL = unset).  A handle is used by one thread at a time, as before. */
KF_API int kf_set_option(kf_handle* h, const char* name, const char* value);
KF_API const char* kf_get_option(const kf_handle* h, const char* name);

/* Tuning knob: lanes of a wavefront that cooperate on one draw (0 = choose from dims).
 * LDS kernels take 4, 8, 16, 32 or 64; the row-split forward kernels take m*H (H lanes per
 * row of P) for the compiled H; other values fall back to the automatic choice. */
KF_API int kf_set_lanes_per_draw(kf_handle* h, int lanes);

/* Forward-kernel selection.  AUTO (static matrices): p == 1, m <= 14 -> 4x4x4-block MFMA kernel with
 * the state vector carried in the covariance tile, one wavefront per draw ("b4": always for m >= 9;
 * for m = 7, 8 and m = 4..6 while the batch cannot fill the chip with packed wavefronts, for m = 3
 * up to 2048 draws), else one lane per draw, 4 / 16 / 64 draws per wavefront by batch size ("lane", m <= 4: always for
 * m <= 2) / register-resident row-owner kernels
 * ("row", m <= 9); 2 <= p <= 8 scalar observations (UnivariateFilter / KF_FLAG_H_DIAGONAL), m <= 15 -> the same block
 * tiles with the observations taken four at a time, a 4 x 4 LDL' on the wavefront ("b4" as well: always for m >= 9, below
 * that up to 2048 draws); m <= 16, p <= 8 with p == 1 / UnivariateFilter / KF_FLAG_H_DIAGONAL otherwise -> covariance
 * in the 16 x 16 MFMA accumulator layout, one wavefront per draw ("mfma": p == 1 with m = 15, 16; p > 1 with m = 16)
 * or the multivariate row kernel ("row": p > 1, m <= 8, large batches); 16 < m <= 48 with the same eligibility -> covariance as 2 x 2 /
 * 3 x 3 blocks of 16 x 16 in accumulator layout ("mfma" as well); 48 < m <= 64 -> one workgroup of
 * four wavefronts per draw, one column block each ("mfma"); 64 < m <= 96 with p == 1 (Standard / Cholesky /
 * SingleTimeseries) and no covariance tape requested -> one workgroup per draw with the covariance resident in LDS for the
 * whole series (needs the workspace kf_workspace_bytes_for asks for); other tiles that fit LDS ->
 * generic LDS-resident kernel ("lds", also all time-varying models); else the workspace + MFMA
 * GEMM path.  The gradient and the smoother follow the same ranges (block-tile kernels for
 * 16 < m <= 48 / 47, workgroup kernels up to m = 64).  Results agree to
 * rounding. */
enum { KF_KERNEL_AUTO = 0, KF_KERNEL_LDS = 1, KF_KERNEL_ROW = 2, KF_KERNEL_LANE = 3, KF_KERNEL_MFMA = 4, KF_KERNEL_B4 = 5 };
KF_API int kf_set_forward_kernel(kf_handle* h, int kernel);
/* Name of the kernel kf_filter_forward would launch for `dims` (for profiles / bench logs), assuming
 * the caller passes the workspace kf_workspace_bytes asks for; the _ws form takes the size of the
 * workspace the caller will actually pass (0 = none).  The name is that of the call WITH covariance tapes: for 64 < m <= 96,
 * p == 1 a call without them (filt_P == pred_P == NULL) runs kf_fwd_wgl_kernel (covariance resident in LDS) instead of the
 * workspace path this reports; for p = 1, m <= 10 a batch of <= 1024 draws with covariance tapes runs the recursion + writer pair
 * kernel (kf_fwd_b4p_kernel). */
KF_API const char* kf_forward_kernel_name(const kf_handle* h, const kf_dims* dims);
KF_API const char* kf_forward_kernel_name_ws(const kf_handle* h, const kf_dims* dims, size_t workspace_bytes);
/* On the workspace path (m > 64) the kernel depends on the DATA: with p == 1, m <= 1024, no filtered covariance tape and a T of at
 * most 4 m non-zeros per draw outside at most 4 rows of more than 32 (the structural models: seasonal dummies, trigonometric
 * terms, companion forms), T P_f T' + R Q R' is evaluated as a gather over a CSR copy of T ("bsp_step_kernel": one pass over P per
 * step, HBM bound) instead of two fp64 MFMA products ("big_gemm_kernel").  Results agree to rounding.  The verdict is found on the
 * device: such a call reads 4 bytes back and SYNCHRONISES `stream` once, before the recursion is enqueued -- the one exception to
 * "asynchronous on stream" above.  While `stream` is being CAPTURED into a graph nothing can be read back: the call then takes the
 * path the handle's last un-captured call on the same (T pointer, B, m) took (round 6: the warm-up call every capture needs decides) --
 * the CSR gather under a device-side guard: the replay rebuilds the CSR from the T it finds and, should that T no longer fit, every
 * draw comes back with logp = NaN and info = 0x7fffff02 (re-capture), never with a wrong number; a capture without such a call before
 * it, or KF_BIG_SPARSE=0 in the environment: the MFMA path.  kf_last_forward_kernel: the kernel that carried the recursion of this
 * handle's last kf_filter_forward on the workspace path ("" before the first).
 * The same holds for kf_filter_backward / kf_filter_value_and_grad with m >= 128 and for kf_smoother_from_filter with m > 15: they
 * compact T' and, when it is sparse without dense rows, evaluate T' X T (X the adjoint / the smoother's carried matrix) as one CSR
 * gather instead of two block-sparse products -- one 8-byte read-back + stream synchronisation per call, skipped under capture or
 * KF_BIG_SPARSE=0. */
KF_API const char* kf_last_forward_kernel(const kf_handle* h);

/* Scratch the three entry points need for `dims` (max over them, for the kernel selection in force
 * on `h`); may be 0.  Non-zero for k_states too large for the on-chip kernels (m > 64, or m > ~45
 * with a dense H: forward 4 B m^2 doubles; smoother 6 B m^2 doubles; gradient 7 B m^2 + 2 B m T
 * doubles) and, a few bytes per draw, for the register-tile smoother
 * smoothers
 * (5 <= m <= 64: per-draw flags; without them kf_smoother uses the slower LDS kernel, m <= 47, or
 * cannot redo singular draws with the pseudo-inverse, 47 < m <= 64). */
KF_API int kf_workspace_bytes(const kf_handle* h, const kf_dims* dims, size_t* bytes);
/* The same per entry point (`which`: OR of KF_WS_*), so that a forward-only or smoother-only caller
 * does not hold the gradient's scratch.  The HBM-workspace kernels (m > 64) put the draw axis in
 * gridDim.y/z: B <= 65535 there (KF_ERR_UNSUPPORTED beyond; chunk the batch). */
enum { KF_WS_FORWARD = 1, KF_WS_BACKWARD = 2, KF_WS_SMOOTHER = 4, KF_WS_ALL = 7,
       KF_WS_VALUE_AND_GRAD = 8 /* kf_filter_value_and_grad: tapes + both passes (not part of KF_WS_ALL) */,
       KF_WS_SMOOTHER_FROM_FILTER = 16 /* kf_smoother_from_filter: 0 for m <= 15; 16 <= m <= 79 run on chip and use none of it (not part of KF_WS_ALL) */ };
KF_API int kf_workspace_bytes_for(const kf_handle* h, const kf_dims* dims, int which, size_t* bytes);

/* Forward filter for B draws.  c, d may be NULL (= 0: initialize_intercepts,
 * kalman_filter.py:36-51).  ll_obs / filt_* / pred_* / info may be NULL.
 * Small batches (round 6, csrc/kf_forward_b4c.hip).  One GPU's share of a sharded job -- 512 draws of the north-star batch -- is one
 * wavefront per two SIMDs, and a launch then lasts as long as ONE wavefront's chain of T dependent steps (0.58 ms at T = 1000
 * whatever the batch).  For p == 1, m <= 10, static matrices and a series without missing observations the covariance sequence does not
 * depend on y, and the s-step map of the filtered covariance is one element (A, C, J) of Sarkka & Garcia-Fernandez' associative
 * operator raised to the s-th power: log2(s) squarings per draw.  Given the workspace kf_workspace_bytes_for(KF_WS_FORWARD) asks for
 * (B T (m + 2) doubles + a few m x m matrices per chunk) the call runs: (1) the doubling and the chunk-start covariances, (2) the
 * block kernel's recursion over B x nc chunks side by side (nc <= 8 chunks of s = 2^k steps; the unchanged step arithmetic), (3) the
 * mean recursion per chunk with the gains of (2), chunk starts chained through Psi_k = A_s (I + P_f J_s)^-1.  Every chunk checks the
 * covariance it ends with against the one its successor started from (1e-12 of the largest entry); a mismatch, a missing observation,
 * Z R Q R' Z' + H <= 0 or an F <= 0 raises a device flag and the sequential kernel, launched behind with that flag as its run
 * condition, redoes the batch: no host synchronisation, capturable into a hipGraph.  Measured (profiles/r06/): 512 x 1000, m = 10, all
 * six outputs 0.58 -> 0.40 ms, 256 draws 0.54 -> 0.31; outputs within 1e-13 of the sequential kernel's (|dlogp| <= 3e-12), asserted at
 * 1e-11 / 1e-8.  KF_FLAG_EXACT_RECURSION or KF_FORWARD_CHUNKED=0 in the environment: the sequential kernel. */
KF_API int kf_filter_forward(kf_handle* h, const kf_dims* dims,
                      const double* y, const double* a0, const double* P0,
                      const double* T, const double* Z, const double* R,
                      const double* H, const double* Q,
                      const double* c, const double* d,
                      double* logp, double* ll_obs,
                      double* filt_a, double* pred_a, double* filt_P, double* pred_P,
                      int32_t* info, void* workspace, size_t workspace_bytes, void* stream);

/* Log-likelihood per draw AND the gradient of  sum_b g_logp[b]*logp[b]  (g_logp NULL = all ones) with respect to every
 * input, in one call -- what a NUTS leapfrog step evaluates (the reference: pytensor.grad through the filter's scan,
 * core/statespace.py:174).  The forward filter keeps its predicted tapes inside the workspace and the adjoint sweep
 * consumes them there: nothing but logp, info and the nine gradient arrays is written for the caller.  Gradient
 * conventions as kf_filter_backward.  workspace: kf_workspace_bytes_for(h, dims, KF_WS_VALUE_AND_GRAD, &bytes) --
 * B (T + 1)(m + m^2) doubles of tape plus what the two passes need themselves.
 * Steady-state adjoint (static models, scalar observations): where the forward kernel has a steady-state phase the steps behind
 * each draw's convergence point are differentiated as what they are -- a linear mean recursion with a constant gain plus ONE
 * Riccati adjoint per draw (kf_backward_steady.hip, kf_dare_backward) -- and only the steps before it by the adjoint of the full
 * recursion; agrees with the adjoint at every step to <= 1e-12 of a gradient's largest entry on the BASELINE workloads (asserted
 * at 1e-10).  Used by default for the multi-observation block kernel (3 <= p <= 5, m >= 8: C4) and, round 4, for the p == 1 block
 * kernel (m <= 10) from two rounds of the chip's wavefront slots on (B >= 32 x CUs), or one round (B >= 16 x CUs) with T >= 768 --
 * below that a launch lasts as long as its slowest draw's chain and the phase gains nothing (measured: kf_api.hip).  There the
 * steps behind the convergence point are time-parallel in BOTH directions (kf_bwd_steady_scan_kernel: the forward means and the
 * adjoint as prefix scans, 64 steps per wavefront pass).  KF_FLAG_STEADY_STATE requests the steady-state adjoint for any batch,
 * KF_FLAG_EXACT_RECURSION (or KF_VG_STEADY=0 in the environment) the adjoint at every step.
 * Behind the lane kernel (p == 1, m <= 3) the same split applies up to 32 x CUs draws: the scan
 * kernel takes a_t and the converged covariance from the compact tapes, kf_bwd_b4_kernel the steps before the change-over.
 * Dense H with p > 1 (round 5; Standard / Cholesky, 2 <= p <= 8, m <= 16, no KF_FLAG_H_DIAGONAL): the call whitens the model per draw
 * (H = L L', y* = L^-1 y, Z* = L^-1 Z, d* = L^-1 d: kf_forward_whiten.hip), evaluates value and gradient of the whitened model on the
 * scalar-observation kernels above, and maps the gradients back -- g_Z = L^-T g_Z*, g_d = L^-T g_d*, and the FULL p x p
 * g_H = L^-T G* L^-1 with G* = -(g n I + sum_t vb_t y~*_t' + g_Z* Z*') / 2 from the invariance of the joint likelihood under
 * (y*, Z*, d*, H*) -> (A y*, A Z*, A d*, A H* A') (n: fully observed steps; vb_t: the adjoint of the innovation, accumulated by the
 * adjoint kernels).  C4 with a dense H: 415 -> 24 ms (14.6 with a diagonal one).  g_H is symmetric; g_P0 is the symmetric part of the
 * entrywise gradient, as with KF_FLAG_H_DIAGONAL (all a symmetric P0 can see).  A partially missing row or an H that is not positive
 * definite raises a device flag and the dense LDS kernels, launched behind with run_if = flag, redo the batch (no host
 * synchronisation).  KF_FORWARD_WHITEN=0 in the environment keeps the dense adjoint for everything.
 * Workspace reuse: on the block kernels' path the launch order of the forward kernel's workgroups follows the convergence points the PREVIOUS
 * call left in the same workspace (same B, T, m: a sampler's successive calls) -- draws that will run long are spread over the
 * SIMDs.  This decides only WHERE a draw runs: every output is bitwise the same with a fresh, a reused or an overwritten
 * workspace (KF_VG_SCHED_HINT=0 in the environment: natural order). */
KF_API int kf_filter_value_and_grad(kf_handle* h, const kf_dims* dims,
                             const double* y, const double* a0, const double* P0,
                             const double* T, const double* Z, const double* R,
                             const double* H, const double* Q,
                             const double* c, const double* d, const double* g_logp,
                             double* logp, int32_t* info,
                             double* g_a0, double* g_P0, double* g_T, double* g_Z,
                             double* g_R, double* g_H, double* g_Q, double* g_c, double* g_d,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Reverse-mode gradient of  sum_b g_logp[b]*logp[b] + sum_{b,t} g_ll_obs[b][t]*ll_obs[b][t]
 * with respect to every input, per draw.  Needs the predicted tapes of a forward
 * call on the same inputs (pred_a, pred_P).  g_logp or g_ll_obs may be NULL (= 0);
 * any output gradient pointer may be NULL (= not wanted).  Gradients treat every
 * matrix entry as an independent variable (what autodiff of the reference graph
 * returns).  With KF_FLAG_H_DIAGONAL and p > 1 (Standard/Cholesky on the scalar-observation
 * kernels) g_H carries the diagonal only (off-diagonal entries 0) and g_P0 the symmetric part
 * of the entrywise gradient -- all a diagonal-H / symmetric-P0 parameterisation can see.
 * Time-varying matrices (KF_FLAG_TV_*): the gradient of a flagged matrix has that matrix's shape, [B][T][rows][cols]
 * (slice t = the partial w.r.t. the slice step t used); unflagged ones stay per draw.  T / Z / H / c / d flagged (R, Q static),
 * m <= 16, scalar observations: the accumulator-layout adjoint with per-step slices (round 5); else the LDS-resident adjoint
 * (KF_ERR_UNSUPPORTED when m, p do not fit 160 KiB). */
KF_API int kf_filter_backward(kf_handle* h, const kf_dims* dims,
                       const double* y, const double* a0, const double* P0,
                       const double* T, const double* Z, const double* R,
                       const double* H, const double* Q,
                       const double* c, const double* d,
                       const double* pred_a, const double* pred_P,
                       const double* g_logp, const double* g_ll_obs,
                       double* g_a0, double* g_P0, double* g_T, double* g_Z,
                       double* g_R, double* g_H, double* g_Q, double* g_c, double* g_d,
                       void* workspace, size_t workspace_bytes, void* stream);

/* kf_filter_backward with cotangents of the filter's TAPE outputs as well (round 6; ABI 7): the gradient of
 *   sum_b g_logp[b] logp[b] + sum g_ll_obs . ll_obs + sum g_filt_a . filtered_states + sum g_pred_a . predicted_states
 *   + sum g_filt_P . filtered_covs + sum g_pred_P . predicted_covs
 * with respect to every input -- what autodiff of the reference's scan gives for a loss on any of the filter's outputs
 * (README.md:6; filters/kalman_filter.py:166-212 returns all six).  Shapes as the forward's outputs: g_filt_a [B][T][m],
 * g_pred_a [B][T+1][m], g_filt_P [B][T][m][m], g_pred_P [B][T+1][m][m]; any of the six cotangents may be NULL (= 0), not all.
 * Covariance cotangents need not be symmetric (the filter's covariances are: only the symmetric part acts); predicted slot 0 is
 * (a0, P0) as given, so its cotangent goes to g_a0 / g_P0 unchanged.  Everything else as kf_filter_backward.
 * Kernels: static matrices, m <= 16, scalar observations (p == 1, the Univariate filter, or KF_FLAG_H_DIAGONAL; then g_H / g_P0 as
 * kf_filter_backward documents for those): the accumulator-layout adjoint (csrc/kf_backward_mfma.hip, one wavefront per draw); else
 * the LDS-resident adjoint (csrc/kf_backward_lds.hip) -- every filter type, dense H, time-varying matrices; KF_ERR_UNSUPPORTED when
 * m, p do not fit 160 KiB of LDS (m > ~45).  With all four tape cotangents NULL the call IS kf_filter_backward (same dispatch). */
KF_API int kf_filter_backward_tapes(kf_handle* h, const kf_dims* dims,
                             const double* y, const double* a0, const double* P0,
                             const double* T, const double* Z, const double* R,
                             const double* H, const double* Q,
                             const double* c, const double* d,
                             const double* pred_a, const double* pred_P,
                             const double* g_logp, const double* g_ll_obs,
                             const double* g_filt_a, const double* g_pred_a,
                             const double* g_filt_P, const double* g_pred_P,
                             double* g_a0, double* g_P0, double* g_T, double* g_Z,
                             double* g_R, double* g_H, double* g_Q, double* g_c, double* g_d,
                             void* workspace, size_t workspace_bytes, void* stream);

/* RTS smoother over the filtered tapes: smooth_a [B][T][m], smooth_P [B][T][m][m].
 * Only dims->B, T, m, r and the KF_FLAG_TV_T / _R / _Q flags are used.  The gain uses the Moore-Penrose
 * pseudo-inverse of the one-step prediction covariance with numpy's default cutoff 1e-15 * sigma_max
 * (kalman_smoother.py:92): a Cholesky / Gauss-Jordan solve -- for 5 <= m <= 15 a Newton-Schulz iteration warm-started
 * from the previous step's inverse (kf_smoother_b4.hip) -- whenever that covariance is safely
 * positive definite, a Jacobi eigen-decomposition otherwise (m <= 64: given the per-draw flag
 * workspace -- without it the 47 < m <= 64 kernel returns NaN for such a draw; m > 64: flagged draws
 * are redone, the whole series, by one workgroup each).  `workspace`: see kf_workspace_bytes.
 * Cost of that semantics: with state noise of few columns (r = 1: every ARMA / SARIMA model) the prediction covariance sits on the
 * cut-off and most draws take the eigen-decomposition at every step -- 4096 x 1000, m = 10: 62 ms against 4 ms with full-rank
 * noise; 1024 x 300, m = 32 / 64: 2 s / 38 s (profiles/r05/smoother_dispatch_rank.txt, smoother_mid_m_rank.txt).  A caller that has
 * the filter's predicted tapes should use kf_smoother_from_filter below: no inverse, the same cost whatever the rank.
 * Time-varying T / R / Q with 5 <= m <= 16 run on the register-tile elimination kernel with per-step slices (round 5). */
KF_API int kf_smoother(kf_handle* h, const kf_dims* dims,
                const double* T, const double* R, const double* Q,
                const double* filt_a, const double* filt_P,
                double* smooth_a, double* smooth_P,
                void* workspace, size_t workspace_bytes, void* stream);

/* The same smoothed moments from the FILTER's own tapes (extension; kf_smoother_bf.hip).  The reference never runs its smoother on
 * anything but the outputs of its filter on the same model (core/statespace.py:164-197: build_statespace_graph hands
 * filtered_states / filtered_covariances straight to KalmanSmoother.build_graph), and with the filter's PREDICTED tapes and the
 * observation side of the model (y, Z, H, d) the smoothed moments follow without inverting the prediction covariance -- the
 * backward recursion of de Jong / Durbin & Koopman (what statsmodels computes):
 *     v = y - Z a_t - d,  F = Z P_t Z' + H,  K = P_t Z' / F,  L = T (I - K Z)         (a_t, P_t = pred_a[t], pred_P[t])
 *     r_{t-1} = Z' v / F + L' r_t,   N_{t-1} = Z' Z / F + L' N_t L,   smooth_a[t] = a_t + P_t r_{t-1},   smooth_P[t] = P_t - P_t N_{t-1} P_t
 * Equal to kf_smoother's output up to rounding (measured <= 2e-12 of a draw's largest entry at the BASELINE sizes, asserted at 1e-11;
 * small shapes at rtol = atol = 1e-9 against the restated reference recursion, tests/test_gpu_r05.py), 1.8x faster at the north-star
 * shape (4.65 -> 2.62 ms), every draw at the same cost, and a singular prediction covariance (the reference's pinv case) is no
 * special case.  Conditioning: smooth_P[t] = P_f - P_f M P_f subtracts terms of the size of the FILTERED covariance, as the
 * reference's G (P_s[t+1] - P_hat) G' does; with a near-diffuse start (P0 = 1e6 I) both lose the same ~10 digits in the first
 * steps and differ there by ~1e-10 of the largest filtered entry (the reference's own test leaves smoothed covariances [:5] out of
 * its comparison with statsmodels for this, tests/test_kalman_filter.py:236-241).  pred_a [B][T+1][m], pred_P [B][T+1][m][m] as kf_filter_forward wrote
 * them (slots 0 .. T-1 are read); y, Z, H, d, T as passed to it (d may be NULL); dims->filter: KF_FILTER_SINGLE keeps that
 * filter's sign of d (kalman_filter.py:335); dims->flags: KF_FLAG_Y_PER_DRAW only.  R, Q are not needed.
 * Supported: p == 1, static matrices, m <= 4 one lane per draw in three phases per block of steps (round 6, kf_smoother_bf_lane.hip: only
 * (r, N) is a recursion -- the filter's quantities of a block and its smoothed moments are computed side by side; C2 0.755 -> 0.246 ms and,
 * unlike the RTS lane kernel's pinv, 1e-9 from statsmodels on an ARMA model with H = 0), 5 <= m <= 15 one wavefront per draw on 4x4x4 block
 * MFMAs, 16 <= m <= 79 one workgroup of ceil((m + 1) / 16) wavefronts per draw on 16 x 16 fp64 MFMA tiles (round 6, kf_smoother_bf_wg.hip:
 * 1024 x 300, m = 32 / 63: 8.2 / 13.3 ms whatever the rank, against 43 / 77 on the workspace and 2.0 / 37.7 s of kf_smoother with r = 1; all
 * three without workspace) and 79 < m <= 1024 on the HBM workspace (also 16 <= m <= 79 with KF_SMOOTHER_BF_WG=0) with the fp64-MFMA
 * products of the m > 64 filter path (kf_smoother_bf_big.hip: two dense products and two with the block-sparse T per step instead of
 * kf_smoother's four products + blocked Cholesky + two triangular solves; workspace: kf_workspace_bytes_for(h, dims,
 * KF_WS_SMOOTHER_FROM_FILTER, &bytes) = 5 B m^2 doubles + vectors); anything else returns KF_ERR_UNSUPPORTED (use kf_smoother on
 * the filtered tapes).  KF_FLAG_EXACT_RECURSION is accepted and has no effect (every step is done in full). */
KF_API int kf_smoother_from_filter(kf_handle* h, const kf_dims* dims, const double* y, const double* T, const double* Z,
                            const double* H, const double* d, const double* pred_a, const double* pred_P,
                            double* smooth_a, double* smooth_P, void* workspace, size_t workspace_bytes, void* stream);

/* ---- the step before the filter ---------------------------
 * Batched theta -> matrix assembly: out[b][:] = tmpl[:], then out[b][dst[i]] = theta[b][src[i]]
 * for i < n_map.  `out` is one flat block of `total` doubles per draw that the caller carves
 * into (a0, P0, T, Z, R, H, Q); dst/src are device int32 index maps built once on the host from
 * the model structure.  Replaces the per-draw set_subtensor sequence of the models' update()
 * (pymc_statespace/models/SARIMAX.py:59-98, models/VARMAX.py:95-143). */
KF_API int kf_assemble(kf_handle* h, int64_t B, int32_t n_theta, const double* theta, int64_t total,
                const double* tmpl, int32_t n_map, const int32_t* dst, const int32_t* src,
                double* out, void* stream);

/* Batched discrete Lyapunov solve P = T P T' + S per draw, with S given ([B][m][m]) or, when
 * S is NULL, S = R Q R' from R [B][m][r], Q [B][r][r].  Replaces
 * solve_discrete_lyapunov(T, R Q R', method="bilinear") of the stationary initialisation
 * (models/SARIMAX.py:100-107, models/VARMAX.py:145-150).  info[b] = 0 ok, 1 = the doubling
 * iteration did not converge (spectral radius of T >= 1). */
KF_API int kf_lyapunov(kf_handle* h, int64_t B, int32_t m, int32_t r, const double* T, const double* R,
                const double* Q, const double* S, double* P, int32_t* info, void* stream);

/* ---- the step after the filter ----------------------------
 * Batched conditional simulation: out[(b*n_sims + j)][t][:] = mu[b][t][:] + L[b][t] z[b][j][t][:]
 * with L[b][t] = chol(cov[b][t] + jitter I); mu [B][n][k], cov [B][n][k][k] are a state tape
 * and its covariance tape (filtered / predicted / smoothed), z [B][n_sims][n][k] standard
 * normals supplied by the caller.  Replaces conditional_simulation / numba_mvn_draws
 * (pymc_statespace/utils/simulation.py:7-26).  info[b] != 0: a block was not positive
 * definite even with the jitter (encoded like kf_filter_forward's info). */
KF_API int kf_conditional_simulation(kf_handle* h, int64_t B, int32_t n, int32_t k, int32_t n_sims,
                              const double* mu, const double* cov, const double* z, double jitter,
                              double* out, int32_t* info, void* stream);

/* ---- SteadyStateFilter and unconditional simulation -----------
 * Batched steady-state covariance of the filter: the fixed point of
 *     P <- T (P - P Z' (Z P Z' + H)^-1 Z P) T' + R Q R'
 * per draw, i.e. solve_discrete_are(T', Z', R Q R', H) of SteadyStateFilter.build_graph
 * (pymc_statespace/filters/kalman_filter.py:388-390).  The recursion is iterated from P_init
 * ([B][m][m], NULL = R Q R') until the largest entrywise change is <= tol * (1 + max|P|) or
 * max_iter steps.  info[b]: 0 converged, 1 max_iter reached, 2 Z P Z' + H not positive
 * definite.  A SteadyStateFilter is then kf_filter_forward started from P (same results: at the
 * fixed point the reference's frozen F_inv equals the F^-1 of every step). */
KF_API int kf_dare(kf_handle* h, int64_t B, int32_t m, int32_t p, int32_t r, const double* T,
            const double* Z, const double* R, const double* H, const double* Q,
            const double* P_init, double* P, int32_t* info, int32_t max_iter, double tol,
            void* stream);

/* Reverse mode through kf_dare: given P ([B][m][m], the fixed point kf_dare returned) and g_P = d objective / d P,
 * the gradient with respect to the filter's own matrices, every entry independent.  Replaces SolveDiscreteARE.grad
 * (pymc_statespace/utils/pytensor_scipy.py:33-60, Kao & Hennequin 2020: one discrete Lyapunov solve with the
 * closed-loop matrix T' - Z' K) composed with A = T', B = Z', Q_d = R Q R', R_d = H of
 * SteadyStateFilter.build_graph (filters/kalman_filter.py:388).  Any output may be NULL; accumulate != 0 adds to what
 * the outputs hold -- the direct gradient kf_filter_backward wrote for the filter started from P, with g_P = its g_P0.
 * info[b] (may be NULL): 0 ok, 1 = the Lyapunov iteration did not converge, 2 = Z P Z' + H not positive definite. */
KF_API int kf_dare_backward(kf_handle* h, int64_t B, int32_t m, int32_t p, int32_t r, const double* T,
                     const double* Z, const double* R, const double* H, const double* Q,
                     const double* P, const double* g_P, double* g_T, double* g_Z, double* g_R,
                     double* g_H, double* g_Q, int32_t accumulate, int32_t* info, void* stream);

/* Batched unconditional simulation of the state space model, n_sims trajectories per draw:
 * simulate_statespace / unconditional_simulations (pymc_statespace/utils/simulation.py:29-79)
 * with their conventions -- innovations are `normals @ chol` (lower factor), the observation at
 * step t reads the state at t-1, row 0 is (x0, Z x0) or zeros when x0 is NULL, H == 0 means no
 * observation noise.  z_state [B][n_sims][n][r], z_obs [B][n_sims][n][p]: standard normals from
 * the caller; states [B*n_sims][n][m], obs [B*n_sims][n][p].  info[b]: 1 = Q, 2 = H not
 * positive definite.  m <= 64, r <= 16. */
KF_API int kf_simulate_statespace(kf_handle* h, int64_t B, int32_t n_sims, int32_t n, int32_t p,
                           int32_t m, int32_t r, const double* T, const double* Z,
                           const double* R, const double* H, const double* Q, const double* x0,
                           const double* z_state, const double* z_obs, double* states,
                           double* obs, int32_t* info, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KF_H_ */
