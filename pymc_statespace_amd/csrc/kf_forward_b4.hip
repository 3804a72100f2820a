// kf_forward_b4.hip -- forward Kalman filter on v_mfma_f64_4x4x4_4b_f64 (one wavefront per draw,
// scalar observation p == 1, k_states + 2 <= 16, static system matrices).
//
// Why a second matrix-core kernel: measured on MI355X (scripts/probes/f64_ops.hip, 4 wavefronts per SIMD)
// v_mfma_f64_16x16x4_f64 holds the fp64 pipe for 65 cycles, v_mfma_f64_4x4x4_4b_f64 (four independent
// 4x4x4 block products) for 10.2: 2.5 cycles per 4x4x4 block product against 4.1, and a 12 x 12 tile
// (the north-star shape: m = 10) is 3 x 3 blocks instead of a zero-padded 16 x 16 -- the two m^3
// products of a time step take 18 x 10.2 = 184 cycles instead of 6 x 65 = 390.
//
// Layout.  lane = 16 g + 4 b + q.  The instruction computes, per block b,
//     D_b[i][j] (lane g=i, q=j)  +=  sum_k A_b[i][k] (lane g=k, q=i) * B_b[k][j] (lane g=k, q=j)
// (measured with one-hot operands, scripts/probes/f64_ops.hip; cbsz / abid do NOT broadcast A blocks on
// this instruction -- scripts/probes/mfma4_bcast*.hip -- so a left factor must sit replicated in the four
// block slots of its register; the neg bits work).  An N4 x N4 matrix X (N4 = 4 NB) lives in NB
// registers, register I holding X[4I+g][4b+q] ("row-block layout": slot b of register I is the 4 x 4
// block X(I, b)).  Such a register is a valid B operand as it stands, so a product with a STATIC left
// factor (its blocks replicated over the slots once per launch) needs no data movement:
//     U(I, :)  = sum_K T(I, K) P(K, :)        : mfma(A = Trep[I][K], B = P[K])           NB^2 instructions
//     Y(I, :)  = sum_K T/2(I, K) U'(K, :)     : mfma(A = Trep[I][K]/2, B = U'[K], C = RQR'/2)
// with Y = (T P T' + RQR')/2 for symmetric P.  U' (row-block layout of the transpose) comes from one
// pass through an LDS tile, and P_hat = Y + Y' (kalman_filter.py:221) from a second pass of the same
// kind -- LDS traffic, not fp64-pipe time.
//
// The tile is the augmented second-moment matrix of (alpha, Z alpha, "1"):
//     index 0..m-1 : states          index JA = N4-2 : Z alpha (kf_forward_mfma.hip's AUG trick)
//     index JC = N4-1 : carrier of the MEAN -- row/column JC holds 2 a (and 2 Z a at JA), with
//     T_ext[JC][JC] = 1, C_ext[.][JC] = 2 c, so the predict products move the state vector along with
//     the covariance (a_hat = T a + c, kalman_filter.py:219) and no separate mean recursion exists.
// Update (kalman_filter.py:255-351, p = 1): with u = row JA of the tile, u_JC replaced by 2 Z a - 2 y
// (= -2 v) and F = H + u_JA, the whole update -- P_f = P - M M'/F, a_f = a + M v / F -- is ONE rank-one
// step  tile -= (u / sqrt F)(u / sqrt F)',  issued as NB more MFMAs (A = -u_hat, B = u_hat, k = 0 only);
// u / sqrt F keeps the tile bitwise symmetric.  Reference operation order differs (Joseph form); results
// agree to rounding (tests: 1e-8 on logp, 1e-9 on tapes).
#include <stdlib.h>

#include <type_traits>

#include "kf_b4.h"
#include "kf_b4_split.h"

namespace {

constexpr int LB = 32;                  // steps per y block / likelihood ring
// loop-time LDS (doubles); 16 wavefronts per CU must fit 160 KiB: <= 1280 doubles each
// (round 4: the tape rings come FIRST, each on a multiple of its own size -- their slots are kept as absolute LDS addresses and
// advanced with one bit-field insert, kf_b4.h TapeStream::next_abs -- then the transpose tile and the small buffers)
constexpr int RING_P = 256, GRAN_P = 128;
constexpr int RING_A = 128, GRAN_A = 64;
constexpr int B4_OFF_RP = 0;                      // covariance tapes: 2 rings of RING_P (or one m x m image + dump slots)
constexpr int B4_OFF_RA = B4_OFF_RP + 2 * RING_P; // state tapes: 2 rings of RING_A
constexpr int B4_OFF_TR = B4_OFF_RA + 2 * RING_A; // transpose tile, 16 rows of TRS
constexpr int B4_OFF_UX = B4_OFF_TR + 16 * TRS;   // u by column index (16) | zeros (4)
constexpr int B4_OFF_RG = B4_OFF_UX + 20;         // likelihood ring: LB x {F, u_hat_JC}
constexpr int B4_OFF_YB = B4_OFF_RG + 2 * LB;     // y block: LB x {-2 (y -+ d) or 0, observed ? 1 : 0}
constexpr int B4_OFF_DUMP = B4_OFF_YB + 2 * LB;   // one slot per lane: where lanes that have nothing to exchange write (no exec masking)
static_assert((B4_OFF_RP * 8) % (RING_P * 8) == 0 && (B4_OFF_RA * 8) % (RING_A * 8) == 0, "rings start on multiples of their size");
constexpr int B4_LOOP_DOUBLES = B4_OFF_DUMP + KF_WAVE;
static_assert(B4_LOOP_DOUBLES <= 1280, "four wavefronts per SIMD need <= 10 KiB of LDS per wavefront");
// Round 4: the step loop writes LDS WITHOUT exec masks.  Measured (profiles/r04/headline_pin.txt): a wave-step is 206 instructions
// at 4.6 SIMD cycles each, 70 of them scalar -- and half of those were the s_and_saveexec / s_or / s_cbranch_execz around ten
// predicated ds_writes (tape-ring appends of the lanes inside the m x m block, the u exchange of DPP row 0).  Now every lane
// writes: a lane without a tape element writes into the DEAD ZONE of its ring -- the GRAN - (elements per step) positions right
// behind this step's append, which the next steps overwrite before any window that contains them is flushed (a window leaves only
// when the append position has passed its end, and appends are contiguous) -- and a lane outside DPP row 0 exchanges into its own
// dump slot.
static_assert(RING_P >= 2 * GRAN_P && RING_A >= 2 * GRAN_A, "dead zone: a ring holds one unflushed window plus one being filled");
// (the covariance rings hold GRAN_P - 1 left-over doubles + one step: m * m <= RING_P - GRAN_P + 1, i.e. m <= 11; used for NB <= 3)

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// STREAM: covariance tapes through TapeStream rings (m <= 10), else one compact image per step.
// TM: which tapes are written, known at compile time -- bit 0 filtered states, 1 predicted states, 2 filtered covariances,
// 3 predicted covariances, 4 the internal tile tape; TM < 0: decided at run time (the uncommon combinations).  Measured
// (SQ counters, profiles/r02/ops_headline_sq_counters.txt): at four wavefronts per SIMD a wavefront's step costs ~4 cycles
// per instruction of ANY kind; the run-time tape switches alone were a dozen scalar instructions per step.
// SS: 0 the recursion at every step (KF_FLAG_EXACT_RECURSION); 1 the steady-state phase inside this kernel (no workspace); 2 one
// CHUNK of the split pipeline (kf_forward_b4_steady.hip): steps [A.t_begin, A.t_end) of the draws A.list names, the wavefront
// leaves as soon as its covariance has converged and hands the draw to the time-parallel steady kernels
// TM == 16 with SS == 1 ("VGS"): the forward half of kf_filter_value_and_grad with the steady-state adjoint behind it
// (kf_backward_steady.hip): the tile tape is written for the steps of the covariance recursion only; from the convergence point ts
// on the mean recursion leaves a_t in the compact predicted-state tape (slots ts .. n) -- all the steady-state adjoint reads -- and
// the draw reports ts (A.hs_ts) and its converged predicted covariance (A.vg_P)
// SS == 3 (round 6): one TIME CHUNK of the chunked pipeline for small batches (kf_forward_b4c.hip; KfFwdArgs::ck_*): workgroup i runs
// chunk i % ck_nc of draw i / ck_nc from that chunk's own predicted covariance -- the covariance sequence of a static model without
// missing observations does not depend on y, and its value at the chunk boundaries comes from log2(s) squarings of the s-step
// Riccati map -- with the mean started at c; the steps are the very arithmetic of SS == 0.  Covariance tapes are final; the state
// tapes, ll_obs and logp are the mean pass's (kf_b4c_mean_kernel: the mean recursion with the gains this kernel leaves in the workspace)
template <int NB, bool STREAM, int TM, int SS>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(4))) void kf_fwd_b4_kernel(const KfFwdArgs A) {
  // (waves_per_eu(4): <= 128 registers, and the compiler then keeps MFMA results in VGPRs -- no v_accvgpr moves)
  constexpr int N4 = 4 * NB, JA = N4 - 2, JC = N4 - 1;
  constexpr bool VGS = TM == 16 && SS == 1 && STREAM;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15;       // bq = 4 b + q: column index of the lane
  const int fb = (lane & 3) * 4 + ((lane >> 2) & 3);  // 4 q + b: position of column bq in a transpose-tile row
  long b = blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m;
  int t0 = 0, t1 = n;
  if constexpr (SS == 2) {
    if (A.list) {
      if ((int)blockIdx.x >= *A.count) return;
      b = A.list[blockIdx.x];
    }
    t0 = A.t_begin;
    t1 = A.t_end < n ? A.t_end : n;
  }
  if constexpr (VGS) {
    if (A.order) b = A.order[blockIdx.x];
  }
  if (A.run_if && *A.run_if == 0) return;   // (the sequential redo behind the chunked pipeline: only when that raised its flag)
  int cj = 0;   // (SS == 3) this workgroup's chunk
  if constexpr (SS == 3) {
    if (*A.ck_flag) return;
    cj = (int)(blockIdx.x % (unsigned)A.ck_nc);
    b = blockIdx.x / (unsigned)A.ck_nc;
    t0 = cj ? cj * A.ck_s + 1 : 0;
    t1 = (cj + 1) * A.ck_s + 1 < n ? (cj + 1) * A.ck_s + 1 : n;
  }

  // The tape rings' absolute slot addresses need the dynamic block on a multiple of the ring size -- true while it is the kernel's
  // only LDS (it then starts at 0).  The HOST checks that before every launch (b4_static_lds_free: hipFuncGetAttributes, static
  // shared size 0, else KF_ERR_UNSUPPORTED -- ADVICE round 4: a device trap would take the whole context down); should a build slip
  // through regardless, the draw comes back as NaN with its info word set instead of being written through misplaced rings
  if (lds_addr(smem) & (unsigned)(8 * RING_P - 1)) {
    if (lane == 0) {
      A.logp[b] = __builtin_nan("");
      if (A.info) A.info[b] = 0x7fffff01;
    }
    return;
  }
  double* TR = smem + B4_OFF_TR;
  double* STG = smem + B4_OFF_RP;  // !STREAM: compact m x m image (<= 196 doubles) + 64 dump slots at 256
  double* UX = smem + B4_OFF_UX;
  double* RG = smem + B4_OFF_RG;
  double* YB = smem + B4_OFF_YB;

  // ---- per-draw constants through an LDS scratch that aliases the loop buffers (kf_b4.h) ------------------------
  // (VGS: T / 2 is not kept as a second register set -- 18 registers that this instantiation, which carries the convergence test AND
  // the tile tape, would otherwise spill; the 1/2 is applied after the symmetrisation instead, three multiplies per step)
  constexpr bool HALF_STATIC = B4Const<NB>::HALF && !VGS;
  B4Const<NB, HALF_STATIC> KC;
  if (SS == 2 && t0 > 0) {
    // a later chunk of the split pipeline: the replicated T blocks and C_ext as the first chunk left them (21 coalesced loads
    // instead of the ~50 us of setup a full launch spends), and the zeroed transpose tile b4_setup leaves behind
    for (int e = lane; e < 16 * TRS; e += KF_WAVE) TR[e] = 0.0;
    const double* hc = A.hs_const + b * (long)B4_HC_STRIDE + lane;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        KC.Ta[I][K] = hc[(I * NB + K) * KF_WAVE];
        if (HALF_STATIC) KC.Tha[HALF_STATIC ? I : 0][HALF_STATIC ? K : 0] = hc[(NB * NB + I * NB + K) * KF_WAVE];
      }
      KC.Ch[I] = hc[(2 * NB * NB + I) * KF_WAVE];
    }
    wave_lds_fence();
  } else {
    if (SS == 3 && cj > 0)
      b4_setup(A, b, smem, lane, KC, TR, A.ck_start + ((long)b * A.ck_nc + cj) * mm, A.c ? A.c + b * (long)m : nullptr, A.c == nullptr);
    else
      b4_setup(A, b, smem, lane, KC, TR);
    if (SS == 2 && t1 < n) {
      double* hc = A.hs_const + b * (long)B4_HC_STRIDE + lane;
#pragma unroll
      for (int I = 0; I < NB; ++I) {
#pragma unroll
        for (int K = 0; K < NB; ++K) {
          hc[(I * NB + K) * KF_WAVE] = KC.Ta[I][K];
          if (HALF_STATIC) hc[(NB * NB + I * NB + K) * KF_WAVE] = KC.Tha[HALF_STATIC ? I : 0][HALF_STATIC ? K : 0];
        }
        hc[(2 * NB * NB + I) * KF_WAVE] = KC.Ch[I];
      }
    }
  }
  double (&Ta)[NB][NB] = KC.Ta;
  double (&Tha)[HALF_STATIC ? NB : 1][HALF_STATIC ? NB : 1] = KC.Tha;
  double (&Ch)[NB] = KC.Ch;
  double (&P)[NB] = KC.P;
  const double* gP = A.P0 + b * (long)mm;
  unsigned stg_off[NB];  // (!STREAM) LDS byte offset of this lane's element in the compact tape image (dump slot if none)
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int rr = 4 * I + g;
    stg_off[I] = (unsigned)(((rr < m && bq < m) ? rr * m + bq : 256 + lane) * 8);
  }
  if (lane < 20) UX[lane] = 0.0;
  wave_lds_fence();

  double H1 = A.H[b];
  asm volatile("" : "+v"(H1));  // VGPR: F = H + readlane(...) then is one instruction (one SGPR operand per VALU op)
  const double d1 = A.d ? A.d[b] : 0.0;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;

  // u exchange: DPP row 0 writes u[column] at position 4 q + b, reads {u[q], u[4+q], u[8+q], u[12+q]};
  // the other rows read zeros
  // Round 4: in the step loop u stays where the tile has it -- row JA = DPP row 2 of the last register -- and the rank-one MFMA
  // contracts over k = 2 instead of k = 0 (the move to row 0 was two v_permlane32_swap and four copies per step); the steady-state
  // loops below keep their row-0 layout
  double* ux_w = (g == 2) ? UX + fb : smem + B4_OFF_DUMP + lane;
  const double* ux_r = (g == 2) ? UX + 4 * (lane & 3) : UX + 16;
  const double mask2 = (g == 2 && bq < N4) ? 1.0 : 0.0;
  const double adjm = (lane == 32 + JC) ? 1.0 : 0.0;
  double* ux_w0 = (g == 0) ? UX + fb : smem + B4_OFF_DUMP + lane;   // (steady-state loops: u in DPP row 0)
  const double* ux_r0 = (g == 0) ? UX + 4 * (lane & 3) : UX + 16;
  const double mask0 = (g == 0 && bq < N4) ? 1.0 : 0.0;
  const bool vec_on = g == 3 && bq < m;                  // row JC of the tile = 2 a'
  // (a later chunk of the split pipeline continues the tapes at step t0: predicted slot t0 was written by the chunk before)
  const int ps0 = t0 ? t0 + 1 : 0;
  double* t_fa = A.filt_a ? A.filt_a + (b * (long)n + t0) * m : nullptr;
  double* t_pa = A.pred_a ? A.pred_a + (b * (long)(n + 1) + ps0) * m : nullptr;
  double* t_fP = A.filt_P ? A.filt_P + (b * (long)n + t0) * mm : nullptr;
  double* t_pP = A.pred_P ? A.pred_P + (b * (long)(n + 1) + ps0) * mm : nullptr;
  const bool w_fa = TM < 0 ? t_fa != nullptr : (TM & 1) != 0, w_pa = TM < 0 ? t_pa != nullptr : (TM & 2) != 0,
             w_fP = TM < 0 ? t_fP != nullptr : (TM & 4) != 0, w_pP = TM < 0 ? t_pP != nullptr : (TM & 8) != 0;

  // state tapes: row JC of the tile (2 a') appended to a ring, scaled by 1/2 on the way out
  TapeStream<RING_A, GRAN_A> s_fa, s_pa;
  s_fa.init(t_fa, smem + B4_OFF_RA);
  s_pa.init(t_pa, smem + B4_OFF_RA + RING_A);
  // (the step loop's writers are the lanes of tile row JC; every other lane appends into the dead zone, see above: m + 32 <= GRAN_A)
  const int ka = vec_on ? bq : m + (lane & 31);
  unsigned o_fa = s_fa.slot_abs(ka), o_pa = s_pa.slot_abs(ka);   // (absolute LDS addresses)
  auto put_a = [&](TapeStream<RING_A, GRAN_A>& st, unsigned& off, double twice_a) {
    lds_store_f64(off, twice_a);
    off = TapeStream<RING_A, GRAN_A>::next_abs(off, m);
    st.appended(m, lane, 0.5);
  };
  // the steady-state loops' writers are the lanes of DPP row 0: masked appends at per-column offsets (set at the phase's entry)
  auto put_a_row = [&](TapeStream<RING_A, GRAN_A>& st, unsigned& off, double twice_a, bool lanes_on) {
    if (lanes_on) lds_store_f64(off, twice_a);
    off = TapeStream<RING_A, GRAN_A>::next_abs(off, m);
    st.appended(m, lane, 0.5);
  };
  // covariance tapes
  TapeStream<RING_P, GRAN_P> s_fP, s_pP;
  s_fP.init(t_fP, smem + B4_OFF_RP);
  s_pP.init(t_pP, smem + B4_OFF_RP + RING_P);
  bool on[NB];
  unsigned o_fP[NB], o_pP[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int rr = 4 * I + g;
    on[I] = rr < m && bq < m;
    // (lanes outside the m x m block: the dead zone behind the step's mm elements -- GRAN_P - mm >= 7 positions for m <= 11)
    const int dz = GRAN_P - mm > 0 ? GRAN_P - mm : 1;
    const int kp = on[I] ? rr * m + bq : mm + (lane % dz);
    o_fP[I] = s_fP.slot_abs(kp);
    o_pP[I] = s_pP.slot_abs(kp);
  }
  auto put_P = [&](TapeStream<RING_P, GRAN_P>& st, unsigned (&off)[NB], const double (&Preg)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      lds_store_f64(off[I], Preg[I]);
      off[I] = TapeStream<RING_P, GRAN_P>::next_abs(off[I], mm);
    }
    st.appended(mm, lane, 1.0);
  };
  // !STREAM (m * m too large for the rings): compact m x m image in LDS -> 16 bytes per lane, one step at a time
  const int nb128 = mm >> 1;
  auto stage_cov = [&](const double (&Preg)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
      *reinterpret_cast<double*>(reinterpret_cast<char*>(STG) + stg_off[I]) = Preg[I];
  };
  const bool st1 = lane < nb128, st1b = (mm & 1) && lane == nb128;
  const bool st2 = lane + KF_WAVE < nb128, st2b = (mm & 1) && lane + KF_WAVE == nb128;
  const bool two_rounds = mm > 2 * KF_WAVE;  // m > 11
  const bool odd_tail = (mm & 1) != 0;
  auto flush_cov = [&](double* dst) {
    wave_lds_fence();
    const double2_t v2 = *reinterpret_cast<const double2_t*>(STG + 2 * lane);
    if (st1) st_stream(reinterpret_cast<double2_t*>(dst) + lane, v2);
    if (odd_tail && st1b) st_stream(dst + mm - 1, v2.x);
    if (two_rounds) {
      const double2_t w2 = *reinterpret_cast<const double2_t*>(STG + 2 * (lane + KF_WAVE));
      if (st2) st_stream(reinterpret_cast<double2_t*>(dst) + lane + KF_WAVE, w2);
      if (odd_tail && st2b) st_stream(dst + mm - 1, w2.x);
    }
    wave_lds_fence();
  };
  bool pred_pending = false;

  // internal tile tape (kf_filter_value_and_grad; read back by kf_backward_b4.hip in the same register layout): per draw
  // T + 1 slots of N4 x N4 doubles, row-major, stored straight from the registers (no staging).  Slot t < T = what the
  // adjoint of step t needs: the FILTERED tile, its row JA (never used there) replaced by u_hat with 1 / sqrt(F) at
  // [JA][JA] (0 on a missing / skipped step); slot T = the predicted tile of step 1 (the de-augmentation at the hand-over)
  const bool w_tile = TM < 0 ? A.tile_tape != nullptr : (TM & 16) != 0;
  double* t_tile = w_tile ? A.tile_tape + b * (long)(n + 1) * (N4 * N4) + g * N4 + bq : nullptr;
  auto put_tile = [&](double* dst, const double (&X)[NB], double last) {
    if (bq < N4) {
#pragma unroll
      for (int I = 0; I < NB - 1; ++I) dst[I * 4 * N4] = X[I];
      dst[(NB - 1) * 4 * N4] = last;
    }
  };
  double logp = 0.0;
  int info = 0;
  if constexpr (SS == 2) {
    if (t0 > 0) {  // the tile, log-likelihood and flag the previous chunk left
      const double* ht = A.hs_tile + b * (long)B4_HS_STRIDE + lane;
#pragma unroll
      for (int I = 0; I < NB; ++I) P[I] = ht[I * KF_WAVE];
      logp = A.hs_logp[b];
      info = A.hs_info[b];
    }
  }
  // predicted tape slot 0 = (a0, P0 as given: the caller's P0, not its symmetrised upper triangle)
  if (w_pa && t0 == 0) put_a(s_pa, o_pa, P[NB - 1]);
  if (w_pP && t0 == 0) {
    if (STREAM) {
      double P0g[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) P0g[I] = on[I] ? gP[(4 * I + g) * m + bq] : 0.0;
      put_P(s_pP, o_pP, P0g);
    } else {
      for (int e = lane; e < mm; e += KF_WAVE) t_pP[e] = gP[e];
      t_pP += mm;
    }
  }

  // row-block layout of X -> row-block layout of X' through the LDS tile (wave-local: LDS is in order)
  auto transpose_tile = [&](const double (&X)[NB], double (&Xt)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = X[I];
    wave_lds_fence();
    const double* src = TR + bq * TRS + 4 * g;
    const double2_t t01 = *reinterpret_cast<const double2_t*>(src);
    double2_t t23 = {0.0, 0.0};
    if (NB > 2) t23 = *reinterpret_cast<const double2_t*>(src + 2);
    wave_lds_fence();
    Xt[0] = t01.x;
    if (NB > 1) Xt[NB > 1 ? 1 : 0] = t01.y;
    if (NB > 2) Xt[NB > 2 ? 2 : 0] = t23.x;
    if (NB > 3) Xt[NB > 3 ? 3 : 0] = t23.y;
  };
  // ll of the steps [tb, tb+cnt) held in the ring, one step per lane: (F, u_hat_JC = -2 v / sqrt F); the flags of
  // these steps are still in the y block.  Everything rare is decided here, LB steps at a time.
  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
    const int ls = lane & (LB - 1);
    const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * ls);
    const bool act = lane < cnt && (SS != 3 || tb + lane >= t0);   // (a chunk may start inside a block: the ring's earlier slots are not its own)
    const bool observed = act && YB[2 * ls + 1] != 0.0;
    const double Fv = fq.x;
    if constexpr (SS == 3) {
      // the likelihood is the mean pass's (v of the draw's own mean); a missing observation, a skipped step or an F that is not
      // positive is the sequential kernel's business: raise the flag
      if (__ballot(act && !(observed && Fv > 0.0)) != 0ull && lane == 0) *A.ck_flag = 1;
      if (act) *reinterpret_cast<double2_t*>(A.ck_fv + 2 * (b * (long)n + tb + lane)) = fq;
      __syncthreads();
      return;
    }
    const bool skipped = observed && univariate && Fv == 0.0;      // :468-471
    const bool notpd = observed && !skipped && !(Fv > 0.0);
    const unsigned long long bad = __ballot(notpd);
    if (bad != 0ull && info == 0) info = ((tb + __ffsll((long long)bad)) << 8) | 1;
    const double quad = 0.25 * fq.y * fq.y;                          // v^2 / F
    double llv;
    if (!observed) llv = 0.0;
    else if (skipped) llv = exact_const ? -0.5 * KF_LOG_2PI : 0.0;
    else if (univariate && !exact_const && Fv == 1.0 && quad == 0.0) llv = 0.0;  // :497-500
    else llv = -0.5 * (KF_LOG_2PI + log(Fv) + quad);
    if (act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
    __syncthreads();
  };

  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);

  auto load_y_block = [&](int t) {
    const int cnt = (n - t) < LB ? (n - t) : LB;
    if (lane < cnt) {
      const double yv = ybase[(long)t + lane];
      const bool ob = !(yv != yv);
      double2_t e = {ob ? -2.0 * (yv + dsign * d1) : 0.0, ob ? 1.0 : 0.0};
      *reinterpret_cast<double2_t*>(YB + 2 * lane) = e;
    }
    wave_lds_fence();
  };

  // ---- steady state ------------------------------------------------------------------------------------------------------------
  // With static matrices and no missing observation ahead, the covariance recursion is a Riccati iteration that converges
  // geometrically; once the predicted covariance has stopped moving (to 1e-15 of its largest entry, tested every 8 steps)
  // the gain, F and both covariances are constants and only the mean is left: a_{t+1} = (T - T K Z) a_t + T K (y_t -+ d) + c
  // -- the loop at the end of this kernel.  (statsmodels' Kalman filter does the same by default, `tolerance`; the
  // reference recomputes everything every step, results agree to ~1e-13 relative.  KF_FLAG_EXACT_RECURSION switches it off.)
  const bool steady_en = (SS == 1 || SS == 2) && STREAM && (!w_tile || VGS) && n >= 64;
  int t_free = 0;  // first step after the last missing observation
  if (SS == 2 && t0 > 0) {
    t_free = A.hs_tfree[b];
  } else if (steady_en) {
    int tn = -1;
    for (int e = lane; e < n; e += KF_WAVE) {
      const double yv = ybase[e];
      if (yv != yv) tn = e;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) tn = max(tn, __shfl_xor(tn, off));
    t_free = __builtin_amdgcn_readfirstlane(tn + 1);   // (wave-uniform: the convergence test's branch is a scalar one)
    if (SS == 2 && lane == 0) A.hs_tfree[b] = t_free;
  }

  if (SS == 2) __builtin_amdgcn_s_setprio(3);  // the recursion's dependency chain before the steady kernels that share the SIMD
  int t = t0;
  int npass = 0;       // consecutive convergence checks passed
  bool handed = false; // (SS == 2) the covariance has converged: the steady kernels take over at step t
  for (; t < t1; ++t) {
    if ((t & (LB - 1)) == 0 || (SS == 3 && t == t0)) load_y_block(t & ~(LB - 1));
    const bool chk = steady_en && (t & 7) == 7 && t >= t_free && t + 16 < n;  // wave-uniform
    if (chk && !w_pP) {  // the predicted covariance this step starts from, parked in the (unused) tape ring
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (on[I]) lds_store_f64(o_pP[I], P[I]);
    }
    const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (t & (LB - 1)));  // {-2 (y -+ d), observed}
    // ---- update: u = row JA of the tile (DPP row 2 of the last register), in place ---------------------------------------
    const double u = fma(P[NB - 1], mask2, adjm * yo.x);  // u_JC = 2 Z a - 2 (y -+ d) = -2 v
    // left factor of the rank-one step: block I of u replicated over the four slots (k = 2: DPP row 2 only).
    // One LDS exchange, issued before F / rs are known: u_rep[I] at lane (g = 2, any b, q) = u[4 I + q]
    *ux_w = u;  // (the other rows: their dump slots)
    wave_lds_fence();
    const double2_t ur01 = *reinterpret_cast<const double2_t*>(ux_r);
    double2_t ur23 = {0.0, 0.0};
    if (NB > 2) ur23 = *reinterpret_cast<const double2_t*>(ux_r + 2);
    wave_lds_fence();
    const double urep[4] = {ur01.x, ur01.y, ur23.x, ur23.y};
    const double F = H1 + readlane_f64(u, 32 + JA);
    // missing -> no update (rs = 0).  F <= 0: flagged when the ring is evaluated (info, NaN ll); the clamp only
    // keeps 0 * rsqrt(F) finite on a missing / skipped step (Univariate skips F == 0, :468-471: u is 0 there)
    const double rs = fast_rsqrt(fmax(F, 1e-300)) * yo.y;
    const double uh = u * rs;
    if (lane == 32 + JC) {
      double2_t fq = {F, uh};
      *reinterpret_cast<double2_t*>(RG + 2 * (t & (LB - 1))) = fq;
    }
    // tile -= u_hat u_hat'
#pragma unroll
    for (int I = 0; I < NB; ++I) P[I] = mfma4_nega(urep[I] * rs, uh, P[I]);
    if constexpr (SS == 3) {
      if (g == 2 && bq < m) A.ck_K[(b * (long)n + t) * m + bq] = uh * rs;                                  // K_t = P_t Z' / F_t
      if (t == t1 - 1 && vec_on) A.ck_zend[((long)b * A.ck_nc + cj) * m + bq] = 0.5 * P[NB - 1];          // a_f of the chunk's last step
    }
    if (w_tile) {
      const double rs_t = (univariate && F == 0.0) ? 0.0 : rs;  // a skipped step has no adjoint terms (:468-471)
      const double rowv = (lane == 32 + JA) ? rs_t : uh;          // (u_hat sits in row JA of the last register already)
      put_tile(t_tile + (long)t * (N4 * N4), P, (g == (JA & 3)) ? rowv : P[NB - 1]);
    }
    if (!STREAM && pred_pending) {  // (image mode) predicted covariance of the previous step: tile image -> HBM
      flush_cov(t_pP);
      t_pP += mm;
      pred_pending = false;
    }
    // ---- filtered tapes -----------------------------------------------------------------------------------------------
    if (w_fa) put_a(s_fa, o_fa, P[NB - 1]);
    if (w_fP) {
      if (STREAM) put_P(s_fP, o_fP, P);
      else stage_cov(P);  // flushed below, behind the MFMAs
    }
    // ---- predict: tile <- sym(T_ext tile T_ext' + C_ext) ---------------------------------------------------------------------
    double U[NB], Y[NB];  // (first product of each chain: the literal 0 as C operand is an inline constant -- no v_mov per step)
#pragma unroll
    for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][0], P[0], 0.0);
#pragma unroll
    for (int K = 1; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][K], P[K], U[I]);
    if (!STREAM && w_fP) {
      flush_cov(t_fP);
      t_fP += mm;
    }
    // pass 1: U' through the tile -- element (i, j) at TR[i * TRS + 4 (j & 3) + (j >> 2)]
    transpose_tile(U, U);
#pragma unroll
    for (int I = 0; I < NB; ++I) Y[I] = Ch[I];
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I)
        Y[I] = mfma4(HALF_STATIC ? Tha[HALF_STATIC ? I : 0][HALF_STATIC ? K : 0] : Ta[I][K], U[K], Y[I]);
    // pass 2: P_hat = Y + Y'
    {
      double Yt[NB];
      transpose_tile(Y, Yt);
#pragma unroll
      for (int I = 0; I < NB; ++I) P[I] = HALF_STATIC ? Y[I] + Yt[I] : 0.5 * (Y[I] + Yt[I]);
    }
    bool conv = false;
    if (chk) {
      // Converged: on TWO consecutive checks (8 steps apart) the one-step change of the predicted covariance is within 1e-15 of
      // its largest entry AND every entry's change is within 1e-12 of its own scale sqrt(P_ii P_jj).  The second, entrywise
      // condition is what keeps a large (diffuse) block from hiding a small block that still moves; it cannot be as tight as the
      // first: an entry far below its scale carries the rounding noise of the products it comes from, ~1e-15 .. 1e-13
      // sqrt(P_ii P_jj) per step (measured on the VARMAX draws, m = 15, scripts/r03/ts_hist.py: at 1e-13 one draw in 200 never
      // passes, at 3e-13 one in 1400, at 1e-12 none).  The frozen covariance is then within tol / (1 - rho) of the fixpoint
      // (rho: contraction per step).
      // (with the predicted-covariance tape on, the previous step's append still sits in the ring, mm doubles back)
      wave_lds_fence();
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (on[I] && 4 * I + g == bq) UX[bq] = P[I];  // the diagonal (UX is rewritten in full at the top of the next step)
      wave_lds_fence();
      const double dcol = UX[bq < m ? bq : 0];
      unsigned pm = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (on[I]) pm = max(pm, abs_hi(P[I]));
      const double lim = 1e-15 * wave_absmax_hi(pm);  // (the largest entry, from the high dwords: within 2^-20)
      bool bad = false;
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        const unsigned oo = w_pP ? TapeStream<RING_P, GRAN_P>::back_abs(o_pP[I], mm) : o_pP[I];
        const double old = lds_load_f64(oo);
        const double drow = UX[on[I] ? 4 * I + g : 0];
        const double df = P[I] - old;
        bad = bad || (on[I] && !(fabs(df) <= lim && df * df <= A.ss_tol2 * fabs(dcol * drow)));  // (NaN: never converged)
      }
      wave_lds_fence();
      const bool ok = __ballot(bad) == 0ull;
      npass = ok ? npass + 1 : 0;
      conv = npass >= 2;
    }
    if (w_tile && t == 0) put_tile(t_tile + (long)n * (N4 * N4), P, P[NB - 1]);
    if (w_pa) put_a(s_pa, o_pa, P[NB - 1]);
    if (w_pP) {
      if (STREAM) {
        put_P(s_pP, o_pP, P);
      } else {
        stage_cov(P);
        pred_pending = true;
      }
    }
    if ((t & (LB - 1)) == LB - 1) flush_ring(t - (LB - 1), LB);
    if (conv) {
      ++t;
      handed = true;
      break;
    }
  }
  if constexpr (VGS) {
    // t = the first step of the steady-state phase (n: the covariance never converged); P = the predicted tile of step t
    if (lane == 0) A.hs_ts[b] = t;
    double* vp = A.vg_P + b * (long)mm;
#pragma unroll
    for (int I = 0; I < NB; ++I)
      if (on[I]) vp[(4 * I + g) * m + bq] = P[I];
    if (t < n) {  // a_t .. a_n into slots t .. n of the compact predicted-state tape
      s_pa.init(A.pred_a + (b * (long)(n + 1) + t) * m, smem + B4_OFF_RA + RING_A);
      o_pa = s_pa.slot_abs(bq);
      put_a_row(s_pa, o_pa, P[NB - 1], vec_on);
    }
  }
  if constexpr (SS == 1 && STREAM) if (t < n && !(VGS && A.vg_stop)) {
    // ---- steady-state steps t .. n-1 (see above); P is the converged predicted tile of step t -------------------------------
    if constexpr (!VGS) {  // this phase's state-tape writers are the lanes of DPP row 0: per-column offsets from here on
      o_fa = s_fa.slot_abs(bq);
      o_pa = s_pa.slot_abs(bq);
    }
    const double us = high_half_everywhere(P[NB - 1]) * mask0;  // row JA: (P Z')' with Z P Z' at JA (DPP row 0)
    const double F = H1 + readlane_f64(us, JA);
    const double rs = fast_rsqrt(fmax(F, 1e-300));
    const double uh = us * rs;
    // filtered covariance: one more rank-one step (its mean row / column is not used)
    *ux_w0 = us;
    wave_lds_fence();
    double Pf[NB];
    {
      const double2_t ur01 = *reinterpret_cast<const double2_t*>(ux_r0);
      double2_t ur23 = {0.0, 0.0};
      if (NB > 2) ur23 = *reinterpret_cast<const double2_t*>(ux_r0 + 2);
      wave_lds_fence();
      const double urep[4] = {ur01.x, ur01.y, ur23.x, ur23.y};
#pragma unroll
      for (int I = 0; I < NB; ++I) Pf[I] = mfma4_nega(urep[I] * rs, uh, P[I]);
    }
    // gain K = P Z' / F by column index, through the (idle) transpose tile; T K per lane
    if (g == 0) TR[bq] = (bq < m) ? uh * rs : 0.0;
    wave_lds_fence();
    const double* gT = A.Tm + b * (long)mm;
    const double* gZ = A.Z + b * (long)m;
    const bool jst = bq < m;
    const double Kj = TR[bq];
    double tkj = 0.0;
    if (jst)
      for (int l = 0; l < m; ++l) tkj = fma(gT[bq * m + l], TR[l], tkj);
    const double cj = (jst && A.c) ? A.c[b * (long)m + bq] : 0.0;
    // the two maps of the extended mean (a, y~ at JA, 1 at JC), rows = input index, columns = output index:
    //   M1: a_{t+1} = (T - T K Z) a + T K y~ + c, and column JA = Z a (the innovation's prediction)
    //   M2: a_f     = (I - K Z) a + K y~
    double M1[NB], M2[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int i = 4 * I + g;
      const double zi = (i < m) ? gZ[i] : 0.0;
      double v1 = 0.0, v2 = 0.0;
      if (i < m && jst) {
        v1 = gT[bq * m + i] - tkj * zi;
        v2 = ((i == bq) ? 1.0 : 0.0) - Kj * zi;
      } else if (i < m && bq == JA) {
        v1 = zi;
      } else if (i == JA && jst) {
        v1 = tkj;
        v2 = Kj;
      } else if (i == JC && jst) {
        v1 = cj;
      }
      M1[I] = v1;
      M2[I] = v2;
    }
    wave_lds_fence();
    // the mean as a one-row left factor: element j at UX[4 (j & 3) + (j >> 2)], lane (g = k, q = 0) reads {a[k], a[4 + k], ..}
    constexpr int PJA = 4 * (JA & 3) + (JA >> 2), PJC = 4 * (JC & 3) + (JC >> 2);
    if (g == 3) UX[fb] = 0.5 * P[NB - 1];  // row JC of the tile = 2 a_t
    const double* lq_r = ((lane & 3) == 0) ? UX + 4 * g : UX + 16;
    const bool row0 = g == 0 && jst;
    for (; t < n; ++t) {
      if ((t & (LB - 1)) == 0) load_y_block(t);
      const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (t & (LB - 1)));
      const double yt = -0.5 * yo.x;  // y -+ d
      if (lane == 0) {
        UX[PJA] = yt;
        UX[PJC] = 1.0;
      }
      wave_lds_fence();
      const double2_t l01 = *reinterpret_cast<const double2_t*>(lq_r);
      double2_t l23 = {0.0, 0.0};
      if (NB > 2) l23 = *reinterpret_cast<const double2_t*>(lq_r + 2);
      wave_lds_fence();
      const double lf[4] = {l01.x, l01.y, l23.x, l23.y};
      double o1 = mfma4(lf[0], M1[0], 0.0), o2 = mfma4(lf[0], M2[0], 0.0);
#pragma unroll
      for (int K = 1; K < NB; ++K) {
        o1 = mfma4(lf[K], M1[K], o1);
        o2 = mfma4(lf[K], M2[K], o2);
      }
      if (lane == JA) {  // o1 = Z a_t here
        double2_t fq = {F, -2.0 * (yt - o1) * rs};
        *reinterpret_cast<double2_t*>(RG + 2 * (t & (LB - 1))) = fq;
      }
      *ux_w0 = o1;  // a_{t+1} (its JA / JC slots are rewritten at the top of the next step)
      if (w_fa) put_a_row(s_fa, o_fa, o2 + o2, row0);
      // (tried: both constant covariance tapes written in one go at the start of this phase, as kf_forward_b4q.hip does -- headline
      // 1.81 ms against 1.63, C3 6.7 against 6.2: the burst of stores delays the window flushes of the SIMD's other wavefronts)
      if (w_fP) put_P(s_fP, o_fP, Pf);
      if (w_pa || VGS) put_a_row(s_pa, o_pa, o1 + o1, row0);
      if (w_pP) put_P(s_pP, o_pP, P);
      if ((t & (LB - 1)) == LB - 1) flush_ring(t - (LB - 1), LB);
    }
  }
  if (!STREAM && pred_pending) flush_cov(t_pP);
  s_fa.finish(lane, 0.5);
  s_pa.finish(lane, 0.5);
  if (STREAM) {
    s_fP.finish(lane, 1.0);
    s_pP.finish(lane, 1.0);
  }
  if (t & (LB - 1)) flush_ring(t & ~(LB - 1), t & (LB - 1));  // (t == n, or the step a chunk stopped at)
  if constexpr (SS == 3) {
    // P is the predicted tile of step t1 -- where the next chunk of this draw started from the doubling's covariance: the two must
    // agree (ck_tol of the largest entry), else nobody vouches for the batch and the sequential kernel redoes it
    if (cj + 1 < A.ck_nc) {
      const double* ns = A.ck_start + ((long)b * A.ck_nc + cj + 1) * mm;
      unsigned pm = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (on[I]) pm = max(pm, abs_hi(P[I]));
      const double lim = A.ck_tol * wave_absmax_hi(pm);
      bool bad = false;
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (on[I]) bad = bad || !(fabs(P[I] - ns[(4 * I + g) * m + bq]) <= lim);
      if (__ballot(bad) != 0ull && lane == 0) *A.ck_flag = 1;
    }
    return;   // (logp, info, ll_obs and the draw's own means: kf_b4c_mean_kernel)
  }
  if constexpr (SS == 2) {
    if (t < n) {  // hand-over: to the next chunk (-1) or, converged, to the steady kernels (step t)
      double* ht = A.hs_tile + b * (long)B4_HS_STRIDE + lane;
#pragma unroll
      for (int I = 0; I < NB; ++I) ht[I * KF_WAVE] = P[I];
      if (lane == 0) {
        A.hs_logp[b] = logp;
        A.hs_info[b] = info;
        A.hs_ts[b] = handed ? t : -1;
      }
      return;
    }
    if (lane == 0) A.hs_ts[b] = -2;
  }
  if (lane == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}


// ---- time-varying system matrices on the block tiles (round 4) --------------------------------------------------------------------
// Reference: a 3-D system matrix is a sequence, time on axis 0, and step t uses slice t for its update AND its predict
// (filters/utilities.py:1-20, kalman_filter.py:79-124, 231-252).  Rounds 1-3 sent every KF_FLAG_TV_* call to the round-1 LDS kernel
// (headline shape with a time-varying T: 12.9 ms = 0.10 of the HBM peak, profiles/r04/tv_dense_lines.txt).  This kernel is the block
// kernel above with what a static model lets it hoist taken back into the step:
//   * no "Z alpha" row in the tile (its entries would need Z_{t+1} T_t): u = S z_t is three MFMAs with z_t as a one-row left factor
//     (DPP row 0 gets z' S; F = H_t + (z' S) z_t by a row sum), the rank-one update contracts over k = 0;
//   * T_t's replicated left-factor blocks, z_t in both layouts and c_t are loaded per step -- each right AFTER its last use in the
//     step, so the loads fly during the rest of it -- ; H_t and d_t come with the 32-step observation block;
//   * R / Q stay static (a time-varying R Q R' is m^2 r + m r^2 products per step: those calls keep the LDS kernel), no steady-state
//     phase, no tile tape.
// Tapes through the same LDS rings, dead zones and absolute slot addresses as above.
// TVM: the flags this instantiation may meet (a superset of the call's): code for the others is not compiled in -- with all five
// paths in one body the compiler kept 25 pointers and 9 lane masks live and spilled them (40 v_readlane per step)
template <int NB, unsigned TVM>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(4))) void kf_fwd_b4tv_kernel(const KfFwdArgs A) {
  constexpr int N4 = 4 * NB, JC = N4 - 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15, q = lane & 3;
  const int fb = q * 4 + ((lane >> 2) & 3);
  const long b = blockIdx.x;
  const int n = A.T, m = A.m, mm = m * m, r_ = A.r;
  const bool tvT = (TVM & KF_FLAG_TV_T) && (A.flags & KF_FLAG_TV_T), tvZ = (TVM & KF_FLAG_TV_Z) && (A.flags & KF_FLAG_TV_Z);
  const bool tvH = (TVM & KF_FLAG_TV_H) && (A.flags & KF_FLAG_TV_H);
  const bool tvC = (TVM & KF_FLAG_TV_C) && (A.flags & KF_FLAG_TV_C) && A.c, tvD = (TVM & KF_FLAG_TV_D) && (A.flags & KF_FLAG_TV_D) && A.d;
  const bool tvR = (TVM & KF_FLAG_TV_R) && (A.flags & KF_FLAG_TV_R), tvQ = (TVM & KF_FLAG_TV_Q) && (A.flags & KF_FLAG_TV_Q);
  constexpr bool RQM = (TVM & (KF_FLAG_TV_R | KF_FLAG_TV_Q)) != 0;   // this instantiation rebuilds C_t = R_t Q_t R_t' per step (r <= 4)
  if (lds_addr(smem) & (unsigned)(8 * RING_P - 1)) {   // (see kf_fwd_b4_kernel: checked on the host before the launch)
    if (lane == 0) {
      A.logp[b] = __builtin_nan("");
      if (A.info) A.info[b] = 0x7fffff01;
    }
    return;
  }
  double* TR = smem + B4_OFF_TR;
  double* UX = smem + B4_OFF_UX;
  double* RG = smem + B4_OFF_RG;
  double* YB = smem + B4_OFF_YB;
  double* HB = smem + B4_OFF_DUMP;  // LB entries of H_t in the first half of the dump slots; the second half takes the u exchange of rows 1 .. 3
  for (int e = lane; e < B4_LOOP_DOUBLES; e += KF_WAVE) smem[e] = 0.0;
  wave_lds_fence();

  // ---- per-draw constants and the first slices -------------------------------------------------------------------------------------
  const double* gT = A.Tm + (tvT ? b * (long)n : b) * mm;   // slice 0
  const double* gZ = A.Z + (tvZ ? b * (long)n : b) * m;
  const double* gc = A.c ? A.c + (tvC ? b * (long)n : b) * m : nullptr;
  const double* gR = A.R + b * (long)m * r_;
  const double* gQ = A.Q + b * (long)r_ * r_;
  const double* gP = A.P0 + b * (long)mm;
  double Ta[NB][NB], zA[NB], Ch[NB], P[NB];
  bool on[NB], cJ[NB];        // the m x m block; the carrier entries (row < m, column JC) and (row JC, column < m)
  int cidx[NB];               // ... and which entry of c they take
  // T_t's block (I, K) is a wave-uniform base plus ONE lane offset (q m + g): scalar base + 32-bit vector offset loads; the lane
  // masks are three row and three column predicates
  const int t_lo = q * m + g;
  bool rowok[NB], colok[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    rowok[I] = 4 * I + q < m;
    colok[I] = 4 * I + g < m;
  }
  auto load_T = [&](const double* Tt) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const double* blk = Tt + (4 * I * m + 4 * K);
        const double fix = (I == NB - 1 && K == NB - 1 && q == 3 && g == 3) ? 1.0 : 0.0;   // T_ext[JC][JC] = 1
        Ta[I][K] = (rowok[I] && colok[K]) ? blk[t_lo] : fix;
      }
  };
  auto load_z = [&](const double* zt, double& zc) {
#pragma unroll
    for (int K = 0; K < NB; ++K) zA[K] = (q == 0 && 4 * K + g < m) ? zt[4 * K + g] : 0.0;   // one-row left factor
    zc = (g == 0 && bq < m) ? zt[bq] : 0.0;                                                  // row-0 layout
  };
  // Round 5: time-varying R / Q (r <= 4).  C_t = R_t (Q_t R_t') is four MFMAs on per-step slices: Q R' (r x N4, one register) from
  // Q_t as a left factor and R_t' as a right factor, then one MFMA per row block with R_t's blocks as left factors -- no LDS; the
  // symmetrisation of the predict step (kalman_filter.py:221) makes the tile symmetric whatever the rounding of C_t's two halves.
  double Rl[RQM ? NB : 1], Rt = 0.0, Ql = 0.0;   // R_t[4I+q][g], R_t[bq][g], Q_t[q][g]  (g < r)
  auto load_R = [&](const double* Rs) {
    if constexpr (RQM) {
#pragma unroll
      for (int I = 0; I < NB; ++I) Rl[I] = (4 * I + q < m && g < r_) ? Rs[(4 * I + q) * r_ + g] : 0.0;
      Rt = (bq < m && g < r_) ? Rs[bq * r_ + g] : 0.0;
    }
  };
  auto load_Q = [&](const double* Qs) {
    if constexpr (RQM) Ql = (q < r_ && g < r_) ? Qs[q * r_ + g] : 0.0;
  };
  double zc;
  load_T(gT);
  load_z(gZ, zc);
  load_R(A.R + (tvR ? b * (long)n : b) * (long)m * r_);
  load_Q(A.Q + (tvQ ? b * (long)n : b) * (long)r_ * r_);
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int rr = 4 * I + g;
    on[I] = rr < m && bq < m;
    cJ[I] = (rr < m && bq == JC) || (rr == JC && bq < m);
    cidx[I] = rr < m ? rr : (bq < m ? bq : 0);
    double cv = 0.0, pv = 0.0;
    if (on[I]) {
      double s1 = 0.0, s2 = 0.0;   // R Q R' at (rr, bq) and at (bq, rr): their mean is bitwise symmetric across the mirrored lanes
      if constexpr (!RQM)          // (time-varying R / Q: rebuild_C below, from the slices)
      for (int l = 0; l < r_; ++l) {
        double rq1 = 0.0, rq2 = 0.0;
        for (int k = 0; k < r_; ++k) {
          rq1 += gR[rr * r_ + k] * gQ[k * r_ + l];
          rq2 += gR[bq * r_ + k] * gQ[k * r_ + l];
        }
        s1 += rq1 * gR[bq * r_ + l];
        s2 += rq2 * gR[rr * r_ + l];
      }
      cv = 0.5 * (s1 + s2);
      pv = (bq >= rr) ? gP[rr * m + bq] : gP[bq * m + rr];
    } else if (cJ[I]) {
      pv = 2.0 * A.a0[b * (long)m + cidx[I]];
    }
    Ch[I] = (cJ[I] && gc) ? 2.0 * gc[cidx[I]] : cv;   // C_t: R Q R' on the state block, 2 c_t on the carrier entries (disjoint lanes)
    P[I] = pv;
  }

  auto rebuild_C = [&]() {   // the state block of C_t from the slices in Rl / Rt / Ql; the carrier entries (2 c_t) stay
    if constexpr (RQM) {
      const double QRt = mfma4(Ql, Rt, 0.0);
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        const double cs = mfma4(Rl[I], QRt, 0.0);
        Ch[I] = cJ[I] ? Ch[I] : cs;
      }
    }
  };
  rebuild_C();
  double H1 = tvH ? 0.0 : A.H[b];
  const double d1 = (A.d && !tvD) ? A.d[b] : 0.0;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  double* ux_w = (g == 0) ? UX + fb : smem + B4_OFF_DUMP + 32 + (lane & 31);   // (rows 1 .. 3: dump slots)
  const double* ux_r = (g == 0) ? UX + 4 * q : UX + 16;
  const bool vec_on = g == 3 && bq < m;   // row JC of the tile = 2 a'

  double* t_fa = A.filt_a ? A.filt_a + b * (long)n * m : nullptr;
  double* t_pa = A.pred_a ? A.pred_a + b * (long)(n + 1) * m : nullptr;
  double* t_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* t_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;
  const bool w_fa = t_fa != nullptr, w_pa = t_pa != nullptr, w_fP = t_fP != nullptr, w_pP = t_pP != nullptr;
  TapeStream<RING_A, GRAN_A> s_fa, s_pa;
  s_fa.init(t_fa, smem + B4_OFF_RA);
  s_pa.init(t_pa, smem + B4_OFF_RA + RING_A);
  const int ka = vec_on ? bq : m + (lane & 31);
  unsigned o_fa = s_fa.slot_abs(ka), o_pa = s_pa.slot_abs(ka);
  auto put_a = [&](TapeStream<RING_A, GRAN_A>& st, unsigned& off, double twice_a) {
    lds_store_f64(off, twice_a);
    off = TapeStream<RING_A, GRAN_A>::next_abs(off, m);
    st.appended(m, lane, 0.5);
  };
  TapeStream<RING_P, GRAN_P> s_fP, s_pP;
  s_fP.init(t_fP, smem + B4_OFF_RP);
  s_pP.init(t_pP, smem + B4_OFF_RP + RING_P);
  unsigned o_fP[NB], o_pP[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int dz = GRAN_P - mm > 0 ? GRAN_P - mm : 1;
    const int kp = on[I] ? (4 * I + g) * m + bq : mm + (lane % dz);
    o_fP[I] = s_fP.slot_abs(kp);
    o_pP[I] = s_pP.slot_abs(kp);
  }
  auto put_P = [&](TapeStream<RING_P, GRAN_P>& st, unsigned (&off)[NB], const double (&Preg)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      lds_store_f64(off[I], Preg[I]);
      off[I] = TapeStream<RING_P, GRAN_P>::next_abs(off[I], mm);
    }
    st.appended(mm, lane, 1.0);
  };
  if (w_pa) put_a(s_pa, o_pa, P[NB - 1]);
  if (w_pP) {
    double P0g[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) P0g[I] = on[I] ? gP[(4 * I + g) * m + bq] : 0.0;   // slot 0 = the caller's P0 as given
    put_P(s_pP, o_pP, P0g);
  }
  auto transpose_tile = [&](const double (&X)[NB], double (&Xt)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = X[I];
    wave_lds_fence();
    const double* src = TR + bq * TRS + 4 * g;
    const double2_t t01 = *reinterpret_cast<const double2_t*>(src);
    double2_t t23 = {0.0, 0.0};
    if (NB > 2) t23 = *reinterpret_cast<const double2_t*>(src + 2);
    wave_lds_fence();
    Xt[0] = t01.x;
    if (NB > 1) Xt[NB > 1 ? 1 : 0] = t01.y;
    if (NB > 2) Xt[NB > 2 ? 2 : 0] = t23.x;
  };
  double logp = 0.0;
  int info = 0;
  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
    const int ls = lane & (LB - 1);
    const double2_t fq = *reinterpret_cast<const double2_t*>(RG + 2 * ls);
    const bool act = lane < cnt;
    const bool observed = act && YB[2 * ls + 1] != 0.0;
    const double Fv = fq.x;
    const bool skipped = observed && univariate && Fv == 0.0;      // :468-471
    const bool notpd = observed && !skipped && !(Fv > 0.0);
    const unsigned long long bad = __ballot(notpd);
    if (bad != 0ull && info == 0) info = ((tb + __ffsll((long long)bad)) << 8) | 1;
    const double quad = 0.25 * fq.y * fq.y;                          // v^2 / F
    double llv;
    if (!observed) llv = 0.0;
    else if (skipped) llv = exact_const ? -0.5 * KF_LOG_2PI : 0.0;
    else if (univariate && !exact_const && Fv == 1.0 && quad == 0.0) llv = 0.0;  // :497-500
    else llv = -0.5 * (KF_LOG_2PI + log(Fv) + quad);
    if (act && A.ll_obs) st_stream(A.ll_obs + b * (long)n + tb + lane, llv);
    logp += col_allsum(row_allsum(llv));
    __syncthreads();
  };
  const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
  auto load_y_block = [&](int t) {
    const int cnt = (n - t) < LB ? (n - t) : LB;
    if (lane < cnt) {
      const double yv = ybase[(long)t + lane];
      const double dv = tvD ? A.d[b * (long)n + t + lane] : d1;
      const bool ob = !(yv != yv);
      double2_t e = {ob ? -2.0 * (yv + dsign * dv) : 0.0, ob ? 1.0 : 0.0};
      *reinterpret_cast<double2_t*>(YB + 2 * lane) = e;
      if (tvH) HB[lane] = A.H[b * (long)n + t + lane];
    }
    wave_lds_fence();
  };

  // T_{t+2} is pulled towards the L2 one step before its loads are issued: one 8-byte load per 64-byte line by the first lanes, its
  // value consumed a whole step later (so nothing ever waits for it): the real loads of T_{t+1}, issued late in step t and needed
  // early in step t + 1, then find their lines on the way instead of starting a full HBM round trip
  // (the loads used to be kept alive by folding their sum into a never-true comparison that could zero logp; now
  // the sum ends in an empty `asm volatile` -- the compiler must keep the loads, and logp is written unconditionally)
  double pf_prev = 0.0, pf_sum = 0.0;
  const bool pf_lane = tvT && lane * 8 < mm;
  for (int t = 0; t < n; ++t) {
    if ((t & (LB - 1)) == 0) load_y_block(t);
    const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (t & (LB - 1)));  // {-2 (y -+ d_t), observed}
    if (tvH) H1 = HB[t & (LB - 1)];
    if (TVM & KF_FLAG_TV_T) {
      pf_sum += pf_prev;
      const long t2 = (t + 2 < n) ? t + 2 : t;
      pf_prev = pf_lane ? A.Tm[(b * (long)n + t2) * mm + lane * 8] : 0.0;
    }
    // ---- update: u = S z_t (z' S in DPP row 0), F = H_t + z' S z ------------------------------------------------------------------
    double rrow = mfma4(zA[0], P[0], 0.0);
#pragma unroll
    for (int K = 1; K < NB; ++K) rrow = mfma4(zA[K], P[K], rrow);
    const double uA = readlane_f64(row_allsum(rrow * zc), 0);
    const long tn = (t + 1 < n) ? t + 1 : t;        // (the slices of the next step: loaded behind their last use in this one)
    if (tvZ) load_z(A.Z + (b * (long)n + tn) * m, zc);
    const double u = rrow + ((lane == JC) ? yo.x : 0.0);  // u_JC = 2 z a - 2 (y -+ d) = -2 v   (rows 1 .. 3 of rrow are exact zeros)
    *ux_w = u;
    wave_lds_fence();
    const double2_t ur01 = *reinterpret_cast<const double2_t*>(ux_r);
    double2_t ur23 = {0.0, 0.0};
    if (NB > 2) ur23 = *reinterpret_cast<const double2_t*>(ux_r + 2);
    wave_lds_fence();
    const double urep[4] = {ur01.x, ur01.y, ur23.x, ur23.y};
    const double F = H1 + uA;
    const double rs = fast_rsqrt(fmax(F, 1e-300)) * yo.y;
    const double uh = u * rs;
    if (lane == JC) {
      double2_t fq = {F, uh};
      *reinterpret_cast<double2_t*>(RG + 2 * (t & (LB - 1))) = fq;
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) P[I] = mfma4_nega(urep[I] * rs, uh, P[I]);
    if (w_fa) put_a(s_fa, o_fa, P[NB - 1]);
    if (w_fP) put_P(s_fP, o_fP, P);
    // ---- predict: tile <- sym(T_t tile T_t' + C_t) -----------------------------------------------------------------------------------
    double U[NB], Y[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][0], P[0], 0.0);
#pragma unroll
    for (int K = 1; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][K], P[K], U[I]);
    transpose_tile(U, U);
#pragma unroll
    for (int I = 0; I < NB; ++I) Y[I] = Ch[I];
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I) Y[I] = mfma4(Ta[I][K], U[K], Y[I]);
    if (tvT) load_T(A.Tm + (b * (long)n + tn) * mm);
    if (tvC) {
      const double* ct = A.c + (b * (long)n + tn) * m;
#pragma unroll
      for (int I = 0; I < NB; ++I) Ch[I] = cJ[I] ? 2.0 * ct[cidx[I]] : Ch[I];
    }
    if constexpr (RQM) {   // the next step's R / Q slices and C_{t+1} (behind the last use of C_t, the products above)
      // (a static R / Q is re-read too -- from L1 / L2 -- rather than held across the step: five more live doubles spill)
      load_R(A.R + (tvR ? b * (long)n + tn : b) * (long)m * r_);
      load_Q(A.Q + (tvQ ? b * (long)n + tn : b) * (long)r_ * r_);
      rebuild_C();
    }
    {
      double Yt[NB];
      transpose_tile(Y, Yt);
#pragma unroll
      for (int I = 0; I < NB; ++I) P[I] = 0.5 * (Y[I] + Yt[I]);
    }
    if (w_pa) put_a(s_pa, o_pa, P[NB - 1]);
    if (w_pP) put_P(s_pP, o_pP, P);
    if ((t & (LB - 1)) == LB - 1) flush_ring(t - (LB - 1), LB);
  }
  s_fa.finish(lane, 0.5);
  s_pa.finish(lane, 0.5);
  s_fP.finish(lane, 1.0);
  s_pP.finish(lane, 1.0);
  if (n & (LB - 1)) flush_ring(n & ~(LB - 1), n & (LB - 1));
  if (lane == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
  pf_sum += pf_prev;
  asm volatile("" ::"v"(pf_sum));
}

}  // namespace

// Host-side guard of the kernels above: their tape rings are addressed by ABSOLUTE LDS address and need the dynamic
// LDS block on a multiple of 8 RING_P bytes, i.e. at 0 -- which holds exactly while the kernel has no static __shared__ of its own
// (an inlined helper with one, a debug build or a toolchain change would move it).  Checked once per kernel function.
#include <mutex>
#include <unordered_map>
static int b4_static_lds_free(kf_handle* h, const void* kernel) {
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> seen;
  size_t stat = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find(kernel);
    if (it == seen.end()) {
      hipFuncAttributes attr;
      KF_HIP_CHECK(h, hipFuncGetAttributes(&attr, kernel));
      it = seen.emplace(kernel, attr.sharedSizeBytes).first;
    }
    stat = it->second;
  }
  if (stat != 0) {
    KF_SET_ERR(h, "forward(b4): this build's kernel carries %zu bytes of static LDS; its tape rings need the dynamic block at address 0", stat);
    return KF_ERR_UNSUPPORTED;
  }
  return KF_OK;
}
#define KF_B4_GO(KERNEL_, ARGS_)                                                                    \
  do {                                                                                              \
    auto kfn_ = KERNEL_;                                                                            \
    const int rc_ = b4_static_lds_free(h, reinterpret_cast<const void*>(kfn_));                     \
    if (rc_) return rc_;                                                                            \
    hipLaunchKernelGGL(kfn_, dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, ARGS_);               \
  } while (0)

bool kf_forward_b4_ok(int m, int p, int r) { return p == 1 && m >= 1 && m + 2 <= 16 && r >= 1 && r <= 64; }

// split pipeline (kf_forward_b4_steady.hip): eligible shapes.  KF_B4_SPLIT=0 / 1 in the environment forces the choice.
bool kf_forward_b4_split_ok(const kf_handle* h, long B, int T, int p, int m, int r, unsigned flags) {
  (void)h;
  if (!kf_forward_b4_ok(m, p, r) || m > 10) return false;        // (m <= 10: covariance tapes through the rings)
  if (flags & (KF_FLAG_TV_ANY | KF_FLAG_EXACT_RECURSION)) return false;
  const char* sse = kf_env_switch("KF_STEADY_STATE");
  if (sse && *sse && atoi(sse) == 0) return false;
  const char* e = kf_env_switch("KF_B4_SPLIT");
  if (e && *e) return atoi(e) != 0 && T >= 64;
  if ((flags & KF_FLAG_STEADY_STATE) || (sse && *sse && atoi(sse) != 0)) return T >= 256;
  return T >= KF_B4_STEADY_MIN_T && B >= 1;
}

static int b4_split_chunk(int n) {
  // a quarter of the series per recursion launch (a launch boundary costs ~40 us: drain, compaction, hand-over loads), at least
  // 128 steps; measured (scripts/r03/split_check.py c3tune, 8192 x 2000): 128 / 256 / 384 / 512 / 768 steps 7.2 / 6.0 / 5.8 / 5.7 / 6.0 ms
  int ch = ((n / 4 + 31) & ~31);
  if (ch < 128) ch = 128;
  const char* e = kf_env_experiment("KF_B4_SPLIT_CHUNK");
  if (e && *e && atoi(e) >= 32) ch = (atoi(e) + 31) & ~31;
  while ((n + ch - 1) / ch > B4_SPLIT_MAX_CHUNKS) ch += 32;
  return ch;
}

// the chunk instantiation of the block kernel (SS == 3), B x nc wavefronts
static int b4_launch_chunks(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  const int lds_setup = b4_setup_doubles(a.m, a.r);
  const size_t lds = sizeof(double) * (size_t)(B4_LOOP_DOUBLES > lds_setup ? B4_LOOP_DOUBLES : lds_setup);
  const int nb = (a.m + 2 + 3) / 4;
  const int mask = (a.filt_a ? 1 : 0) | (a.pred_a ? 2 : 0) | (a.filt_P ? 4 : 0) | (a.pred_P ? 8 : 0);
  const unsigned grid = (unsigned)(a.B * a.ck_nc);
#define KF_B4C_GO(NB_, TM_)                                                                          \
  do {                                                                                               \
    auto kfn_ = kf_fwd_b4_kernel<NB_, true, TM_, 3>;                                                 \
    const int rc_ = b4_static_lds_free(h, reinterpret_cast<const void*>(kfn_));                      \
    if (rc_) return rc_;                                                                             \
    hipLaunchKernelGGL(kfn_, dim3(grid), dim3(KF_WAVE), lds, stream, a);                             \
  } while (0)
#define KF_B4C_LAUNCH(NB_)                        \
  do {                                            \
    if (mask == 12) KF_B4C_GO(NB_, 12);           \
    else if (mask == 0) KF_B4C_GO(NB_, 0);        \
    else KF_B4C_GO(NB_, -1);                      \
  } while (0)
  switch (nb) {
    case 1: KF_B4C_LAUNCH(1); break;
    case 2: KF_B4C_LAUNCH(2); break;
    default: KF_B4C_LAUNCH(3); break;
  }
#undef KF_B4C_LAUNCH
#undef KF_B4C_GO
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}

int kf_launch_forward_b4(kf_handle* h, const KfFwdArgs& a_in, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!kf_forward_b4_ok(a_in.m, a_in.p, a_in.r) || (a_in.flags & KF_FLAG_TV_ANY)) {
    KF_SET_ERR(h, "forward(b4): needs p == 1, m <= 14, static matrices (got p=%d m=%d r=%d)", a_in.p, a_in.m, a_in.r);
    return KF_ERR_UNSUPPORTED;
  }
  KfFwdArgs a = a_in;
  // Round 6: small batches through the time-chunked pipeline (kf_forward_b4c.hip) -- doubling for the chunk-start covariances, B x nc
  // chunk wavefronts, the mean pass -- and, behind it, the sequential kernel with run_if = the pipeline's flag (a missing observation,
  // a chunk whose end does not meet its successor's start: rare, and then the whole batch is redone; nothing synchronises the host)
  if (!a.tile_tape && !a.list && !a.hs_ts && !a.run_if && a.t_end == 0 && workspace &&
      kf_forward_b4c_shape_ok(h, a.B, a.T, a.p, a.m, a.r, a.flags)) {
    int cs = 0, cn = 0;
    kf_forward_b4c_plan(h, a.B, a.T, &cs, &cn);
    B4cWs W;
    const size_t need = kf_forward_b4c_carve(nullptr, a.B, a.T, a.m, cn, &W);
    if (cn >= 2 && workspace_bytes >= need) {
      void* wbase = (void*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
      kf_forward_b4c_carve(wbase, a.B, a.T, a.m, cn, &W);
      KfFwdArgs ac = a;
      ac.filt_a = nullptr; ac.pred_a = nullptr;   // the state tapes (and ll_obs, logp, info) are the mean pass's
      ac.ck_nc = cn; ac.ck_s = cs; ac.ck_start = W.start; ac.ck_K = W.K; ac.ck_fv = W.fv; ac.ck_zend = W.zend; ac.ck_flag = W.flag;
      int rc = kf_launch_b4c_starts(h, ac, W, cs, cn, stream);
      if (rc) return rc;
      rc = b4_launch_chunks(h, ac, stream);
      if (rc) return rc;
      ac.filt_a = a.filt_a; ac.pred_a = a.pred_a;
      rc = kf_launch_b4c_means(h, ac, W, cs, cn, stream);
      if (rc) return rc;
      a.run_if = W.flag;   // the sequential kernel below: a no-op unless the pipeline raised its flag
    }
  }
  // Small batches with covariance tapes: the pair kernel (kf_forward_b4p.hip) from step 0.  With at most one draw per SIMD the
  // step time is one wavefront's dependency chain, and a writer wavefront takes the tape path off it (measured, 1000 steps:
  // 0.59 / 0.63 / 0.65 ms at 256 / 512 / 1024 draws against 0.81 / 0.83 / 0.86; beyond 2048 draws the fused kernel wins).
  // KF_FORWARD_B4P=0 / 1 forces the choice.
  bool pair_env_off = false;
  {
    const char* e = kf_env_switch("KF_FORWARD_B4P");
    // (the pair kernel is the recursion at every step: a caller who asked for the steady-state phase -- the flag, or KF_STEADY_STATE=1
    // in the environment, which include/kf.h documents as the same request -- gets the kernels that have one)
    const char* sreq = kf_env_switch("KF_STEADY_STATE");
    const bool steady_requested = !(a.flags & KF_FLAG_EXACT_RECURSION) && ((a.flags & KF_FLAG_STEADY_STATE) || (sreq && atoi(sreq) != 0));
    bool pair = a.B <= B4_PAIR_MAX && (a.filt_P || a.pred_P) && !steady_requested && !a.run_if;
    if (e && *e) {
      pair = atoi(e) != 0;
      pair_env_off = !pair;
    }
    if (pair && !a.tile_tape && !a.run_if && kf_forward_b4p_ok(a.m, a.p, a.r)) return kf_launch_forward_b4p(h, a, stream);
  }
  const int lds_setup = b4_setup_doubles(a.m, a.r);
  const size_t lds = sizeof(double) * (size_t)(B4_LOOP_DOUBLES > lds_setup ? B4_LOOP_DOUBLES : lds_setup);
  const int nb = (a.m + 2 + 3) / 4;
  // kf_filter_value_and_grad with the steady-state adjoint behind it ("VGS", kernel header): the tile tape for the recursion's steps,
  // a_t in the compact predicted-state tape from the convergence point on -- pred_a is then NOT a tape request (mask stays 16)
  const bool vgs = a.tile_tape && a.hs_ts && a.vg_P && a.pred_a && !a.filt_a && !a.filt_P && !a.pred_P && a.T >= 64 &&
                   !(a.flags & KF_FLAG_EXACT_RECURSION);
  const int mask = (a.filt_a ? 1 : 0) | ((a.pred_a && !vgs) ? 2 : 0) | (a.filt_P ? 4 : 0) | (a.pred_P ? 8 : 0) | (a.tile_tape ? 16 : 0);
  // steady-state phase (kernel header): a second instantiation, so that KF_FLAG_EXACT_RECURSION runs the very kernel it ran before
  // (KF_STEADY_STATE=0 in the environment: the same switch for A/B measurements)
  // Steady-state phase: only for LONG series (T >= KF_B4_STEADY_MIN_T).  At the north-star length (T = 1000) the recursion at every
  // step is the faster kernel on most boxes (round 3, six boxes: 1.65 - 1.73 ms against 1.78 - 1.84 with the in-kernel phase and
  // 2.0 - 2.4 with the split pipeline) -- a launch lasts as long as its slowest SIMD, and 5 % of the structural draws never
  // converge; at T = 2000 the split pipeline wins (C3: 5.7 against 6.8 - 7.0 ms).  KF_STEADY_STATE=1 / 0 forces it on / off.
  const char* sse = kf_env_switch("KF_STEADY_STATE");
  const bool ss_env_on = (sse && *sse && atoi(sse) != 0) || (a.flags & KF_FLAG_STEADY_STATE), ss_env_off = sse && *sse && atoi(sse) == 0;
  const bool ss = !(a.flags & KF_FLAG_EXACT_RECURSION) && !a.tile_tape && a.T >= 64 && !ss_env_off &&
                  (a.T >= KF_B4_STEADY_MIN_T || ss_env_on || kf_env_switch("KF_B4_SPLIT") != nullptr);
  // (tried: capping the residency of this instantiation at 3 / 2 wavefronts per SIMD so that the dispatcher hands queued draws to the
  // SIMDs whose draws converged early -- headline 2.29 / 2.44 ms against 1.63: the queued wavefronts run as a second round)
  // split pipeline: recursion chunks over compacted draw lists + the time-parallel steady kernels on the second stream
  const bool split = ss && nb <= 3 && workspace && !a.run_if && kf_forward_b4_split_ok(h, a.B, a.T, a.p, a.m, a.r, a.flags) &&
                     workspace_bytes >= kf_b4_split_workspace_bytes(a.B, a.m, a.T);
  KfFwdArgs ac = a;  // (the chunk's arguments in split mode)
#define KF_B4_LAUNCH2(NB_, ST_, TM_)                                                                                       \
  do {                                                                                                                     \
    if (split && ST_ && TM_ != 16)                                                                                         \
      KF_B4_GO((kf_fwd_b4_kernel<NB_, ST_, TM_, (ST_ && TM_ != 16) ? 2 : 0>), ac); \
    else if (ss && ST_ && TM_ != 16)                                                                                       \
      KF_B4_GO((kf_fwd_b4_kernel<NB_, ST_, TM_, (ST_ && TM_ != 16) ? 1 : 0>), a); \
    else                                                                                                                   \
      KF_B4_GO((kf_fwd_b4_kernel<NB_, ST_, TM_, 0>), a);        \
  } while (0)
#define KF_B4_LAUNCH(NB_, ST_)                           \
  do {                                                   \
    if (mask == 15) KF_B4_LAUNCH2(NB_, ST_, 15);         \
    else if (mask == 0) KF_B4_LAUNCH2(NB_, ST_, 0);      \
    else if (mask == 10) KF_B4_LAUNCH2(NB_, ST_, 10);    \
    else if (mask == 16 && vgs && ST_)                   \
      KF_B4_GO((kf_fwd_b4_kernel<NB_, ST_, 16, ST_ ? 1 : 0>), a); \
    else if (mask == 16) KF_B4_LAUNCH2(NB_, ST_, 16);    \
    else KF_B4_LAUNCH2(NB_, ST_, -1);                    \
  } while (0)
  if (split) {
    const int rc0 = kf_handle_aux(h);
    if (rc0) return rc0;
    B4SplitWs W;
    void* wbase = (void*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    kf_b4_split_carve(wbase, a.B, a.m, a.T, &W);
    KfAuxJoinGuard join_guard(h, stream);
    ac.hs_const = W.hs_const; ac.hs_tile = W.hs_tile; ac.hs_logp = W.hs_logp; ac.hs_info = W.hs_info; ac.hs_ts = W.hs_ts; ac.hs_tfree = W.hs_tfree;
    const int ch = b4_split_chunk(a.T), nch = (a.T + ch - 1) / ch;
    // the second stream starts behind everything already queued on the caller's stream (its inputs) ...
    KF_HIP_CHECK(h, hipEventRecord(h->aux_ev[33], stream));
    KF_HIP_CHECK(h, hipStreamWaitEvent(h->aux_stream, h->aux_ev[33], 0));
    KF_HIP_CHECK(h, hipStreamWaitEvent(h->aux_stream2, h->aux_ev[33], 0));
    KF_HIP_CHECK(h, hipStreamWaitEvent(h->aux_stream3, h->aux_ev[33], 0));
    const bool pair_ok = !pair_env_off && (a.filt_P || a.pred_P) && kf_forward_b4p_ok(a.m, a.p, a.r);
    for (int c = 0; c < nch; ++c) {
      ac.t_begin = c * ch;
      ac.t_end = (c + 1) * ch < a.T ? (c + 1) * ch : a.T;
      ac.list = c ? W.listR : nullptr;
      ac.count = c ? W.cnt : nullptr;
      switch (nb) {
        case 1: KF_B4_LAUNCH(1, true); break;
        case 2: KF_B4_LAUNCH(2, true); break;
        default: KF_B4_LAUNCH(3, true); break;
      }
      KF_HIP_CHECK(h, hipGetLastError());
      const bool more = ac.t_end < a.T;
      int rc = kf_b4_split_compact(h, W, a.B, c, ac.t_end, pair_ok && more, stream);
      if (rc) return rc;
      KF_HIP_CHECK(h, hipEventRecord(h->aux_ev[c], stream));
      KF_HIP_CHECK(h, hipStreamWaitEvent(h->aux_stream, h->aux_ev[c], 0));
      KF_HIP_CHECK(h, hipStreamWaitEvent(h->aux_stream2, h->aux_ev[c], 0));
      rc = kf_b4_split_steady(h, a, W, c, h->aux_stream, h->aux_stream2);
      if (rc) return rc;
      if (pair_ok && more && c > 0) {
        // the pair kernel's launch for THIS hand-over point: runs only if this compaction decided the switch (pair_ctl[2] == c + 1)
        KF_HIP_CHECK(h, hipStreamWaitEvent(h->aux_stream3, h->aux_ev[c], 0));
        KfFwdArgs ap = a;
        ap.list = W.listR; ap.count = W.pair_ctl; ap.t_begin = c + 1;
        ap.hs_tile = W.hs_tile; ap.hs_logp = W.hs_logp; ap.hs_info = W.hs_info; ap.hs_ts = W.hs_ts;
        rc = kf_launch_forward_b4p(h, ap, h->aux_stream3);
        if (rc) return rc;
      }
    }
    KF_HIP_CHECK(h, hipEventRecord(h->aux_ev[30], h->aux_stream3));
    KF_HIP_CHECK(h, hipStreamWaitEvent(stream, h->aux_ev[30], 0));
    {  // the recursion is done: the rest of the constant covariance tapes at full width (shares the queues of the throttled launches)
      const int rc = kf_b4_split_drain(h, a, W, nch - 1, stream);
      if (rc) return rc;
    }
    // ... and the caller's stream continues behind the last steady kernel
    KF_HIP_CHECK(h, hipEventRecord(h->aux_ev[32], h->aux_stream));
    KF_HIP_CHECK(h, hipStreamWaitEvent(stream, h->aux_ev[32], 0));
    KF_HIP_CHECK(h, hipEventRecord(h->aux_ev[31], h->aux_stream2));
    KF_HIP_CHECK(h, hipStreamWaitEvent(stream, h->aux_ev[31], 0));
    join_guard.dismiss();
    return KF_OK;
  }
  switch (nb) {
    case 1: KF_B4_LAUNCH(1, true); break;
    case 2: KF_B4_LAUNCH(2, true); break;
    case 3: KF_B4_LAUNCH(3, true); break;
    default: KF_B4_LAUNCH(4, false); break;  // m = 11 would fit the rings, but measured 2.48 vs 2.40 ms
  }
#undef KF_B4_LAUNCH2
#undef KF_B4_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}


// time-varying T / Z / H / c / d on the block tiles: p == 1, m <= 10 (the tape rings); R / Q too when r <= 4 (round 5)
bool kf_forward_b4tv_ok(int m, int p, int r, unsigned flags) {
  if (!(flags & KF_FLAG_TV_ANY)) return false;
  if ((flags & (KF_FLAG_TV_R | KF_FLAG_TV_Q)) && r > 4) return false;   // (round 5: R_t / Q_t slices as single 4 x 4 block factors)
  return p == 1 && m >= 1 && m <= 10 && r >= 1 && r <= 64;
}

int kf_launch_forward_b4tv(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  if (!kf_forward_b4tv_ok(a.m, a.p, a.r, a.flags) || a.tile_tape) {
    KF_SET_ERR(h, "forward(b4tv): needs p == 1, m <= 10 (time-varying R / Q: r <= 4) (got p=%d m=%d r=%d flags=%#x)", a.p, a.m, a.r, a.flags);
    return KF_ERR_UNSUPPORTED;
  }
  const size_t lds = sizeof(double) * (size_t)B4_LOOP_DOUBLES;
  const int nb = (a.m + 2 + 3) / 4;
  constexpr unsigned M_T = KF_FLAG_TV_T, M_TZ = KF_FLAG_TV_T | KF_FLAG_TV_Z,
                     M_ALL = KF_FLAG_TV_T | KF_FLAG_TV_Z | KF_FLAG_TV_H | KF_FLAG_TV_C | KF_FLAG_TV_D,
                     M_RQ = KF_FLAG_TV_R | KF_FLAG_TV_Q, M_TRQ = M_T | M_RQ, M_ALLRQ = M_ALL | M_RQ;
  const unsigned tvf = a.flags & KF_FLAG_TV_ANY;
#define KF_TV_LAUNCH(NB_)                                                                                                   \
  do {                                                                                                                      \
    if (!(tvf & ~M_T)) KF_B4_GO((kf_fwd_b4tv_kernel<NB_, M_T>), a);          \
    else if (!(tvf & ~M_TZ)) KF_B4_GO((kf_fwd_b4tv_kernel<NB_, M_TZ>), a);   \
    else if (!(tvf & ~M_ALL)) KF_B4_GO((kf_fwd_b4tv_kernel<NB_, M_ALL>), a);  \
    else if (!(tvf & ~M_RQ)) KF_B4_GO((kf_fwd_b4tv_kernel<NB_, M_RQ>), a);    \
    else if (!(tvf & ~M_TRQ)) KF_B4_GO((kf_fwd_b4tv_kernel<NB_, M_TRQ>), a);  \
    else KF_B4_GO((kf_fwd_b4tv_kernel<NB_, M_ALLRQ>), a);                     \
  } while (0)
  switch (nb) {
    case 1: KF_TV_LAUNCH(1); break;
    case 2: KF_TV_LAUNCH(2); break;
    default: KF_TV_LAUNCH(3); break;
  }
#undef KF_TV_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
