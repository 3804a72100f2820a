// kf_forward_b4p.hip -- the 4x4x4-block forward filter as a PAIR of wavefronts per draw: one runs the filter
// recursion (the logp-only loop of kf_forward_b4.hip and nothing else), the other turns what the first leaves in LDS
// into the four output tapes.
//
// Where it is used (round 3).  Built in round 2 for the full batch (eight wavefronts per SIMD, 64 registers each) it lost
// to the fused kernel there (2.25 against 1.84 ms at 4096 draws).  It wins where a SIMD holds at most ONE draw: then the
// step time is the dependency chain of one wavefront, and the tape path is half of that chain (measured, 1024 draws x 1000
// steps: fused 0.86 ms, log-likelihood only 0.42 ms, this kernel 0.65 ms) -- the writer runs on the SIMD's idle issue slots.
//   * batches of <= 1024 draws (one of eight GPUs' share of the north-star batch: 512) run on it from step 0;
//   * the split pipeline (kf_forward_b4.hip / kf_forward_b4_steady.hip) hands it the draws that are still in the full
//     recursion once <= 1024 of them are left: it continues from the hand-over tile to the end of the series.
//
// Why: at the north-star shape a SIMD holds four draws.  With the tapes written by the wavefront that runs the
// recursion, every LDS round trip of the tape path and, worse, the time a store waits to ISSUE while the CU's memory
// pipeline is busy, sit in the one dependency chain the step time is made of: 1.05 ms of arithmetic + 0.93 ms of
// stores take 1.70 - 1.85 ms (kf_forward_b4.hip; scripts/probes/overlap_probe.hip reproduces it without a filter).
// Here the recursion never executes a global store (but ll_obs, 32 steps at a time).  Per step it publishes what it
// writes to LDS anyway -- the pass-2 transpose tile Y = (T_ext P_f T_ext' + C_ext) / 2, the update vector u, and
// 1 / sqrt F -- into one of NSLOT step slots and bumps a counter.  The storer reconstructs, in the COMPACT layout of
// the tapes (lane L owns elements 2L, 2L+1 of the m x m matrix):
//     predicted[t+1] = Y + Y'            filtered[t] = predicted[t] - (u rs)(u rs)'        states: row JC / 2
// and streams them out with 16 bytes per lane.  A store that blocks stalls only the storer; the recursion waits only if
// the storer is NSLOT - 1 steps behind.  (A first attempt with ONE storer per FOUR draws -- five wavefronts per SIMD, 96
// registers -- was correct and 2.3x slower: scripts/r02/attic.)
//
// Register budget: 64 per wavefront.  The recursion keeps one set of replicated T blocks (the 1/2 of the
// symmetrisation is applied to Y), and C_ext lives in LDS (it is the C operand of three MFMAs per step).
#include "kf_b4.h"
#include "kf_b4_split.h"

namespace {

constexpr int P_LB = 32;  // steps per y block / likelihood ring
constexpr int P_THREADS = 2 * KF_WAVE;

// LDS per draw (doubles): pass-1 tile | C_ext (NB x 64) | NSLOT x { Y tile (N4 rows) | u by column, permuted (16) |
//                         zeros (4) | rs (4) } | (4 zeros) | likelihood ring | y block | ... | flags {produced, consumed} at the end
template <int NB>
struct PLay {
  static constexpr int N4 = 4 * NB;
  static constexpr int NSLOT = NB <= 3 ? 3 : 2;
  static constexpr int TILE = N4 * TRS;
  static constexpr int SLOT = TILE + 16 + 4 + 4;
  static constexpr int OFF_CH = N4 * TRS;           // (b4_setup's scratch overlays everything; it is done before CH is written)
  static constexpr int OFF_Z4 = OFF_CH + NB * 64;   // four zeros: what the lanes of the unused block slot read back
  static constexpr int OFF_SLOT = OFF_Z4 + 4;
  static constexpr int OFF_RG = OFF_SLOT + NSLOT * SLOT;
  static constexpr int OFF_YB = OFF_RG + 2 * P_LB;
  static constexpr int DRAW = OFF_YB + 2 * P_LB;
};
template <int NB>
__host__ __device__ inline int p_draw_doubles(int m, int r) {
  // the flags sit BEHIND whatever b4_setup uses as scratch (it runs while the storer already polls them)
  const int a = PLay<NB>::DRAW, b = 2 * m * m + m * r + r * r + 5 * m + 32 * r + 4;
  return (((a > b ? a : b) + 1) & ~1) + 2;
}

// the flags are LDS words: say so in the type, or the accesses become FLAT ones that wait on vmcnt (all the storer's
// outstanding tape stores) every step
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ int lds_flag(const int* p) {
  return __builtin_amdgcn_readfirstlane(*(const volatile lds_int_t*)p);
}
__device__ __forceinline__ void lds_flag_set(int* p, int v) { *(volatile lds_int_t*)p = v; }

#ifndef B4P_WPE
#define B4P_WPE 4
#endif
template <int NB>
__global__ __launch_bounds__(P_THREADS) __attribute__((amdgpu_waves_per_eu(B4P_WPE))) void kf_fwd_b4p_kernel(const KfFwdArgs A) {
  using L = PLay<NB>;
  constexpr int N4 = 4 * NB, JA = N4 - 2, JC = N4 - 1, NSLOT = L::NSLOT;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = A.T, m = A.m, mm = m * m;
  const int stride = p_draw_doubles<NB>(m, A.r);
  // continuation of the split pipeline: the draws pair_list[0 .. pair_ctl[0]) from step pair_ctl[1], when pair_ctl[2] names this launch
  long b = blockIdx.x;
  int t0 = 0;
  if (A.list) {
    if (A.count[2] != A.t_begin || (int)blockIdx.x >= A.count[0]) return;  // (t_begin: this launch's index)
    b = A.list[blockIdx.x];
    t0 = A.count[1];
  }
  int* fl = reinterpret_cast<int*>(smem + stride - 2);

  if (threadIdx.x < 2) fl[threadIdx.x] = t0;
  __syncthreads();  // the only workgroup barrier: the roles never meet again

  if (wave == 0) {
    // ================================== recursion ===================================================================
    const int g = lane >> 4, bq = lane & 15;
    const int fb = (lane & 3) * 4 + ((lane >> 2) & 3);
    double* TR = smem;
    double* CH = smem + L::OFF_CH;
    double* RG = smem + L::OFF_RG;
    double* YB = smem + L::OFF_YB;

    // (round 3: the kernel runs at <= 4 wavefronts per SIMD now -- 128 registers: T / 2 and C_ext / 2 live in registers as in
    // kf_forward_b4.hip; with 64 registers they were a multiply per product and three LDS reads on the step's dependency chain)
    double Ta[NB][NB], Tha[NB][NB], Chr[NB], P[NB];
    {
      B4Const<NB, false> KC;
      b4_setup(A, b, smem, lane, KC);
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        P[I] = KC.P[I];
        Chr[I] = 0.5 * KC.Ch[I];  // C_ext / 2: the C operand of the second product
        CH[I * 64 + lane] = Chr[I];
#pragma unroll
        for (int K = 0; K < NB; ++K) {
          Ta[I][K] = KC.Ta[I][K];
          double th = 0.5 * KC.Ta[I][K];
          asm volatile("" : "+v"(th));  // keep T / 2 in registers (the compiler otherwise re-multiplies per step)
          Tha[I][K] = th;
        }
      }
    }
    double logp = 0.0;
    int info = 0;
    if (t0 > 0) {  // the tile, log-likelihood and flag the last chunk of the fused kernel left
      const double* ht = A.hs_tile + b * (long)B4_HS_STRIDE + lane;
#pragma unroll
      for (int I = 0; I < NB; ++I) P[I] = ht[I * KF_WAVE];
      logp = A.hs_logp[b];
      info = A.hs_info[b];
    }
    for (int e = lane; e < 4 + NSLOT * L::SLOT; e += KF_WAVE) smem[L::OFF_Z4 + e] = 0.0;
    wave_lds_fence();

    double H1 = A.H[b];
    asm volatile("" : "+v"(H1));
    const double d1 = A.d ? A.d[b] : 0.0;
    const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
    const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
    const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
    // (round 4: u stays in DPP row 2, where row JA of the tile is -- the rank-one MFMA contracts over k = 2; the two v_permlane32_swap
    // that brought it to row 0 sat on the step's dependency chain, which is all this kernel's step time is made of)
    const bool row2 = g == 2 && bq < N4;
    // lanes of the unused block slot (bq >= N4: NB = 3) read zeros back, so that their registers stay zero
    const bool col_ok = bq < N4;
    const int tr_r = bq * TRS + 4 * g;
    const int tr_w = g * TRS + fb;
    const double* zero4 = smem + L::OFF_Z4;

    auto transpose_via = [&](double* tile, const double (&X)[NB], double (&Xt)[NB]) {
#pragma unroll
      for (int I = 0; I < NB; ++I) tile[4 * I * TRS + tr_w] = X[I];
      wave_lds_fence();
      const double* src = col_ok ? tile + tr_r : zero4;
      const double2_t t01 = *reinterpret_cast<const double2_t*>(src);
      double2_t t23 = {0.0, 0.0};
      if (NB > 2) t23 = *reinterpret_cast<const double2_t*>(src + 2);
      wave_lds_fence();
      Xt[0] = t01.x;
      if (NB > 1) Xt[NB > 1 ? 1 : 0] = t01.y;
      if (NB > 2) Xt[NB > 2 ? 2 : 0] = t23.x;
      if (NB > 3) Xt[NB > 3 ? 3 : 0] = t23.y;
    };

    auto flush_ring = [&](int tb, int cnt) {
      wave_lds_fence();
      const int ls = lane & (P_LB - 1);
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
      wave_lds_fence();
    };

    const double* ybase = A.y + ((A.flags & KF_FLAG_Y_PER_DRAW) ? b * (long)n : 0L);
    int cons_seen = 0;  // what the storer had consumed when we last looked

#ifdef B4P_NOREC
    if (H1 != 1.2345e300) return;
#endif
    for (int t = t0; t < n; ++t) {
      if ((t & (P_LB - 1)) == 0) {
        const int cnt = (n - t) < P_LB ? (n - t) : P_LB;
        if (lane < cnt) {
          const double yv = ybase[(long)t + lane];
          const bool ob = !(yv != yv);
          double2_t e = {ob ? -2.0 * (yv + dsign * d1) : 0.0, ob ? 1.0 : 0.0};
          *reinterpret_cast<double2_t*>(YB + 2 * lane) = e;
        }
        wave_lds_fence();
      }
      // slot t % NSLOT is free once the storer has consumed step t - NSLOT, i.e. consumed >= t - NSLOT + 1
#if defined(B4P_NOWAIT) || defined(B4P_FREE)  // experiment builds: recursion alone
      cons_seen = t;
#endif
      while (cons_seen < t - NSLOT + 1) {
        cons_seen = lds_flag(fl + 1);
        if (cons_seen < t - NSLOT + 1) __builtin_amdgcn_s_sleep(1);
      }
      double* slot = smem + L::OFF_SLOT + (t % NSLOT) * L::SLOT;
      double* UX = slot + L::TILE;
      const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (t & (P_LB - 1)));  // {-2 (y -+ d), observed}
      // ---- update (kf_forward_b4.hip) ----------------------------------------------------------------------------------
      double u = row2 ? P[NB - 1] : 0.0;
      if (lane == 32 + JC) u += yo.x;  // u_JC = 2 Z a - 2 (y -+ d) = -2 v
      if (g == 2) UX[fb] = u;
      wave_lds_fence();
      const double* ux_r = (g == 2) ? UX + 4 * (lane & 3) : UX + 16;
      const double2_t ur01 = *reinterpret_cast<const double2_t*>(ux_r);
      double2_t ur23 = {0.0, 0.0};
      if (NB > 2) ur23 = *reinterpret_cast<const double2_t*>(ux_r + 2);
      wave_lds_fence();
      const double urep[4] = {ur01.x, ur01.y, ur23.x, ur23.y};
      const double F = H1 + readlane_f64(u, 32 + JA);
      const double rs = fast_rsqrt(fmax(F, 1e-300)) * yo.y;
      const double uh = u * rs;
      if (lane == 32 + JC) {
        double2_t fq = {F, uh};
        *reinterpret_cast<double2_t*>(RG + 2 * (t & (P_LB - 1))) = fq;
        UX[20] = rs;
      }
#pragma unroll
      for (int I = 0; I < NB; ++I) P[I] = mfma4_nega(urep[I] * rs, uh, P[I]);
      // ---- predict -----------------------------------------------------------------------------------------------------
      double U[NB], Y[NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) U[I] = 0.0;
#pragma unroll
      for (int K = 0; K < NB; ++K)
#pragma unroll
        for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][K], P[K], U[I]);
      transpose_via(TR, U, U);
#pragma unroll
      for (int I = 0; I < NB; ++I) Y[I] = Chr[I];
#pragma unroll
      for (int K = 0; K < NB; ++K) {
#pragma unroll
        for (int I = 0; I < NB; ++I) Y[I] = mfma4(Tha[I][K], U[K], Y[I]);  // (T / 2) U': exact halves
      }
      {
        double Yt[NB];
        transpose_via(slot, Y, Yt);  // pass 2 through the slot's tile: this IS the published Y
#pragma unroll
        for (int I = 0; I < NB; ++I) P[I] = Y[I] + Yt[I];
      }
      // publish step t: every LDS write above is older than this one in the wavefront's (in-order) LDS stream
      if (lane == 0) lds_flag_set(fl, t + 1);
      if ((t & (P_LB - 1)) == P_LB - 1) flush_ring(t - (P_LB - 1), P_LB);
    }
    if (n & (P_LB - 1)) flush_ring(n & ~(P_LB - 1), n & (P_LB - 1));
    if (lane == 0) {
      A.logp[b] = logp;
      if (A.info) A.info[b] = info;
      if (A.hs_ts) A.hs_ts[b] = -2;
    }
    return;
  }

  // ===================================== storer =======================================================================
  constexpr int NR = NB <= 3 ? 1 : 2;  // rounds of 128 elements: m * m <= 100 (NB <= 3), <= 196
  auto perm = [](int j) { return 4 * (j & 3) + (j >> 2); };
  int oy1[NR][2], oy2[NR][2], our[NR][2], ouc[NR][2];
  bool el[NR][2];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = 128 * r + 2 * lane + k;
      el[r][k] = e < mm;
      const int ee = el[r][k] ? e : 0;
      const int i = ee / m, j = ee - i * m;
      oy1[r][k] = i * TRS + perm(j);
      oy2[r][k] = j * TRS + perm(i);
      our[r][k] = L::TILE + perm(i);
      ouc[r][k] = L::TILE + perm(j);
    }
  const bool two_rounds = NR > 1 && mm > 128;
  const bool av = lane < m;
  const int oa1 = JC * TRS + perm(av ? lane : 0), oa2 = (av ? lane : 0) * TRS + perm(JC);
  const int oua = L::TILE + perm(av ? lane : 0), ouj = L::TILE + perm(JC);

  double2_t pp[NR];  // predicted covariance of the step being filtered, this lane's elements
  double pa = 0.0;   // predicted state (row JC / 2)
  if (t0 > 0) {
    // continuation: the predicted moments of step t0 are the hand-over tile (row-major [N4][16]); slot t0 is on the tapes already
    const double* ht = A.hs_tile + b * (long)B4_HS_STRIDE;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = 128 * r + 2 * lane;
      pp[r] = double2_t{0.0, 0.0};
      if (el[r][0]) pp[r].x = ht[(e / m) * 16 + e % m];
      if (el[r][1]) pp[r].y = ht[((e + 1) / m) * 16 + (e + 1) % m];
    }
    if (av) pa = 0.5 * ht[JC * 16 + lane];
  } else {
    // predicted slot 0: the caller's (a0, P0); the recursion starts from P0's upper triangle
    const double* gP = A.P0 + b * (long)mm;
    auto P0s = [&](int e) {
      const int i = e / m, j = e - i * m;
      return (j >= i) ? gP[e] : gP[j * m + i];
    };
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = 128 * r + 2 * lane;
      pp[r] = double2_t{0.0, 0.0};
      if (el[r][0]) pp[r].x = P0s(e);
      if (el[r][1]) pp[r].y = P0s(e + 1);
    }
    if (av) pa = A.a0[b * (long)m + lane];
    if (A.pred_P) {
      double* o = A.pred_P + b * (long)(n + 1) * mm;
      for (int e = lane; e < mm; e += KF_WAVE) o[e] = gP[e];
    }
    if (A.pred_a && av) A.pred_a[b * (long)(n + 1) * m + lane] = pa;
  }
  auto store2 = [&](double* base, int e, double2_t v) {  // elements e, e + 1 of a compact m x m matrix
    if (e + 1 < mm) *reinterpret_cast<double2_t*>(base + e) = v;
    else if (e < mm) base[e] = v.x;
  };
  double* o_fP = A.filt_P ? A.filt_P + (b * (long)n + t0) * mm : nullptr;
  double* o_pP = A.pred_P ? A.pred_P + (b * (long)(n + 1) + t0 + 1) * mm : nullptr;
  double* o_fa = A.filt_a ? A.filt_a + (b * (long)n + t0) * m : nullptr;
  double* o_pa = A.pred_a ? A.pred_a + (b * (long)(n + 1) + t0 + 1) * m : nullptr;

#ifdef B4P_NOWAIT
  return;
#endif
  int slot_off = L::OFF_SLOT + (t0 % NSLOT) * L::SLOT;
#pragma clang loop unroll(disable)
  for (int t = t0; t < n; ++t) {
#ifndef B4P_FREE
    while (lds_flag(fl) <= t) __builtin_amdgcn_s_sleep(1);
#endif
    const double* slot = smem + slot_off;
    slot_off = (slot_off == L::OFF_SLOT + (NSLOT - 1) * L::SLOT) ? L::OFF_SLOT : slot_off + L::SLOT;
    const double rs = slot[L::TILE + 20];
    double2_t f[NR], q[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (r == 1 && !two_rounds) break;
      const double ur0 = slot[our[r][0]] * rs, uc0 = slot[ouc[r][0]] * rs;
      const double ur1 = slot[our[r][1]] * rs, uc1 = slot[ouc[r][1]] * rs;
      f[r].x = fma(-ur0, uc0, pp[r].x);
      f[r].y = fma(-ur1, uc1, pp[r].y);
      q[r].x = slot[oy1[r][0]] + slot[oy2[r][0]];
      q[r].y = slot[oy1[r][1]] + slot[oy2[r][1]];
      pp[r] = q[r];
    }
    const double uj = slot[ouj] * rs, ua = slot[oua] * rs;
    const double fa = fma(-0.5 * uj, ua, pa);
    const double qa = 0.5 * (slot[oa1] + slot[oa2]);
    pa = qa;
    // the slot is read: hand it back before the stores (they may block; the recursion must not)
    wave_lds_fence();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane == 0) lds_flag_set(fl + 1, t + 1);
#ifdef B4P_NOSTORE
    if (f[0].x == 1.2345e300 && q[0].y == fa + qa) A.logp[b] = 0.0;
    continue;
#endif
    if (o_fP) {
      store2(o_fP, 2 * lane, f[0]);
      if (NR > 1 && two_rounds) store2(o_fP, 128 + 2 * lane, f[NR - 1]);
      o_fP += mm;
    }
    if (o_pP) {
      store2(o_pP, 2 * lane, q[0]);
      if (NR > 1 && two_rounds) store2(o_pP, 128 + 2 * lane, q[NR - 1]);
      o_pP += mm;
    }
    if (o_fa) {
      if (av) o_fa[lane] = fa;
      o_fa += m;
    }
    if (o_pa) {
      if (av) o_pa[lane] = qa;
      o_pa += m;
    }
  }
}

}  // namespace

// sixteen draws per CU (two wavefronts each: all 32 wavefront slots) must fit 160 KiB of LDS
bool kf_forward_b4p_ok(int m, int p, int r) {
  if (!(p == 1 && m >= 1 && m + 2 <= 16 && r >= 1 && r <= 64)) return false;
  const int nb = (m + 2 + 3) / 4;
  const int d = nb == 1 ? p_draw_doubles<1>(m, r) : nb == 2 ? p_draw_doubles<2>(m, r)
              : nb == 3 ? p_draw_doubles<3>(m, r) : p_draw_doubles<4>(m, r);
  return sizeof(double) * (size_t)d * 16 <= KF_LDS_BYTES_MAX;
}

int kf_launch_forward_b4p(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  const unsigned grid = (unsigned)(a.list ? (a.B < B4_PAIR_MAX ? a.B : B4_PAIR_MAX) : a.B);
  if (!kf_forward_b4p_ok(a.m, a.p, a.r) || (a.flags & KF_FLAG_TV_ANY)) {
    KF_SET_ERR(h, "forward(b4p): needs p == 1, m <= 14, static matrices (got p=%d m=%d r=%d)", a.p, a.m, a.r);
    return KF_ERR_UNSUPPORTED;
  }
  const int nb = (a.m + 2 + 3) / 4;
#define KF_B4P_LAUNCH(NB_)                                                                 \
  {                                                                                        \
    const size_t lds = sizeof(double) * (size_t)p_draw_doubles<NB_>(a.m, a.r);             \
    hipLaunchKernelGGL((kf_fwd_b4p_kernel<NB_>), dim3(grid), dim3(P_THREADS), lds, stream, a); \
  }
  switch (nb) {
    case 1: KF_B4P_LAUNCH(1) break;
    case 2: KF_B4P_LAUNCH(2) break;
    case 3: KF_B4P_LAUNCH(3) break;
    default: KF_B4P_LAUNCH(4) break;
  }
#undef KF_B4P_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
