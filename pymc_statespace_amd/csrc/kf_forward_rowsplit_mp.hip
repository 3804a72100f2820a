// kf_forward_rowsplit_mp.hip -- row-owner forward Kalman filter for MULTIVARIATE observations
// (runtime p <= 8, runtime m <= compile-time even M <= 16), static system matrices.
//
// BASELINE config C4 (VARMAX(1,1), k_endog 5: m = 15 / 10, p = 5) ran on the generic
// LDS-resident kernel (kf_forward_lds.hip): every matrix element goes through LDS for every
// product, ~13 us per time step.  This kernel keeps the row-owner mapping of
// kf_forward_rowsplit.hip (row i of P in the registers of the H lanes that own it, the two
// m x m x m products split over H lanes per row) and processes the p observations of a step
// ONE AT A TIME (Koopman & Durbin sequential processing, the reference's UnivariateFilter,
// kalman_filter.py:444-505).  With a diagonal observation-noise covariance H the sequential
// update is algebraically identical to the joint update of Standard/CholeskyFilter
// (kalman_filter.py:255-318): the q-th conditional innovation variance F_q is the q-th pivot of
// chol(F), log|F| = sum log F_q, v'F^-1 v = sum v_q^2/F_q, and skipping a missing observation is
// the row deletion the reference approximates with its masking.  The host only
// routes the joint filters here when the caller flags H as diagonal (KF_FLAG_H_DIAGONAL);
// UnivariateFilter reads only diag(H) anyway (:463).
//
// Lane (i,h) holds only ITS column pairs of row i of P.  Per observation q: the H lanes of a row
// add up their partial dot products M_i = P[i,:] Z_q' (DPP within a quad for H = 2, 4), one
// wave-local all-gather publishes M and the partial innovations (2M doubles), then O(M/H) FMAs
// per lane: P[i,own] -= M_i M_own/F_q, a_i += M_i v_q / F_q.
// log F_q is not evaluated in the step loop: the mantissa product and exponent sum of the F_q
// are kept and one log per (draw, step) is taken 64 steps at a time (ring, as in
// kf_forward_rowsplit.hip).  y is staged to LDS 64 steps at a time.
//
// Padded rows/columns (m < M) hold exact zeros everywhere; tapes are written with the real m.
#include "kf_internal.h"

#if !defined(KF_ROW_M) || !defined(KF_ROW_H)
#error "compile with -DKF_ROW_M=<even padded k_states> -DKF_ROW_H=<lanes per row>"
#endif

namespace {

template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  __builtin_nontemporal_store(v, p);
}

typedef double double2_t __attribute__((ext_vector_type(2)));

// lane <- lane ^ 1 / lane ^ 2 within a quad (DPP quad_perm: no LDS traffic)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// sum over the H lanes that own one row (lanes base .. base+H-1, base = lane - h); every lane
// of the row gets bit-identical results
template <int H>
__device__ __forceinline__ double row_sum(double x, int lane, int h) {
  // pin the value: fma(a, b, neighbour's ROUNDED a*b) would differ between the lanes of a row
  // (-ffp-contract=fast fuses in the backend whatever a contract pragma says)
  asm volatile("" : "+v"(x));
  if (H == 2) return x + dpp_quad<0xB1>(x);
  if (H == 4) {
    x += dpp_quad<0xB1>(x);
    return x + dpp_quad<0x4E>(x);
  }
  if (H == 1) return x;
  const int base = lane - h;  // callers pass the REAL lane of (draw, row i, h): see `rlane`
  double s = __shfl(x, base, KF_WAVE);
  for (int k = 1; k < H; ++k) s += __shfl(x, base + k, KF_WAVE);
  return s;
}

template <int M, int H>
struct MpCfg {
  static_assert(M % 2 == 0, "row-split kernel needs an even (padded) state dimension");
  static constexpr int MM = M * M;
  static constexpr int NPAIR = M / 2;
  static constexpr int NP = (NPAIR + H - 1) / H;
  static constexpr int LPD = M * H;
  static_assert(LPD <= KF_WAVE, "M*H must fit a wavefront");
  static constexpr int DPW = KF_WAVE / LPD;
  static constexpr int RING = 64;
  static constexpr int NST = (DPW * MM + KF_WAVE - 1) / KF_WAVE;  // 8-byte staging slots per lane
  // per-draw LDS region (doubles): TLh, X, Y | V | ring | Z (p x M) | diag H, d | [y block]
  static constexpr int OFF_V = 3 * MM;
  static constexpr int OFF_RG = OFF_V + 2 * M;
  static constexpr int OFF_Z = OFF_RG + 2 * RING;
};

// host + device: layout of the runtime-sized tail of the region
__host__ __device__ inline int mp_off_hd(int offz, int M, int p) { return offz + p * M; }
__host__ __device__ inline int mp_off_yb(int offz, int M, int p) { return (mp_off_hd(offz, M, p) + 2 * p + 1) & ~1; }
__host__ __device__ inline int mp_stride(int offz, int M, int p, bool y_per_draw) {
  int raw = mp_off_yb(offz, M, p) + (y_per_draw ? 64 * p : 0);
  return ((raw + 29) / 32) * 32 + 2;  // == 2 (mod 32): neighbouring draws on different banks
}

// Two wavefronts per SIMD hide each other's LDS round trips: cap the allocation at 256 VGPRs
// where that costs at most a couple of spilled loop invariants (measured; M = 16, H = 2 would
// spill ~150 and stays at one wave with AGPR overflow).
#if KF_ROW_M * KF_ROW_H >= 64 || (KF_ROW_M < 16 && KF_ROW_M >= 10)
#define KF_MP_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define KF_MP_OCC
#endif

template <int M, int H>
__global__ __launch_bounds__(KF_WAVE) KF_MP_OCC void kf_fwd_rowsplit_mp_kernel(const KfFwdArgs A) {
  using C = MpCfg<M, H>;
  constexpr int MM = C::MM, NP = C::NP;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  int grp = lane / C::LPD;
  const int lid = lane - grp * C::LPD;
  const int i = lid / H;
  const int h = lid - i * H;
  const bool lane_ok = grp < C::DPW;
  if (!lane_ok) grp = C::DPW - 1;
  // spare lanes shadow the last draw: their row sums read the real lanes of that draw
  const int rlane = lane_ok ? lane : grp * C::LPD + lid;
  const long b0 = (long)blockIdx.x * C::DPW;
  long b = b0 + grp;
  const bool live = lane_ok && b < A.B;
  if (b >= A.B) b = A.B - 1;
  const int n = A.T, r = A.r, m = A.m, p = A.p, mm = m * m;
  const bool y_per_draw = (A.flags & KF_FLAG_Y_PER_DRAW) != 0;
  const int stride = mp_stride(C::OFF_Z, M, p, y_per_draw);
  const int off_hd = mp_off_hd(C::OFF_Z, M, p);
  const bool row_ok = i < m;

  double* TLh = smem + grp * stride;
  double* X = TLh + MM;
  double* Y = X + MM;
  double* V = TLh + C::OFF_V;
  double* RG = TLh + C::OFF_RG;
  double* Zt = TLh + C::OFF_Z;
  double* Hd = TLh + off_hd;
  double* dl = Hd + p;
  // y block of the current 64 steps: per draw, or one per wavefront behind the draw regions
  double* yb = y_per_draw ? TLh + mp_off_yb(C::OFF_Z, M, p) : smem + C::DPW * stride;

  int j0[NP];
  double wown[NP];  // 0 for a surplus slot that clones another lane's pair
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    int c = h + pp * H;
    wown[pp] = (c > C::NPAIR - 1) ? 0.0 : 1.0;
    if (c > C::NPAIR - 1) c = C::NPAIR - 1;
    j0[pp] = 2 * c;
  }

  // ---- per-draw constants ------------------------------------------------------------------------
  double Trow[M];
  double RQRo[NP][2], Po[NP][2];
  {
    const double* gT = A.Tm + b * (long)mm;
    const double* gP = A.P0 + b * (long)mm;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const bool ok = row_ok && k < m;
      Trow[k] = ok ? gT[i * m + k] : 0.0;
      TLh[i * M + k] = 0.5 * Trow[k];
    }
    const double* gR = A.R + b * (long)m * r;
    const double* gQ = A.Q + b * (long)r * r;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = j0[pp] + e;
        double s = 0.0, pv = 0.0;
        if (row_ok && j < m) {
          for (int l = 0; l < r; ++l) {
            double rq = 0.0;
            for (int k = 0; k < r; ++k) rq += gR[i * r + k] * gQ[k * r + l];
            s += rq * gR[j * r + l];
          }
          pv = (j >= i) ? gP[i * m + j] : gP[j * m + i];
        }
        RQRo[pp][e] = 0.5 * s;
        Po[pp][e] = pv;
      }
    }
    const double* gZ = A.Z + b * (long)p * m;
    for (int e = lid; e < p * M; e += C::LPD) {
      const int q = e / M, k = e - q * M;
      Zt[e] = (k < m) ? gZ[q * m + k] : 0.0;
    }
    for (int q = lid; q < p; q += C::LPD) {
      Hd[q] = A.H[b * (long)p * p + q * p + q];
      dl[q] = A.d ? A.d[b * (long)p + q] : 0.0;
    }
  }
  const double ci = (A.c && row_ok) ? A.c[b * (long)m + i] : 0.0;
  double ai = row_ok ? A.a0[b * (long)m + i] : 0.0;

  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :340 adds d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;
  const double joint_const = (A.filter == KF_FILTER_CHOLESKY) ? (double)p : 1.0;

  const bool own = live && h == 0 && row_ok;
  double* o_fa = (A.filt_a && own) ? A.filt_a + b * (long)n * m + i : nullptr;
  double* o_pa = (A.pred_a && own) ? A.pred_a + b * (long)(n + 1) * m + i : nullptr;

  // ---- coalesced tape staging: lane -> one real matrix element per slot -------------------------------
  const bool stage_fP = A.filt_P != nullptr, stage_pP = A.pred_P != nullptr;
  int st_lds[C::NST];
  bool st_on[C::NST];
  int gfo[C::NST], gpo[C::NST];  // element offsets from the block's first draw (fit 31 bits: host check)
  double* const base_fP = stage_fP ? A.filt_P + b0 * (long)n * mm : nullptr;
  double* const base_pP = stage_pP ? A.pred_P + b0 * (long)(n + 1) * mm : nullptr;
  {
    const int tile_real = C::DPW * mm;
#pragma unroll
    for (int q = 0; q < C::NST; ++q) {
      int e = lane + KF_WAVE * q;
      st_on[q] = e < tile_real;
      if (!st_on[q]) e = 0;
      const int dr = e / mm;
      const int off = e - dr * mm;
      const int ii = off / m, jj = off - ii * m;
      st_lds[q] = dr * stride + MM + ii * M + jj;  // tile X of draw dr; tile Y = + MM
      long bd = b0 + dr;
      if (bd >= A.B) bd = A.B - 1;  // tail draws are clones of the last draw: identical bytes
      gfo[q] = (int)(bd - b0) * n * mm + off;
      gpo[q] = (int)(bd - b0) * (n + 1) * mm + off;
    }
  }

  if (o_pa) { *o_pa = ai; o_pa += m; }
  if (stage_pP) {
    const double* gP = A.P0 + b * (long)mm + i * m;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = j0[pp] + e;
        Y[i * M + j] = (row_ok && j < m) ? gP[j] : 0.0;
      }
    }
  }
  const bool pending_pP = stage_pP;

  double logp = 0.0;
  int info = 0;

  auto flush_ring = [&](int tb, int cnt) {
    __syncthreads();
#pragma unroll 1  // cold (once per 64 steps): keep log()'s temporaries out of the register budget
    for (int dd = 0; dd < C::DPW; ++dd) {
      const double* rg = smem + dd * stride + C::OFF_RG;
      const double2_t fq = *reinterpret_cast<const double2_t*>(rg + 2 * lane);
      const bool act = lane < cnt;
      const double xv = act ? fq.x : -1.0;
      // -1 marks a step without likelihood contribution; NaN a not-PD innovation covariance
      const double llv = (xv == -1.0) ? 0.0 : -0.5 * (log(xv) + fq.y);
      const long bd = b0 + dd;
      if (act && bd < A.B && A.ll_obs) st_stream(A.ll_obs + bd * (long)n + tb + lane, llv);
      double ssum = llv;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
      if (lane == dd * C::LPD) logp += ssum;
    }
  };

  const double* ybase = A.y + (y_per_draw ? b * (long)n * p : 0L);

  for (int t = 0; t < n; ++t) {
    if ((t & 63) == 0) {
      // stage y[t .. t+64) (all p series); previous readers are done (program order, one wave)
      const int cnt = ((n - t) < 64 ? (n - t) : 64) * p;
      if (y_per_draw) {
        for (int e = lid; e < cnt; e += C::LPD) yb[e] = ybase[(long)t * p + e];
      } else {
        for (int e = lane; e < cnt; e += KF_WAVE) yb[e] = ybase[(long)t * p + e];
      }
      __syncthreads();
    }
    const double* yrow = yb + (t & 63) * p;
    double mant = 1.0, qs = 0.0;
    int ex = 0, cnt_ll = 0, nob = 0;
    bool bad = false;
#pragma unroll 1
    for (int q = 0; q < p; ++q) {
      const double yq = yrow[q];
      const bool obs = !(yq != yq);
      const double* zq = Zt + q * M;
      // ---- M_i = P[i,:] Z_q' ; partial innovation ------------------------------------------------------
      double m0 = 0.0, m1 = 0.0;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const double2_t z2 = *reinterpret_cast<const double2_t*>(zq + j0[pp]);
        const double zx = (NP * H != C::NPAIR) ? wown[pp] * z2.x : z2.x;
        const double zy = (NP * H != C::NPAIR) ? wown[pp] * z2.y : z2.y;
        m0 = fma(Po[pp][0], zx, m0);
        m1 = fma(Po[pp][1], zy, m1);
      }
      const double Mi = row_sum<H>(m0 + m1, rlane, h);
      V[2 * i] = Mi;
      V[2 * i + 1] = zq[i] * ai;
      __syncthreads();  // q == 0: also publishes the predicted rows written to tile Y last step
      if (q == 0 && pending_pP) {
        double* gp = base_pP + (long)t * mm;
        double sv[C::NST];  // all LDS reads first, then the (exec-masked) stores: no wait per store
#pragma unroll
        for (int s = 0; s < C::NST; ++s) sv[s] = smem[st_lds[s] + MM];
#pragma unroll
        for (int s = 0; s < C::NST; ++s)
          if (st_on[s]) st_stream(gp + gpo[s], sv[s]);
      }
      double F0 = Hd[q], F1 = 0.0, z0 = 0.0, z1 = 0.0;
#pragma unroll
      for (int k = 0; k < M; k += 2) {
        const double2_t va = *reinterpret_cast<const double2_t*>(V + 2 * k);
        const double2_t vb = *reinterpret_cast<const double2_t*>(V + 2 * k + 2);
        const double2_t z2 = *reinterpret_cast<const double2_t*>(zq + k);
        F0 = fma(z2.x, va.x, F0);
        z0 += va.y;
        F1 = fma(z2.y, vb.x, F1);
        z1 += vb.y;
      }
      double Mo[NP][2];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        Mo[pp][0] = V[2 * j0[pp]];
        Mo[pp][1] = V[2 * j0[pp] + 2];
      }
      const double F = F0 + F1;
      double v = (yq + dsign * dl[q]) - (z0 + z1);
      // missing -> skipped; UnivariateFilter also skips F == 0 (:468-471)
      const bool degenerate = !obs || (univariate && F == 0.0);
      const bool notpd = !degenerate && !(F > 0.0);
      if (notpd && info == 0) info = ((t + 1) << 8) | (q + 1);
      bad = bad || notpd;
      const double invF = degenerate ? 0.0 : 1.0 / F;
      if (degenerate) v = 0.0;
      nob += obs ? 1 : 0;
      if (!degenerate) {
        int e2;
        mant *= frexp(F, &e2);
        ex += e2;
        qs = fma(v * v, invF, qs);
        // the reference counts log(2 pi) only for scalar terms that are not exactly 0 (:497-500)
        cnt_ll += (F == 1.0 && v == 0.0) ? 0 : 1;
      }
      // ---- scalar update: a += M v/F ; P -= M M'/F (bitwise symmetric: M_i*M_j commutes) -----------
      const double Mis = Mi * invF;
      ai = fma(Mis, v, ai);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        Po[pp][0] = fma(-(Mi * Mo[pp][0]), invF, Po[pp][0]);
        Po[pp][1] = fma(-(Mi * Mo[pp][1]), invF, Po[pp][1]);
      }
    }
    const double afi = ai;
    // ---- all-gather P_f and a_f ----------------------------------------------------------------------------------
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      double2_t t2 = {Po[pp][0], Po[pp][1]};
      *reinterpret_cast<double2_t*>(X + i * M + j0[pp]) = t2;
    }
    V[i] = afi;
    __syncthreads();
    if (lid == 0) {
      double nconst, xv;
      if (univariate) {
        nconst = exact_const ? (double)nob : (double)cnt_ll;
        xv = mant;
      } else {
        nconst = exact_const ? (double)nob : joint_const;
        xv = (nob == 0) ? -1.0 : mant;  // all-missing step: ll = 0
      }
      if (bad) xv = __builtin_nan("");
      double2_t fq = {xv, fma(0.69314718055994530942, (double)ex, qs) + nconst * KF_LOG_2PI};
      if (!univariate && nob == 0) fq.y = 0.0;
      *reinterpret_cast<double2_t*>(RG + 2 * (t & (C::RING - 1))) = fq;
    }
    if (o_fa) { *o_fa = afi; o_fa += m; }
    if (stage_fP) {
      double* gf = base_fP + (long)t * mm;
      double sv[C::NST];
#pragma unroll
      for (int s = 0; s < C::NST; ++s) sv[s] = smem[st_lds[s]];
#pragma unroll
      for (int s = 0; s < C::NST; ++s)
        if (st_on[s]) st_stream(gf + gfo[s], sv[s]);
    }
    // W[i, own] = T[i,:] P_f[:, own] ;  a_hat_i = T[i,:] a_f + c_i
    double Wo[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) Wo[pp][0] = Wo[pp][1] = 0.0;
    double an = ci;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double tk = Trow[k];
      an = fma(tk, V[k], an);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const double2_t x2 = *reinterpret_cast<const double2_t*>(X + k * M + j0[pp]);
        Wo[pp][0] = fma(tk, x2.x, Wo[pp][0]);
        Wo[pp][1] = fma(tk, x2.y, Wo[pp][1]);
      }
      if (M * NP > 32 && (k & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound loads in flight
    }
    ai = an;
    double Wr[M];
    if (H > 1) {
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        double2_t t2 = {Wo[pp][0], Wo[pp][1]};
        *reinterpret_cast<double2_t*>(Y + i * M + j0[pp]) = t2;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < M; k += 2) {
        const double2_t w2 = *reinterpret_cast<const double2_t*>(Y + i * M + k);
        Wr[k] = w2.x;
        Wr[k + 1] = w2.y;
      }
    } else {
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        Wr[2 * pp] = Wo[pp][0];
        Wr[2 * pp + 1] = Wo[pp][1];
      }
    }
    // S[i, own] = W[i,:] T[own,:]' + RQR'[i, own]   (halved: TLh = T/2, RQRo = RQR'/2)
    double So[NP][2];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double* Tj = TLh + (j0[pp] + e) * M;
        double s0 = RQRo[pp][e], s1 = 0.0;
#pragma unroll
        for (int k = 0; k < M; k += 2) {
          const double2_t t2 = *reinterpret_cast<const double2_t*>(Tj + k);
          s0 = fma(Wr[k], t2.x, s0);
          s1 = fma(Wr[k + 1], t2.y, s1);
        }
        So[pp][e] = s0 + s1;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the T-row loads of one pair in flight, not all NP
    }
    // ---- transpose S: P_hat = (S + S')/2 (:221) -------------------------------------------------------------------------
    double* Ttile = (H > 1) ? X : Y;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      double2_t t2 = {So[pp][0], So[pp][1]};
      *reinterpret_cast<double2_t*>(Ttile + i * M + j0[pp]) = t2;
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      Po[pp][0] = So[pp][0] + Ttile[j0[pp] * M + i];
      Po[pp][1] = So[pp][1] + Ttile[(j0[pp] + 1) * M + i];
    }
    if (o_pa) { *o_pa = ai; o_pa += m; }
    if ((t & (C::RING - 1)) == C::RING - 1) flush_ring(t - (C::RING - 1), C::RING);
    if (stage_pP) {
      if (H == 1) __syncthreads();
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        double2_t t2 = {Po[pp][0], Po[pp][1]};
        *reinterpret_cast<double2_t*>(Y + i * M + j0[pp]) = t2;
      }
    }
  }
  if (pending_pP) {
    __syncthreads();
    double* gp = base_pP + (long)n * mm;
#pragma unroll
    for (int s = 0; s < C::NST; ++s)
      if (st_on[s]) st_stream(gp + gpo[s], smem[st_lds[s] + MM]);
  }
  if (n & (C::RING - 1)) flush_ring(n & ~(C::RING - 1), n & (C::RING - 1));
  if (live && lid == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

#define KF_CAT3_(a, b, c, d) a##b##c##d
#define KF_CAT3(a, b, c, d) KF_CAT3_(a, b, c, d)
int KF_CAT3(kf_launch_forward_rowsplit_mp_m, KF_ROW_M, _h, KF_ROW_H)(kf_handle* h, const KfFwdArgs& a,
                                                                    hipStream_t stream) {
  constexpr int M = KF_ROW_M, HH = KF_ROW_H;
  using C = MpCfg<M, HH>;
  if (a.m > M || a.p > 8 || a.p < 1) {
    KF_SET_ERR(h, "forward(rowsplit_mp<%d,%d>): m=%d p=%d out of range", M, HH, a.m, a.p);
    return KF_ERR_UNSUPPORTED;
  }
  const bool ypd = (a.flags & KF_FLAG_Y_PER_DRAW) != 0;
  const int stride = mp_stride(C::OFF_Z, M, a.p, ypd);
  const size_t lds = sizeof(double) * ((size_t)C::DPW * stride + (ypd ? 0 : 64 * a.p));
  static size_t lds_allowed = 64 * 1024;
  if (lds > lds_allowed) {
    KF_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&kf_fwd_rowsplit_mp_kernel<M, HH>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, KF_LDS_BYTES_MAX));
    lds_allowed = KF_LDS_BYTES_MAX;
  }
  long blocks = (a.B + C::DPW - 1) / C::DPW;
  hipLaunchKernelGGL((kf_fwd_rowsplit_mp_kernel<M, HH>), dim3((unsigned)blocks), dim3(KF_WAVE), lds, stream, a);
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
