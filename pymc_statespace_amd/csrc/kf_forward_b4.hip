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
#include <type_traits>

#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int TRS = 18;                 // transpose tile row stride (doubles): conflict-free 16-byte reads
constexpr int LB = 32;                  // steps per y block / likelihood ring
// loop-time LDS (doubles); 16 wavefronts per CU must fit 160 KiB: <= 1280 doubles each
constexpr int B4_OFF_UX = 16 * TRS;               // u by column index (16) | zeros (4)
constexpr int B4_OFF_RG = B4_OFF_UX + 20;         // likelihood ring: LB x {F, u_hat_JC}
constexpr int B4_OFF_YB = B4_OFF_RG + 2 * LB;     // y block: LB x {-2 (y -+ d) or 0, observed ? 1 : 0}
constexpr int B4_OFF_RA = B4_OFF_YB + 2 * LB;     // state tapes: 2 rings of RING_A
constexpr int RING_A = 128, GRAN_A = 64;
constexpr int B4_OFF_RP = B4_OFF_RA + 2 * RING_A; // covariance tapes: 2 rings of RING_P (or one m x m image + dump slots)
constexpr int RING_P = 256, GRAN_P = 128;
constexpr int B4_LOOP_DOUBLES = B4_OFF_RP + 2 * RING_P;
static_assert(B4_LOOP_DOUBLES <= 1280, "four wavefronts per SIMD need <= 10 KiB of LDS per wavefront");
constexpr int B4_STREAM_MM_MAX = RING_P - GRAN_P + 1;  // largest m*m the covariance rings can stream (m <= 11)
// setup scratch (aliases the loop buffers): T, P0, R, Q, Z, a0, c, R_ext, R_ext Q, Z T, P0 Z', 3 scalars
inline int b4_setup_doubles(int m, int r) { return 2 * m * m + m * r + r * r + 5 * m + 32 * r + 4; }

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double mfma4_nega(double a, double b, double c) {  // c - a b   (neg:[1,0,0])
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1);
}
template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// rows {2, 3} of the wavefront (lanes 32..63) copied over rows {0, 1}: v_permlane32_swap
__device__ __forceinline__ double high_half_everywhere(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[1], a[1]);
}

// 1/sqrt(x) to rounding accuracy: v_rsq_f64 + two Newton steps (x finite, normal, > 0)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  r = fma(r, fma(-hx * r, r, 0.5), r);
  r = fma(r, fma(-hx * r, r, 0.5), r);
  return r;
}

// One output tape of one draw, streamed through an LDS ring.  Why: measured on MI355X (scripts/r02, headline
// shape) the per-step stores of kf_forward_mfma.hip -- 800-byte covariance pieces that start and end in the middle
// of 128-byte lines, 80-byte state vectors -- cost 0.75 ms on top of 1.09 ms of arithmetic: partially written
// lines are expensive at the memory side (a tape written with 224-byte holes per KiB ran at HALF the speed of one
// written in full KiB), and an 80-byte store costs a whole wave-instruction.  Here elements are appended to a ring
// indexed by their ABSOLUTE element number (address / 8 relative to a GRAN-aligned base) and leave for HBM only as
// windows of GRAN doubles that start on GRAN * 8-byte boundaries: 16 bytes per lane, whole lines, every lane busy.
template <int RING, int GRAN>
struct TapeStream {
  double* ring;
  double* tape;  // the draw's tape (kept as the kernel argument's pointer: global_store, not flat_store)
  int e0;        // absolute element number of tape[0]: (address / 8) mod GRAN
  int E;         // absolute element number of the next append (wave-uniform)
  int Ef;        // elements below Ef have been stored

  __device__ __forceinline__ void init(double* tape_, double* lds_ring) {
    ring = lds_ring;
    tape = tape_;
    e0 = tape_ ? (int)((reinterpret_cast<unsigned long long>(tape_) >> 3) & (unsigned long long)(GRAN - 1)) : 0;
    E = Ef = e0;
  }
  // LDS byte offset of the element `o` doubles past the next append position
  __device__ __forceinline__ unsigned slot(int o) const { return (unsigned)(((E + o) & (RING - 1)) * 8); }
  __device__ __forceinline__ static unsigned next(unsigned byte_off, int cnt) {
    return (byte_off + 8u * (unsigned)cnt) & (unsigned)(8 * RING - 1);
  }
  __device__ __forceinline__ void window(int w0, int lo, int hi, int lane, double scale) {
    const int e = w0 + 2 * lane;
    double2_t v = *reinterpret_cast<const double2_t*>(ring + (e & (RING - 1)));
    v.x *= scale;
    v.y *= scale;
    char* dst = reinterpret_cast<char*>(tape + (w0 - e0));  // (may point below the tape: only lanes >= lo store)
    const unsigned off = 16u * (unsigned)lane;
#ifdef B4_NO_GSTORE  // experiment builds (scripts/r02/b4_variants.sh): ring traffic only
    if (v.x == 1.2345e300) *reinterpret_cast<double2_t*>(dst + off) = v;
    return;
#endif
    if (lo <= w0 && hi >= w0 + GRAN) {  // wave-uniform: the whole window, every lane
      if (GRAN == 128 || lane < GRAN / 2) *reinterpret_cast<double2_t*>(dst + off) = v;
    } else if (lane < GRAN / 2) {
      const bool k0 = e >= lo && e < hi, k1 = e + 1 >= lo && e + 1 < hi;
      if (k0 && k1) *reinterpret_cast<double2_t*>(dst + off) = v;
      else if (k0) *reinterpret_cast<double*>(dst + off) = v.x;
      else if (k1) *reinterpret_cast<double*>(dst + off + 8) = v.y;
    }
  }
  // after the lanes wrote `cnt` more elements into the ring: at most one window becomes complete per append
  __device__ __forceinline__ void bump(int cnt) { E += cnt; }
  __device__ __forceinline__ bool ready() const { return (E & ~(GRAN - 1)) > Ef; }
  // read / store halves of flushing the complete window, so that a caller with several streams can issue all
  // LDS reads first and pay ONE round trip before the stores
  __device__ __forceinline__ double2_t read_window(int lane) const {
    const int w0 = (E & ~(GRAN - 1)) - GRAN;
    return *reinterpret_cast<const double2_t*>(ring + ((w0 + 2 * lane) & (RING - 1)));
  }
  __device__ __forceinline__ void store_window(int lane, double2_t v) {
    const int bound = E & ~(GRAN - 1), w0 = bound - GRAN;
    char* dst = reinterpret_cast<char*>(tape + (w0 - e0));
    const unsigned off = 16u * (unsigned)lane;
    if (Ef <= w0) {  // wave-uniform: the whole window (always, except the first window of a tape)
      if (GRAN == 128 || lane < GRAN / 2) *reinterpret_cast<double2_t*>(dst + off) = v;
    } else if (lane < GRAN / 2) {
      const int e = w0 + 2 * lane;
      const bool k0 = e >= Ef, k1 = e + 1 >= Ef;
      if (k0 && k1) *reinterpret_cast<double2_t*>(dst + off) = v;
      else if (k1) *reinterpret_cast<double*>(dst + off + 8) = v.y;
    }
    Ef = bound;
  }
  __device__ __forceinline__ void finish(int lane, double scale) {
    if (E > Ef) {
      wave_lds_fence();
      window(Ef & ~(GRAN - 1), Ef, E, lane, scale);
      Ef = E;
    }
  }
};

// STREAM: covariance tapes through TapeStream rings (m * m <= B4_STREAM_MM_MAX), else one compact image per step
template <int NB, bool STREAM>
__global__ __launch_bounds__(KF_WAVE) __attribute__((amdgpu_waves_per_eu(4))) void kf_fwd_b4_kernel(const KfFwdArgs A) {
  // (waves_per_eu(4): <= 128 registers, and the compiler then keeps MFMA results in VGPRs -- no v_accvgpr moves)
  constexpr int N4 = 4 * NB, JA = N4 - 2, JC = N4 - 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int g = lane >> 4, bq = lane & 15;       // bq = 4 b + q: column index of the lane
  const int fb = (lane & 3) * 4 + ((lane >> 2) & 3);  // 4 q + b: position of column bq in a transpose-tile row
  const long b = blockIdx.x;
  const int n = A.T, r_ = A.r, m = A.m, mm = m * m;

  double* TR = smem;
  double* STG = smem + B4_OFF_RP;  // !STREAM: compact m x m image (<= 196 doubles) + 64 dump slots at 256
  double* UX = smem + B4_OFF_UX;
  double* RG = smem + B4_OFF_RG;
  double* YB = smem + B4_OFF_YB;

  // ---- per-draw constants through an LDS scratch that aliases the loop buffers ----------------------------
  constexpr bool HALF_STATIC = NB <= 3;  // keep T/2 as a second set of replicated blocks (else scale at the end)
  double Ta[NB][NB], Tha[HALF_STATIC ? NB : 1][HALF_STATIC ? NB : 1], Ch[NB], P[NB];
  unsigned stg_off[NB];  // LDS byte offset of this lane's element in the compact tape image (dummy slot if none)
  const double* gP = A.P0 + b * (long)mm;
  {
    double* sT = smem;                 // T    m x m
    double* sP = sT + mm;              // P0   m x m (as given; the upper triangle is used)
    double* sR = sP + mm;              // R    m x r
    double* sQ = sR + m * r_;          // Q    r x r
    double* sZ = sQ + r_ * r_;         // Z    m
    double* sa = sZ + m;               // a0
    double* sc = sa + m;               // c
    double* sZT = sc + m;              // Z T
    double* sPZ = sZT + m;             // P0 Z'
    double* sS = sPZ + m;              // [0] Z P0 Z'   [1] Z a0   [2] Z c
    double* sRe = sS + 4;              // R_ext   16 x r  rows: R | 0 | Z R (row JA)
    double* sRQ = sRe + 16 * r_;       // R_ext Q
    {
      const double* gT = A.Tm + b * (long)mm;
      const double* gR = A.R + b * (long)m * r_;
      const double* gQ = A.Q + b * (long)r_ * r_;
      const double* gZ = A.Z + b * (long)m;
      for (int e = lane; e < mm; e += KF_WAVE) {
        sT[e] = gT[e];
        sP[e] = gP[e];
      }
      for (int e = lane; e < m * r_; e += KF_WAVE) sR[e] = gR[e];
      for (int e = lane; e < r_ * r_; e += KF_WAVE) sQ[e] = gQ[e];
      if (lane < m) {
        sZ[lane] = gZ[lane];
        sa[lane] = A.a0[b * (long)m + lane];
        sc[lane] = A.c ? A.c[b * (long)m + lane] : 0.0;
      }
    }
    __syncthreads();
    auto P0s = [&](int i, int j) { return (j >= i) ? sP[i * m + j] : sP[j * m + i]; };
    for (int e = lane; e < 16 * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      if (i < m) {
        v = sR[e];
      } else if (i == JA) {
        for (int j = 0; j < m; ++j) v += sZ[j] * sR[j * r_ + l];
      }
      sRe[e] = v;
    }
    if (lane < m) {
      double tz = 0.0, pz = 0.0;
      for (int l = 0; l < m; ++l) {
        tz += sZ[l] * sT[l * m + lane];
        pz += P0s(lane, l) * sZ[l];
      }
      sZT[lane] = tz;
      sPZ[lane] = pz;
    }
    __syncthreads();
    for (int e = lane; e < 16 * r_; e += KF_WAVE) {
      const int i = e / r_, l = e - i * r_;
      double v = 0.0;
      for (int kk = 0; kk < r_; ++kk) v += sRe[i * r_ + kk] * sQ[kk * r_ + l];
      sRQ[e] = v;
    }
    if (lane < 3) {
      double v = 0.0;
      for (int l = 0; l < m; ++l) v += sZ[l] * (lane == 0 ? sPZ[l] : lane == 1 ? sa[l] : sc[l]);
      sS[lane] = v;
    }
    __syncthreads();
    const int cc = bq;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g;
      // replicated left-factor blocks: Ta[I][K] at lane (g = k, any b, q = i) = T_ext[4I+q][4K+g]
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const int ti = 4 * I + (lane & 3), tk = 4 * K + g;
        double tv = 0.0;
        if (tk < m) tv = (ti < m) ? sT[ti * m + tk] : (ti == JA ? sZT[tk] : 0.0);
        else if (ti == JC && tk == JC) tv = 1.0;
        Ta[I][K] = tv;
        if (HALF_STATIC) {
          double th = 0.5 * tv;
          asm volatile("" : "+v"(th));  // keep T/2 in registers: the compiler otherwise re-multiplies NB^2 times per step
          Tha[HALF_STATIC ? I : 0][HALF_STATIC ? K : 0] = th;
        }
      }
      // P0_ext[rr][cc]
      double pv = 0.0;
      if (rr < m) pv = (cc < m) ? P0s(rr, cc) : (cc == JA ? sPZ[rr] : (cc == JC ? 2.0 * sa[rr] : 0.0));
      else if (rr == JA) pv = (cc < m) ? sPZ[cc] : (cc == JA ? sS[0] : (cc == JC ? 2.0 * sS[1] : 0.0));
      else if (rr == JC) pv = (cc < m) ? 2.0 * sa[cc] : (cc == JA ? 2.0 * sS[1] : 0.0);
      P[I] = pv;
      // C_ext[rr][cc] / 2: R_ext Q R_ext' on states + JA, the intercept 2 c on the carrier row / column
      double cv = 0.0;
      const bool r_cov = rr < m || rr == JA, c_cov = cc < m || cc == JA;
      if (r_cov && c_cov) {
        for (int l = 0; l < r_; ++l) cv += sRQ[rr * r_ + l] * sRe[cc * r_ + l];
      } else if (rr == JC && cc != JC && c_cov) {
        cv = 2.0 * (cc < m ? sc[cc] : sS[2]);
      } else if (cc == JC && rr != JC && r_cov) {
        cv = 2.0 * (rr < m ? sc[rr] : sS[2]);
      }
      Ch[I] = HALF_STATIC ? 0.5 * cv : cv;
      stg_off[I] = (unsigned)(((rr < m && cc < m) ? rr * m + cc : 256 + lane) * 8);  // (!STREAM only)
    }
    __syncthreads();  // scratch is dead: the loop buffers may be written
  }
  // C_ext must be bitwise symmetric (both mirrored entries are the same sum in a different order otherwise):
  // transpose once through the tile and average -- also zeroes the tile rows no register ever writes
  for (int e = lane; e < 16 * TRS; e += KF_WAVE) TR[e] = 0.0;
  if (lane < 20) UX[lane] = 0.0;
  __syncthreads();
#pragma unroll
  for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = Ch[I];
  __syncthreads();
#pragma unroll
  for (int I = 0; I < NB; ++I) Ch[I] = 0.5 * (Ch[I] + TR[bq * TRS + 4 * g + I]);
  __syncthreads();

  double H1 = A.H[b];
  asm volatile("" : "+v"(H1));  // VGPR: F = H + readlane(...) then is one instruction (one SGPR operand per VALU op)
  const double d1 = A.d ? A.d[b] : 0.0;
  const bool univariate = A.filter == KF_FILTER_UNIVARIATE;
  const double dsign = (A.filter == KF_FILTER_SINGLE) ? 1.0 : -1.0;  // :335 quirk: y_hat = Z a - d
  const bool exact_const = (A.flags & KF_FLAG_MVN_CONST_EXACT) != 0;

  // u exchange: DPP row 0 writes u[column] at position 4 q + b, reads {u[q], u[4+q], u[8+q], u[12+q]};
  // the other rows read zeros
  double* ux_w = UX + fb;
  const double* ux_r = (g == 0) ? UX + 4 * (lane & 3) : UX + 16;
  const double mask0 = (g == 0 && bq < N4) ? 1.0 : 0.0;  // u lives in DPP row 0 (k = 0 of the rank-one MFMA)
  const double adjm = (lane == JC) ? 1.0 : 0.0;
  const bool vec_on = g == 3 && bq < m;                  // row JC of the tile = 2 a'
  double* t_fa = A.filt_a ? A.filt_a + b * (long)n * m : nullptr;
  double* t_pa = A.pred_a ? A.pred_a + b * (long)(n + 1) * m : nullptr;
  double* t_fP = A.filt_P ? A.filt_P + b * (long)n * mm : nullptr;
  double* t_pP = A.pred_P ? A.pred_P + b * (long)(n + 1) * mm : nullptr;
#ifdef B4_NO_PTAPE
  t_fP = t_pP = nullptr;
#endif
#ifdef B4_NO_ATAPE
  t_fa = t_pa = nullptr;
#endif
  const bool w_fa = t_fa != nullptr, w_pa = t_pa != nullptr, w_fP = t_fP != nullptr, w_pP = t_pP != nullptr;

  // state tapes: row JC of the tile (2 a') appended to a ring, scaled by 1/2 on the way out
  TapeStream<RING_A, GRAN_A> s_fa, s_pa;
  s_fa.init(t_fa, smem + B4_OFF_RA);
  s_pa.init(t_pa, smem + B4_OFF_RA + RING_A);
  unsigned o_fa = s_fa.slot(bq), o_pa = s_pa.slot(bq);
  auto put_a = [&](TapeStream<RING_A, GRAN_A>& st, unsigned& off, double twice_a) {
    if (vec_on) *reinterpret_cast<double*>(reinterpret_cast<char*>(st.ring) + off) = twice_a;
    off = TapeStream<RING_A, GRAN_A>::next(off, m);
    st.bump(m);
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
    o_fP[I] = s_fP.slot(on[I] ? rr * m + bq : 0);
    o_pP[I] = s_pP.slot(on[I] ? rr * m + bq : 0);
  }
  auto put_P = [&](TapeStream<RING_P, GRAN_P>& st, unsigned (&off)[NB], const double (&Preg)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      if (on[I]) *reinterpret_cast<double*>(reinterpret_cast<char*>(st.ring) + off[I]) = Preg[I];
      off[I] = TapeStream<RING_P, GRAN_P>::next(off[I], mm);
    }
    st.bump(mm);
  };
  // complete windows of all four tapes -> HBM: every LDS read first, then the stores (one round trip)
  auto flush_tapes = [&]() {
    const bool r1 = w_fa && s_fa.ready(), r2 = w_pa && s_pa.ready();
    const bool r3 = STREAM && w_fP && s_fP.ready(), r4 = STREAM && w_pP && s_pP.ready();
    if (!(r1 || r2 || r3 || r4)) return;
    wave_lds_fence();
    double2_t v1 = {0.0, 0.0}, v2 = v1, v3 = v1, v4 = v1;
    if (r3) v3 = s_fP.read_window(lane);
    if (r4) v4 = s_pP.read_window(lane);
    if (r1) v1 = s_fa.read_window(lane);
    if (r2) v2 = s_pa.read_window(lane);
    wave_lds_fence();
    if (r3) s_fP.store_window(lane, v3);
    if (r4) s_pP.store_window(lane, v4);
    if (r1) s_fa.store_window(lane, 0.5 * v1);
    if (r2) s_pa.store_window(lane, 0.5 * v2);
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

#ifdef B4_DIRECT
  // tapes straight from the registers: register I covers rows 4I..4I+3 = 4 m contiguous doubles (8 B per lane)
  unsigned d_off[NB];
#pragma unroll
  for (int I = 0; I < NB; ++I) d_off[I] = (unsigned)(((4 * I + g) * m + bq) * 8);
  auto direct_P = [&](double*& dst, const double (&Preg)[NB]) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
      if (on[I]) *reinterpret_cast<double*>(reinterpret_cast<char*>(dst) + d_off[I]) = Preg[I];
    dst += mm;
  };
  auto direct_a = [&](double*& dst, double twice_a) {
    if (vec_on) *reinterpret_cast<double*>(reinterpret_cast<char*>(dst) + 8u * (unsigned)bq) = 0.5 * twice_a;
    dst += m;
  };
#endif
  // predicted tape slot 0 = (a0, P0 as given: the caller's P0, not its symmetrised upper triangle)
#ifdef B4_DIRECT
  if (w_pa) direct_a(t_pa, P[NB - 1]);
  if (w_pP) {
    for (int e = lane; e < mm; e += KF_WAVE) t_pP[e] = gP[e];
    t_pP += mm;
  }
#else
  if (w_pa) put_a(s_pa, o_pa, P[NB - 1]);
  if (w_pP) {
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
  flush_tapes();
#endif

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
  double logp = 0.0;
  int info = 0;
  // ll of the steps [tb, tb+cnt) held in the ring, one step per lane: (F, u_hat_JC = -2 v / sqrt F); the flags of
  // these steps are still in the y block.  Everything rare is decided here, LB steps at a time.
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

  for (int t = 0; t < n; ++t) {
    if ((t & (LB - 1)) == 0) {
      const int cnt = (n - t) < LB ? (n - t) : LB;
      if (lane < cnt) {
        const double yv = ybase[(long)t + lane];
        const bool ob = !(yv != yv);
        double2_t e = {ob ? -2.0 * (yv + dsign * d1) : 0.0, ob ? 1.0 : 0.0};
        *reinterpret_cast<double2_t*>(YB + 2 * lane) = e;
      }
      wave_lds_fence();
    }
    const double2_t yo = *reinterpret_cast<const double2_t*>(YB + 2 * (t & (LB - 1)));  // {-2 (y -+ d), observed}
    // ---- update: u = row JA of the tile (DPP row 2 of the last register) brought to DPP row 0 ---------------
    const double raw = high_half_everywhere(P[NB - 1]);
    const double u = fma(raw, mask0, adjm * yo.x);  // u_JC = 2 Z a - 2 (y -+ d) = -2 v
    // left factor of the rank-one step: block I of u replicated over the four slots (k = 0: DPP row 0 only).
    // One LDS exchange, issued before F / rs are known: u_rep[I] at lane (g = 0, any b, q) = u[4 I + q]
    if (g == 0) *ux_w = u;
    wave_lds_fence();
    const double2_t ur01 = *reinterpret_cast<const double2_t*>(ux_r);
    double2_t ur23 = {0.0, 0.0};
    if (NB > 2) ur23 = *reinterpret_cast<const double2_t*>(ux_r + 2);
    wave_lds_fence();
    const double urep[4] = {ur01.x, ur01.y, ur23.x, ur23.y};
    const double F = H1 + readlane_f64(u, JA);
    // missing -> no update (rs = 0).  F <= 0: flagged when the ring is evaluated (info, NaN ll); the clamp only
    // keeps 0 * rsqrt(F) finite on a missing / skipped step (Univariate skips F == 0, :468-471: u is 0 there)
    const double rs = fast_rsqrt(fmax(F, 1e-300)) * yo.y;
    const double uh = u * rs;
    if (lane == JC) {
      double2_t fq = {F, uh};
      *reinterpret_cast<double2_t*>(RG + 2 * (t & (LB - 1))) = fq;
    }
    // tile -= u_hat u_hat'
#pragma unroll
    for (int I = 0; I < NB; ++I) P[I] = mfma4_nega(urep[I] * rs, uh, P[I]);
    if (!STREAM && pred_pending) {  // (image mode) predicted covariance of the previous step: tile image -> HBM
      flush_cov(t_pP);
      t_pP += mm;
      pred_pending = false;
    }
    // ---- filtered tapes -----------------------------------------------------------------------------------------------
#ifdef B4_DIRECT
    if (w_fa) direct_a(t_fa, P[NB - 1]);
    if (w_fP) direct_P(t_fP, P);
#else
    if (w_fa) put_a(s_fa, o_fa, P[NB - 1]);
    if (w_fP) {
      if (STREAM) put_P(s_fP, o_fP, P);
      else stage_cov(P);  // flushed below, behind the MFMAs
    }
#endif
    // ---- predict: tile <- sym(T_ext tile T_ext' + C_ext) ---------------------------------------------------------------------
    double U[NB], Y[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) U[I] = 0.0;
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int I = 0; I < NB; ++I) U[I] = mfma4(Ta[I][K], P[K], U[I]);
#ifndef B4_DIRECT
    flush_tapes();  // windows completed by this step's filtered and the previous step's predicted moments
    if (!STREAM && w_fP) {
      flush_cov(t_fP);
      t_fP += mm;
    }
#endif
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
#ifdef B4_DIRECT
    if (w_pa) direct_a(t_pa, P[NB - 1]);
    if (w_pP) direct_P(t_pP, P);
#else
    if (w_pa) put_a(s_pa, o_pa, P[NB - 1]);
    if (w_pP) {
      if (STREAM) {
        put_P(s_pP, o_pP, P);
      } else {
        stage_cov(P);
        pred_pending = true;
      }
    }
#endif
    if ((t & (LB - 1)) == LB - 1) flush_ring(t - (LB - 1), LB);
  }
#ifndef B4_DIRECT
  if (!STREAM && pred_pending) flush_cov(t_pP);
  flush_tapes();
  s_fa.finish(lane, 0.5);
  s_pa.finish(lane, 0.5);
  if (STREAM) {
    s_fP.finish(lane, 1.0);
    s_pP.finish(lane, 1.0);
  }
#endif
  if (n & (LB - 1)) flush_ring(n & ~(LB - 1), n & (LB - 1));
  if (lane == 0) {
    A.logp[b] = logp;
    if (A.info) A.info[b] = info;
  }
}

}  // namespace

bool kf_forward_b4_ok(int m, int p, int r) { return p == 1 && m >= 1 && m + 2 <= 16 && r >= 1 && r <= 64; }

int kf_launch_forward_b4(kf_handle* h, const KfFwdArgs& a, hipStream_t stream) {
  if (!kf_forward_b4_ok(a.m, a.p, a.r) || (a.flags & KF_FLAG_TV_ANY)) {
    KF_SET_ERR(h, "forward(b4): needs p == 1, m <= 14, static matrices (got p=%d m=%d r=%d)", a.p, a.m, a.r);
    return KF_ERR_UNSUPPORTED;
  }
  const int lds_setup = b4_setup_doubles(a.m, a.r);
  const size_t lds = sizeof(double) * (size_t)(B4_LOOP_DOUBLES > lds_setup ? B4_LOOP_DOUBLES : lds_setup);
  const int nb = (a.m + 2 + 3) / 4;
#define KF_B4_LAUNCH(NB_, ST_) \
  hipLaunchKernelGGL((kf_fwd_b4_kernel<NB_, ST_>), dim3((unsigned)a.B), dim3(KF_WAVE), lds, stream, a)
  const bool stream_p = a.m * a.m <= B4_STREAM_MM_MAX;
  switch (nb) {
    case 1: KF_B4_LAUNCH(1, true); break;
    case 2: KF_B4_LAUNCH(2, true); break;
    case 3: KF_B4_LAUNCH(3, true); break;
    default:
      if (stream_p) KF_B4_LAUNCH(4, true);
      else KF_B4_LAUNCH(4, false);
      break;
  }
#undef KF_B4_LAUNCH
  KF_HIP_CHECK(h, hipGetLastError());
  return KF_OK;
}
