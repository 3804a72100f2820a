// kf_b4.h -- shared pieces of the 4x4x4-block forward kernels (kf_forward_b4.hip, kf_forward_b4s.hip):
// the MFMA wrappers, the per-draw constants of the augmented tile and the LDS transpose tile geometry.
// Layout and algebra: see the header of kf_forward_b4.hip.
#pragma once
#include "kf_internal.h"
#include "kf_tile.h"

namespace {

constexpr int TRS = 18;  // transpose tile row stride (doubles): conflict-free 16-byte reads
// setup scratch (aliases the loop buffers): T, P0, R, Q, Z, a0, c, R_ext, R_ext Q, Z T, P0 Z', 3 scalars
inline int b4_setup_doubles(int m, int r) { return 2 * m * m + m * r + r * r + 5 * m + 32 * r + 4; }

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double mfma4_nega(double a, double b, double c) {  // c - a b   (neg:[1,0,0])
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1);
}

// rows {2, 3} of the wavefront (lanes 32..63) copied over rows {0, 1}: v_permlane32_swap
__device__ __forceinline__ double high_half_everywhere(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[1], a[1]);
}

// rows {0, 1} (lanes 0..31) copied over rows {2, 3}
__device__ __forceinline__ double low_half_everywhere(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]);
}

// 1/sqrt(x) to rounding accuracy: v_rsq_f64 + two Newton steps (x finite, normal, > 0)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  r = fma(r, fma(-hx * r, r, 0.5), r);
  r = fma(r, fma(-hx * r, r, 0.5), r);
  return r;
}

// wave-wide max |x| on the high dwords (integer max orders non-negative doubles; Inf / NaN come out largest): a wave-uniform
// lower bound of max |x| within 2^-20
__device__ __forceinline__ unsigned abs_hi(double x) { return (unsigned)__double2hiint(x) & 0x7fffffffu; }
__device__ __forceinline__ double wave_absmax_hi(unsigned h) {
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x128, 0xf, 0xf, false));
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x124, 0xf, 0xf, false));
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x122, 0xf, 0xf, false));
  h = max(h, (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x121, 0xf, 0xf, false));
  auto a = __builtin_amdgcn_permlane16_swap((int)h, (int)h, false, false);
  h = max((unsigned)a[0], (unsigned)a[1]);
  auto c = __builtin_amdgcn_permlane32_swap((int)h, (int)h, false, false);
  h = max((unsigned)c[0], (unsigned)c[1]);
  return __hiloint2double(__builtin_amdgcn_readfirstlane((int)h), 0);
}
__device__ __forceinline__ double uniform_f64(double x) {  // lane 0's value as a wave-uniform scalar
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// One output tape of one draw, streamed through an LDS ring.  Why: measured on MI355X (scripts/r02, headline
// shape) the per-step stores of kf_forward_mfma.hip -- 800-byte covariance pieces that start and end in the middle
// of 128-byte lines, 80-byte state vectors -- cost 0.75 ms on top of 1.09 ms of arithmetic: partially written
// lines are expensive at the memory side (a tape written with 224-byte holes per KiB ran at HALF the speed of one
// written in full KiB), and an 80-byte store costs a whole wave-instruction.  Here elements are appended to a ring
// indexed by their ABSOLUTE element number (address / 8 relative to a GRAN-aligned base) and leave for HBM only as
// windows of GRAN doubles that start on GRAN * 8-byte boundaries: 16 bytes per lane, whole lines, every lane busy.
// LDS by absolute byte address (the low dword of a generic pointer into LDS): a ds_write whose address register is the ring slot
// itself -- through a generic pointer the compiler adds the (zero) base of the dynamic LDS block again at every store
typedef double __attribute__((address_space(3))) kf_lds_f64;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ void lds_store_f64(unsigned addr, double v) { *(kf_lds_f64*)(size_t)addr = v; }
__device__ __forceinline__ double lds_load_f64(unsigned addr) { return *(const kf_lds_f64*)(size_t)addr; }

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
  // the same as absolute LDS byte addresses, for rings that START ON A MULTIPLE OF THEIR SIZE (8 RING bytes): advancing is an add and
  // one bit-field insert, and the address goes into the ds_write as it stands
  __device__ __forceinline__ unsigned slot_abs(int o) const { return lds_addr(ring) + slot(o); }
  __device__ __forceinline__ static unsigned next_abs(unsigned addr, int cnt) {
    constexpr unsigned MASK = (unsigned)(8 * RING - 1);
    return (addr & ~MASK) | ((addr + 8u * (unsigned)cnt) & MASK);
  }
  __device__ __forceinline__ static unsigned back_abs(unsigned addr, int cnt) {  // `cnt` elements back (the previous append)
    constexpr unsigned MASK = (unsigned)(8 * RING - 1);
    return (addr & ~MASK) | ((addr + 8u * (unsigned)(RING - cnt)) & MASK);
  }
  __device__ __forceinline__ void window(int w0, int lo, int hi, int lane, double scale) {
    // (w0 is a multiple of GRAN and a window never wraps: the ring position is a wave-uniform base plus a lane constant -- one
    // vector add per window instead of the add / and / shift of the general element number)
    const int lane2 = 2 * (lane & (GRAN / 2 - 1));
    const int e = w0 + 2 * lane;
    double2_t v = *reinterpret_cast<const double2_t*>(ring + ((w0 & (RING - 1)) + lane2));
    v.x *= scale;
    v.y *= scale;
    char* dst = reinterpret_cast<char*>(tape + (w0 - e0));  // (may point below the tape: only lanes >= lo store)
    const unsigned off = 16u * (unsigned)lane;
    if (lo <= w0 && hi >= w0 + GRAN) {  // wave-uniform: the whole window, every lane
      // (write-back; non-temporal window stores measured the same: 1.755 vs 1.783 ms)
      if (GRAN == 128 || lane < GRAN / 2) *reinterpret_cast<double2_t*>(dst + off) = v;
    } else if (lane < GRAN / 2) {
      const bool k0 = e >= lo && e < hi, k1 = e + 1 >= lo && e + 1 < hi;
      if (k0 && k1) *reinterpret_cast<double2_t*>(dst + off) = v;
      else if (k0) *reinterpret_cast<double*>(dst + off) = v.x;
      else if (k1) *reinterpret_cast<double*>(dst + off + 8) = v.y;
    }
  }
  // after the lanes wrote `cnt` more elements into the ring: at most one window becomes complete per append.
  // (Tried and dropped, both slower at the headline shape: reading the window right after the ring writes and
  // storing it later in the step -- 2.05 vs 1.84 ms, register pressure; collecting the windows of all four tapes
  // once per step, reads first -- 1.89 ms.)
  __device__ __forceinline__ void appended(int cnt, int lane, double scale) {
    E += cnt;
    const int bound = E & ~(GRAN - 1);
    if (bound > Ef) {
      wave_lds_fence();
      window(bound - GRAN, Ef, bound, lane, scale);
      wave_lds_fence();
      Ef = bound;
    }
  }
  __device__ __forceinline__ void finish(int lane, double scale) {
    if (E > Ef) {
      wave_lds_fence();
      window(Ef & ~(GRAN - 1), Ef, E, lane, scale);
      Ef = E;
    }
  }
};

// Per-draw constants in register layout (lane = 16 g + 4 b + q):
//   Ta[I][K]  left-factor block T_ext(I, K) replicated over the four slots: lane (g = k, any b, q = i) holds T_ext[4I+q][4K+g]
//   Tha       = Ta / 2 (kept as registers for NB <= 3; else the 1/2 is applied after the symmetrisation)
//   Ch[I]     row-block register I of C_ext / 2 (C_ext itself when !HALF), bitwise symmetric
//   P[I]      row-block register I of the initial tile P0_ext
template <int NB, bool HALF_ = (NB <= 3)>
struct B4Const {
  static constexpr bool HALF = HALF_;
  double Ta[NB][NB];
  double Tha[HALF ? NB : 1][HALF ? NB : 1];
  double Ch[NB];
  double P[NB];
};

// `scratch`: LDS private to this wavefront, >= max(b4_setup_doubles(m, r), 16 * TRS) doubles.  All
// synchronisation is wave-local (LDS executes a wavefront's instructions in order).  On return
// scratch[0 .. 16 TRS) is a zeroed transpose tile except for the entries the registers wrote.
// `tile` (default: scratch itself): where the zeroed transpose tile is left, anywhere inside the scratch.
// `P0w` / `a0w` (default: the draw's P0 / a0): another initial covariance (m x m, row-major) and mean (m; with `a0_zero` set: zeros) --
// the time-chunked pipeline starts chunk j > 0 of a draw from ck_start[b][j] and the mean c (kf_forward_b4c.hip).
template <int NB, bool HALF_>
__device__ __forceinline__ void b4_setup(const KfFwdArgs& A, long b, double* scratch, int lane, B4Const<NB, HALF_>& K,
                                         double* tile = nullptr, const double* P0w = nullptr, const double* a0w = nullptr,
                                         bool a0_zero = false) {
  constexpr int N4 = 4 * NB, JA = N4 - 2, JC = N4 - 1;
  constexpr bool HALF_STATIC = HALF_;
  const int g = lane >> 4, bq = lane & 15;
  const int fb = (lane & 3) * 4 + ((lane >> 2) & 3);
  const int r_ = A.r, m = A.m, mm = m * m;
  const double* gP = P0w ? P0w : A.P0 + b * (long)mm;
  {
    double* sT = scratch;              // T    m x m
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
        sa[lane] = a0_zero ? 0.0 : (a0w ? a0w[lane] : A.a0[b * (long)m + lane]);
        sc[lane] = A.c ? A.c[b * (long)m + lane] : 0.0;
      }
    }
    wave_lds_fence();
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
    wave_lds_fence();
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
    wave_lds_fence();
    const int cc = bq;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      const int rr = 4 * I + g;
      // replicated left-factor blocks: Ta[I][K] at lane (g = k, any b, q = i) = T_ext[4I+q][4K+g]
#pragma unroll
      for (int Kk = 0; Kk < NB; ++Kk) {
        const int ti = 4 * I + (lane & 3), tk = 4 * Kk + g;
        double tv = 0.0;
        if (tk < m) tv = (ti < m) ? sT[ti * m + tk] : (ti == JA ? sZT[tk] : 0.0);
        else if (ti == JC && tk == JC) tv = 1.0;
        K.Ta[I][Kk] = tv;
        if (HALF_STATIC) {
          double th = 0.5 * tv;
          asm volatile("" : "+v"(th));  // keep T/2 in registers: the compiler otherwise re-multiplies NB^2 times per step
          K.Tha[HALF_STATIC ? I : 0][HALF_STATIC ? Kk : 0] = th;
        }
      }
      // P0_ext[rr][cc]
      double pv = 0.0;
      if (rr < m) pv = (cc < m) ? P0s(rr, cc) : (cc == JA ? sPZ[rr] : (cc == JC ? 2.0 * sa[rr] : 0.0));
      else if (rr == JA) pv = (cc < m) ? sPZ[cc] : (cc == JA ? sS[0] : (cc == JC ? 2.0 * sS[1] : 0.0));
      else if (rr == JC) pv = (cc < m) ? 2.0 * sa[cc] : (cc == JA ? 2.0 * sS[1] : 0.0);
      K.P[I] = pv;
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
      K.Ch[I] = HALF_STATIC ? 0.5 * cv : cv;
    }
    wave_lds_fence();  // scratch is dead: the loop buffers may be written
  }
  // C_ext must be bitwise symmetric (both mirrored entries are the same sum in a different order otherwise):
  // transpose once through the tile and average -- also zeroes the tile rows no register ever writes
  double* TR = tile ? tile : scratch;
  for (int e = lane; e < 16 * TRS; e += KF_WAVE) TR[e] = 0.0;
  wave_lds_fence();
#pragma unroll
  for (int I = 0; I < NB; ++I) TR[(4 * I + g) * TRS + fb] = K.Ch[I];
  wave_lds_fence();
#pragma unroll
  for (int I = 0; I < NB; ++I) K.Ch[I] = 0.5 * (K.Ch[I] + TR[bq * TRS + 4 * g + I]);
  wave_lds_fence();
}

}  // namespace
