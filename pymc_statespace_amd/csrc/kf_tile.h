// kf_tile.h -- cross-lane helpers for the kernels that keep a 16 x 16 tile in the register layout of
// v_mfma_f64_16x16x4_f64's accumulator: lane l = (g = l >> 4, c = l & 15), register s holds
// X[g + 4s][c].  Used as B operand the registers are X, used as A operand they are X' (so a
// symmetric X serves as either).  All helpers assume one 64-lane wavefront per workgroup.
#pragma once
#include "kf_internal.h"

namespace {


typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  // plain write-back stores: measured 1.4x faster here than non-temporal ones (headline shape,
  // full tapes: 1.99 vs 2.78 ms) -- L2 assembles the lines before they go to HBM
#ifdef KF_MFMA_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// Ordering point between LDS accesses of the lanes of ONE wavefront.  A wavefront's LDS
// instructions execute in order, so a ds_read issued after a ds_write of another lane already
// sees the data; only the compiler must not reorder them.  Unlike __syncthreads() this does not
// emit `s_waitcnt lgkmcnt(0)` on the spot: the wait lands where the loaded value is first used.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// all-reduce over the 16 lanes of a DPP row (same g): every lane gets the bit-identical sum
__device__ __forceinline__ double row_allsum(double x) {
  // no FMA contraction with the caller's product: fma(a, b, other lane's ROUNDED a*b) would give
  // the two lanes of a pair different bits (-ffp-contract=fast fuses in the backend whatever the
  // pragma says, so the value is pinned with an empty asm)
  asm volatile("" : "+v"(x));
  x += dpp_mov<0x128>(x);  // row_ror:8
  x += dpp_mov<0x124>(x);  // row_ror:4
  x += dpp_mov<0x122>(x);  // row_ror:2
  x += dpp_mov<0x121>(x);  // row_ror:1
  return x;
}

// all-reduce over the 4 lane groups g (lanes c, c+16, c+32, c+48): (x0 + x1) + (x2 + x3) everywhere
__device__ __forceinline__ double readlane_f64(double x, int src_lane) {  // src_lane wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
  return __hiloint2double(hi, lo);
}

// value held by lane group g_src (compile-time) -> all four groups, exact (no arithmetic)
template <int G_SRC>
__device__ __forceinline__ double group_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  // [0] = rows {0,0,2,2}, [1] = rows {1,1,3,3}
  lo = (G_SRC & 1) ? a[1] : a[0];
  hi = (G_SRC & 1) ? b[1] : b[0];
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  // [0] = low half everywhere, [1] = high half everywhere
  return (G_SRC & 2) ? __hiloint2double(d[1], c[1]) : __hiloint2double(d[0], c[0]);
}

// group_bcast with the source group chosen by a value that is a compile-time constant after unrolling
__device__ __forceinline__ double group_bcast_sel(double x, int g_src) {
  switch (g_src) {
    case 0: return group_bcast<0>(x);
    case 1: return group_bcast<1>(x);
    case 2: return group_bcast<2>(x);
    default: return group_bcast<3>(x);
  }
}

// all-reduce max over the wavefront
__device__ __forceinline__ double wave_allmax(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, KF_WAVE));
  return x;
}

// 1/x to rounding accuracy: v_rcp_f64 + two Newton steps (x finite, normal, > 0)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double col_allsum(double x) {
  asm volatile("" : "+v"(x));
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(x);
  hi = __double2hiint(x);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}


}  // namespace
