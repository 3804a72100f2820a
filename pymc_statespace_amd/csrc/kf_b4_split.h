// kf_b4_split.h -- workspace of the split forward pipeline (kf_forward_b4.hip + kf_forward_b4_steady.hip).
#pragma once
#include "kf_internal.h"
#include "kf_tile.h"

constexpr int B4_HS_STRIDE = 4 * KF_WAVE;  // doubles per draw in hs_tile: the register image, row-major [16][16]
constexpr int B4_HC_STRIDE = 24 * KF_WAVE;  // doubles per draw in hs_const (NB <= 3: 2 * 9 + 3 registers)
constexpr int B4_SPLIT_MAX_CHUNKS = 32;
constexpr int KF_B4_STEADY_MIN_T = 1536;  // series length from which kf_forward_b4.hip uses its steady-state phase by default
constexpr int B4_PAIR_MAX = 1024;  // draws left in the full recursion at which the pair kernel (kf_forward_b4p.hip) takes them over: one per SIMD

struct B4SplitWs {
  double* hs_tile;     // [B][B4_HS_STRIDE]
  double* hs_const;    // [B][B4_HC_STRIDE]
  double* hs_logp;     // [B]
  int32_t* hs_info;    // [B]
  int32_t* hs_ts;      // [B]
  int32_t* hs_tfree;   // [B]
  int32_t* listR;      // [B] draws whose recursion continues
  int32_t* listS;      // [B] draws handed to the steady kernels, in hand-over order
  int32_t* listS_ts;   // [B] their first steady step
  int32_t* cnt;        // [4] cnt[0] = entries of listR
  int32_t* pair_ctl;   // [4] {draws handed to the pair kernel (they are listR[0 ..)), their first step, 1 + the chunk that decided it (0: not yet)}
  int32_t* fillq;      // [B4_SPLIT_MAX_CHUNKS] next fill item of the entries chunk c handed over
  int32_t* offS;       // [B4_SPLIT_MAX_CHUNKS + 1] chunk c handed over listS[offS[c] .. offS[c+1])
};

// carve the workspace (base may be NULL: size only); returns the bytes needed
inline size_t kf_b4_split_carve(void* base, long B, int m, int n, B4SplitWs* W) {
  (void)n;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char* p = (char*)base;
  size_t off = 0;
  (void)m;
  W->hs_tile = (double*)(p + off); off += up(sizeof(double) * (size_t)B * B4_HS_STRIDE);
  W->hs_const = (double*)(p + off); off += up(sizeof(double) * (size_t)B * B4_HC_STRIDE);
  W->hs_logp = (double*)(p + off); off += up(sizeof(double) * (size_t)B);
  W->hs_info = (int32_t*)(p + off); off += up(4 * (size_t)B);
  W->hs_ts = (int32_t*)(p + off); off += up(4 * (size_t)B);
  W->hs_tfree = (int32_t*)(p + off); off += up(4 * (size_t)B);
  W->listR = (int32_t*)(p + off); off += up(4 * (size_t)B);
  W->listS = (int32_t*)(p + off); off += up(4 * (size_t)B);
  W->listS_ts = (int32_t*)(p + off); off += up(4 * (size_t)B);
  W->cnt = (int32_t*)(p + off); off += up(16);
  W->pair_ctl = (int32_t*)(p + off); off += up(16);
  W->fillq = (int32_t*)(p + off); off += up(4 * (size_t)B4_SPLIT_MAX_CHUNKS);
  W->offS = (int32_t*)(p + off); off += up(4 * (size_t)(B4_SPLIT_MAX_CHUNKS + 1));
  return off + 256;  // (+ alignment slack for a base that is not 256-byte aligned)
}

// One tape range [dst, dst + count) of a periodic image: element i = img[(phase + i) mod mm]; img has mm + 1 entries
// (img[mm] = img[0]).  Single elements up to the first 128-byte line boundary, then 16 bytes per lane, whole lines.
// Windows [w_lo, w_hi) of 128 elements (counted from the first line boundary) are this caller's share; the head belongs
// to the caller with w_lo == 0.
#ifdef __HIPCC__
static __device__ __forceinline__ void fill_periodic(double* dst, long count, const double* img, int mm, int phase, int lane,
                                              long w_lo, long w_hi) {
  const long a0 = (long)((reinterpret_cast<unsigned long long>(dst) >> 3) & 15ull);
  long lead = (16 - a0) & 15;
  if (lead > count) lead = count;
  if (w_lo == 0 && lane < lead) dst[lane] = img[(phase + lane) % mm];
  const int step = 128 % mm;
  long pos = lead + 128 * w_lo + 2 * lane;
  int off = (int)((phase + pos) % mm);
  const long end = (lead + 128 * w_hi < count) ? lead + 128 * w_hi : count;
  for (; pos < end; pos += 128) {
    if (pos + 1 < end) {
      double2_t v = {img[off], img[off + 1]};
      *reinterpret_cast<double2_t*>(dst + pos) = v;
    } else {
      dst[pos] = img[off];
    }
    off += step;
    if (off >= mm) off -= mm;
  }
}
#endif

// the split pipeline serves: p == 1, m <= 10 (covariance tapes through the rings), static matrices, steady-state phase allowed
bool kf_forward_b4_split_ok(const kf_handle* h, long B, int T, int p, int m, int r, unsigned flags);
size_t kf_b4_split_workspace_bytes(long B, int m, int n);
int kf_b4_split_compact(kf_handle* h, const B4SplitWs& W, long B, int chunk, int t_next, int pair_ok, hipStream_t stream);
int kf_b4_split_steady(kf_handle* h, const KfFwdArgs& a, const B4SplitWs& W, int chunk, hipStream_t s_fill, hipStream_t s_mean);
int kf_b4_split_drain(kf_handle* h, const KfFwdArgs& a, const B4SplitWs& W, int last_chunk, hipStream_t stream);
