// Counting-sort stage of the Pippenger MSM (msm.hip): digit codes -> per-window bucket-ordered index lists.
// Kept in its own translation unit: the kernels are field-independent.
#pragma once
#include "common.hpp"

namespace csh {

struct MsmParams {
  uint32_t n;
  int c;        // window bits
  int W;        // windows
  uint32_t NB;  // buckets per window = 2^(c-1), bucket ids 1..NB
  uint32_t L;   // max entries per task
  uint32_t tmax;  // task slots per window
  uint32_t S;     // reduce segments per window (power of two)
  int mont;
  uint32_t CH;         // point chunks per window in the LDS counting sort
  uint32_t chunk_len;  // points per chunk
  // merged-window mode (fixed-base tables, msm_impl.hpp): a sort window holds n = n_points * g codes (g table rows); flat index
  // f = k * remap_n + i is stored as the table index k * remap_stride + remap_off + i. remap_n = 0: identity.
  uint32_t remap_n, remap_stride, remap_off;
  // digit kernel only (grouped tables): digit window w is written to row (w % dig_wp) * dig_g + w / dig_wp, so that the dig_g
  // rows of sort window w' = w % dig_wp are contiguous; rows up to dig_g * dig_wp that no digit window maps to are zero-filled.
  // dig_g <= 1: identity.
  uint32_t dig_g, dig_wp;
  // balanced windows (round 5): the low `wide` windows take c bits, windows wide .. W-1 take c - 1 bits (their digits only reach the lower
  // half of the NB buckets; every array keeps the NB stride). wide == W: uniform c-bit windows. Used by the digit kernel and the
  // Horner fold only -- the sort and bucket stages see windows whose upper buckets happen to be empty.
  int wide;
  // entries per accumulate lane in the narrow windows (w >= wide): their buckets hold twice the entries of a wide window's, so a lane of
  // 2 L entries leaves the same number of partial sums per bucket for k_msm_merge as L does in a wide window. Ln == L: one length.
  uint32_t Ln;
};
// entries per accumulate lane of window w (absolute window index)
__host__ __device__ inline uint32_t lane_len(const MsmParams& p, int w) { return w >= p.wide ? p.Ln : p.L; }

// Digit code (one u16 per point and window): bits 0..14 = bucket-1, bit 15 = negative; 0xFFFF = zero digit.
constexpr uint32_t DIG_ZERO = 0xFFFFu;
constexpr int MAX_WINDOWS = 128;

struct SortBuffers {
  // [W][NB+2] scratch. Contract: CONSUMED AND LEFT DIRTY. k_msm_colscan rewrites every entry on every call (which is why
  // nothing zeroes it between calls: colscan must always run before scan), k_msm_scan turns it into exclusive starts, and the two-level
  // scatter (k_msm_scatter_l2t) then uses it as the buckets' write cursors, advanced with global atomics -- after a call it holds
  // end-of-bucket positions. Consequence of the atomic reservation: the ORDER of the entries inside a bucket, hence the order of the
  // additions in k_msm_accum, varies from run to run (same group element, different device intermediates); the per-partition
  // level 2 (tune msm_variant bit 5) is the deterministic form kept for debugging.
  uint32_t* hist;
  uint32_t* start;     // [W][NB+2] out: first sorted slot per bucket; [NB+1] = entries of the window
  uint32_t* nlanes;    // [MAX_WINDOWS] out: ceil(entries / L) per window
  uint32_t* sorted;    // [W][n] out: (point index | sign << 31) in bucket order
  const uint16_t* dig; // [W][n] in: digit codes
  uint32_t* blkcnt;    // [W][CH][NB] scratch
  uint64_t* inter;     // [W][n] scratch (two-level mode only)
  uint32_t* part_cnt;  // [W][CH][NB/256] scratch (two-level mode only)
};

#if defined(__HIPCC__)
// Wave-aggregated LDS counter increment: returns this lane's slot in counter[b] (old value + rank).
// Lanes that share the wave leader's bucket are peeled off with one atomic per group (up to 4 rounds), so a
// skewed digit distribution (top window, 0/1-heavy witnesses) does not serialise on one LDS address.
__device__ __forceinline__ uint32_t lds_slot(uint32_t* counter, uint32_t b, bool valid) {
  uint32_t slot = 0;
  bool todo = valid;
  for (int round = 0; round < 4; ++round) {
    const unsigned long long act = __ballot(todo);
    if (!act) return slot;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t lb = (uint32_t)__shfl((int)b, leader);
    const unsigned long long grp = __ballot(todo && b == lb);
    const int cnt = __popcll(grp);
    if (cnt < 8) break;  // wave-uniform: not worth peeling, fall through to per-lane atomics
    uint32_t base = 0;
    const int lane = threadIdx.x & 63;
    if (lane == leader) base = atomicAdd(&counter[lb], (uint32_t)cnt);
    base = (uint32_t)__shfl((int)base, leader);
    if (todo && b == lb) {
      slot = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
      todo = false;
    }
  }
  if (todo) slot = atomicAdd(&counter[b], 1u);
  return slot;
}
#endif

bool msm_sort_two_level(const MsmParams& p);
size_t msm_sort_extra_bytes(const MsmParams& p);  // arena bytes for inter + part_cnt (0 in single-level mode)
// Launches hist -> colscan -> scan -> scatter on `st`. ev (nullable): records ev[1] after the histogram, ev[2] after
// the scan, ev[3] after the scatter.
int msm_sort_launch(const MsmParams& p, const SortBuffers& b, hipStream_t st, hipEvent_t* ev);

}  // namespace csh
