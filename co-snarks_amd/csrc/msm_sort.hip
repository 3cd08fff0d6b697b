// Sort stage of the MSM: LDS counting sort of the signed-digit codes (see msm.hip for the pipeline).
#include <stdlib.h>

#include <type_traits>

#include "msm_sort.hpp"

namespace csh {

constexpr int SORT_BLK = 1024;

// index written into the sorted list for flat position i of the digit array (identity unless merged-window mode)
__device__ __forceinline__ uint32_t entry_id(const MsmParams& p, size_t i) {
  if (p.remap_n == 0) return (uint32_t)i;
  const uint32_t f = (uint32_t)i;
  const uint32_t w = f / p.remap_n;
  return w * p.remap_stride + p.remap_off + (f - w * p.remap_n);
}

// Block (chunk ch, window w): LDS histogram of the chunk's digits -> blkcnt[w][ch][0..NB)
__global__ __launch_bounds__(SORT_BLK) void k_msm_hist_lds(MsmParams p, const uint16_t* __restrict__ dig, uint32_t* __restrict__ blkcnt,
                                                            uint32_t* __restrict__ part_cnt /* two-level mode: [w][ch][NB/256], else NULL */) {
  extern __shared__ uint32_t lds_cnt[];
  const uint32_t ch = blockIdx.x, w = blockIdx.y;
  for (uint32_t b = threadIdx.x; b < p.NB; b += SORT_BLK) lds_cnt[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)ch * p.chunk_len;
  size_t hi = lo + p.chunk_len;
  if (hi > p.n) hi = p.n;
  const uint16_t* d = dig + (size_t)w * p.n;
  for (size_t i0 = lo; i0 < hi; i0 += 4 * SORT_BLK) {
    uint32_t code[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t i = i0 + (size_t)k * SORT_BLK + threadIdx.x;
      code[k] = i < hi ? (uint32_t)d[i] : DIG_ZERO;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) (void)lds_slot(lds_cnt, code[k] & 0x7fffu, code[k] != DIG_ZERO);
  }
  __syncthreads();
  uint32_t* out = blkcnt + ((size_t)w * p.CH + ch) * p.NB;
  for (uint32_t b = threadIdx.x; b < p.NB; b += SORT_BLK) out[b] = lds_cnt[b];
  if (part_cnt) {  // entries of this chunk per partition of 256 adjacent buckets, straight from the LDS histogram (one wave each)
    const uint32_t P = p.NB / 256;
    const int lane = threadIdx.x & 63;
    for (uint32_t part = threadIdx.x >> 6; part < P; part += SORT_BLK / 64) {
      const uint32_t* c = lds_cnt + part * 256;
      uint32_t v = c[lane] + c[lane + 64] + c[lane + 128] + c[lane + 192];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
      if (lane == 0) part_cnt[((size_t)w * p.CH + ch) * P + part] = v;
    }
  }
}

// Per (window, bucket): exclusive prefix over chunks (in place) and the bucket total -> hist[w][b+1]
__global__ __launch_bounds__(256) void k_msm_colscan(MsmParams p, uint32_t* __restrict__ blkcnt, uint32_t* __restrict__ hist) {
  const uint32_t w = blockIdx.y;
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= p.NB) return;
  // eight chunks per round: the loads of a round are issued together (a one-at-a-time walk paid one memory latency per chunk: 14 us)
  uint32_t acc = 0;
  uint32_t* col = blkcnt + (size_t)w * p.CH * p.NB + b;
  for (uint32_t ch0 = 0; ch0 < p.CH; ch0 += 8) {
    uint32_t t[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) t[j] = ch0 + j < p.CH ? col[(size_t)(ch0 + j) * p.NB] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      if (ch0 + j < p.CH) col[(size_t)(ch0 + j) * p.NB] = acc;
      acc += t[j];
    }
  }
  hist[(size_t)w * (p.NB + 2) + b + 1] = acc;
  if (b == 0) {  // the two border entries the scan reads as zero (no memset of the histogram per call)
    hist[(size_t)w * (p.NB + 2)] = 0;
    hist[(size_t)w * (p.NB + 2) + p.NB + 1] = 0;
  }
}

// One 1024-thread block per window. In: hist[w][0..NB+1] counts (index 0 and NB+1 unused = 0).
// Out: start[w][b] = first sorted slot of bucket b (start[w][NB+1] = total entries of the window) and
// nlanes[w] = ceil(total / L): the accumulate kernel cuts the sorted array into equal runs of L entries, one lane
// each, so every lane of a wave does the same number of mixed additions whatever the bucket sizes are.
// The counts are staged in LDS (coalesced loads / stores; each thread then walks its own contiguous run of counters in LDS, where
// the odd run length keeps the lanes on different banks) and the 1024 per-thread totals are scanned with wave shuffles + one
// LDS hop: 52 us -> ~10 us per launch at NB = 2^14 against the former strided global walk + 20-barrier Hillis-Steele scan.
__global__ __launch_bounds__(1024) void k_msm_scan(MsmParams p, uint32_t* hist, uint32_t* start, uint32_t* nlanes) {
  extern __shared__ uint32_t sc_lds[];  // len counters, then 16 wave totals
  const int w = blockIdx.x;
  const uint32_t len = p.NB + 2;
  uint32_t* h = hist + (size_t)w * len;
  uint32_t* st = start + (size_t)w * len;
  uint32_t* wave_tot = sc_lds + len;
  for (uint32_t b = threadIdx.x; b < len; b += 1024) sc_lds[b] = h[b];
  __syncthreads();
  const uint32_t per = (len + 1023) / 1024;
  const uint32_t b0 = threadIdx.x * per;
  uint32_t cnt = 0;
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t b = b0 + k;
    if (b < len) cnt += sc_lds[b];
  }
  // inclusive scan of cnt over the block: within the wave by shuffles, across the 16 waves through LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t t = wave_tot[i];
    if (i < wv) base += t;
    total += t;
  }
  uint32_t run_c = base + incl - cnt;  // exclusive prefix of this thread's run
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t b = b0 + k;
    if (b < len) {
      const uint32_t cv = sc_lds[b];
      sc_lds[b] = run_c;
      run_c += cv;
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < len; b += 1024) {
    st[b] = sc_lds[b];
    h[b] = sc_lds[b];  // the same offsets again, as the global bucket cursors of the tile-parallel level-2 scatter (consumed by atomicAdd)
  }
  if (threadIdx.x == 0) nlanes[w] = (total + lane_len(p, w) - 1) / lane_len(p, w);
}

// Block (chunk ch, window w): LDS cursors = bucket start + this chunk's prefix; scatter (index | sign) into
// bucket order. No global atomics; the order inside a bucket is deterministic per chunk.
__global__ __launch_bounds__(SORT_BLK) void k_msm_scatter_lds(MsmParams p, const uint16_t* __restrict__ dig,
                                                               const uint32_t* __restrict__ start, const uint32_t* __restrict__ blkcnt,
                                                               uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lds_cur[];
  const uint32_t ch = blockIdx.x, w = blockIdx.y;
  const uint32_t* st = start + (size_t)w * (p.NB + 2) + 1;
  const uint32_t* pre = blkcnt + ((size_t)w * p.CH + ch) * p.NB;
  for (uint32_t b = threadIdx.x; b < p.NB; b += SORT_BLK) lds_cur[b] = st[b] + pre[b];
  __syncthreads();
  const size_t lo = (size_t)ch * p.chunk_len;
  size_t hi = lo + p.chunk_len;
  if (hi > p.n) hi = p.n;
  const uint16_t* d = dig + (size_t)w * p.n;
  uint32_t* so = sorted + (size_t)w * p.n;
  for (size_t i0 = lo; i0 < hi; i0 += SORT_BLK) {
    const size_t i = i0 + threadIdx.x;
    uint32_t code = DIG_ZERO;
    if (i < hi) code = d[i];
    const bool valid = code != DIG_ZERO;
    const uint32_t pos = lds_slot(lds_cur, code & 0x7fffu, valid);
    if (valid) so[pos] = entry_id(p, i) | ((code >> 15) << 31);
  }
}


// ---- two-level scatter (large n): write-combining friendly ------------------------------------------------------
// A single-level scatter keeps 2^(c-1) open 4-byte write streams per block (HBM sees mostly partial-line writes:
// 6.2 ms at n = 2^24). Level 1 splits a chunk's entries into P = NB/256 partitions of 256 adjacent buckets (128 open
// streams of 8-byte records, consecutive positions -> full lines); level 2 sorts each partition by the low bucket
// byte with 256 open streams inside one contiguous output slice. All offsets come from the histogram already built.
constexpr uint32_t PART_BUCKETS = 256;
constexpr int SORT_UNROLL = 4;

// in place: part_cnt[w][ch][p] -> first intermediate slot of (chunk ch, partition p) = start of the partition's first
// bucket + entries of earlier chunks. (Folding this into k_msm_scan, 64 threads per window walking the chunks, was measured:
// 24 us against 7 + 10 us as two launches.)
__global__ __launch_bounds__(256) void k_msm_part_offsets(MsmParams p, const uint32_t* __restrict__ start, uint32_t* part_cnt) {
  const uint32_t P = p.NB / PART_BUCKETS;
  const uint32_t part = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
  if (part >= P) return;
  uint32_t run = start[(size_t)w * (p.NB + 2) + (size_t)part * PART_BUCKETS + 1];
  uint32_t* col = part_cnt + (size_t)w * p.CH * P + part;
  for (uint32_t ch0 = 0; ch0 < p.CH; ch0 += 8) {  // loads of eight chunks in flight together, as in k_msm_colscan
    uint32_t c[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) c[j] = ch0 + j < p.CH ? col[(size_t)(ch0 + j) * P] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      if (ch0 + j < p.CH) col[(size_t)(ch0 + j) * P] = run;
      run += c[j];
    }
  }
}

// Level 1: block (chunk, window); tiles of L1_TILE digit codes are counting-sorted by partition inside LDS and
// written out as 8-byte records (index | sign << 31 | low bucket byte << 32) in runs of neighbouring addresses.
// REC = 1 (entry ids < 2^23, i.e. n <= 2^23 and no table remap): 4-byte records (index | sign << 23 | low bucket byte << 24)
// -- 14 instead of 22 bytes of HBM traffic per entry over the two levels. REC = 0: the 8-byte records.
#ifndef CSH_L1_WPE
#define CSH_L1_WPE 8  // two 1024-lane blocks per CU (64 VGPRs instead of 75): one block's barriers and LDS phases under the other's loads;
                      // scatter stage 0.124 -> 0.110 ms at 2^20, 0.384 -> 0.351 at 2^22 (profiles/archive/r03_y_scatter_l1_occupancy.log)
#endif
#if CSH_L1_WPE > 0
#define CSH_L1_OCC __attribute__((amdgpu_waves_per_eu(CSH_L1_WPE)))
#else
#define CSH_L1_OCC
#endif
constexpr int L1_EPT = 8;
constexpr int L1_TILE = L1_EPT * SORT_BLK;
template <int REC>
__global__ __launch_bounds__(SORT_BLK) CSH_L1_OCC void k_msm_scatter_l1(MsmParams p, const uint16_t* __restrict__ dig, const uint32_t* __restrict__ part_off,
                                                             void* __restrict__ inter) {
  using Rec = typename std::conditional<REC != 0, uint32_t, uint64_t>::type;
  constexpr uint32_t MAXP = 128;  // NB <= 2^15
  __shared__ uint32_t gcur[MAXP];
  __shared__ uint32_t cnt[MAXP];
  __shared__ uint32_t toff[MAXP];
  __shared__ uint32_t wsum[2];
  __shared__ uint32_t pay[L1_TILE];
  __shared__ uint8_t slo[L1_TILE];
  __shared__ uint8_t sbin[L1_TILE];
  const uint32_t ch = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  const uint32_t P = p.NB / PART_BUCKETS;
  if (tid < MAXP) {
    gcur[tid] = tid < P ? part_off[((size_t)w * p.CH + ch) * P + tid] : 0;
    cnt[tid] = 0;
  }
  __syncthreads();
  const size_t lo = (size_t)ch * p.chunk_len;
  size_t hi = lo + p.chunk_len;
  if (hi > p.n) hi = p.n;
  const uint16_t* d = dig + (size_t)w * p.n;
  Rec* out = reinterpret_cast<Rec*>(inter) + (size_t)w * p.n;
  for (size_t t0 = lo; t0 < hi; t0 += L1_TILE) {
    uint32_t code[L1_EPT], rank[L1_EPT];
#pragma unroll
    for (int k = 0; k < L1_EPT; ++k) {
      const size_t i = t0 + (size_t)k * SORT_BLK + tid;
      code[k] = i < hi ? (uint32_t)__builtin_nontemporal_load(d + i) : DIG_ZERO;
    }
#pragma unroll
    for (int k = 0; k < L1_EPT; ++k) rank[k] = lds_slot(cnt, (code[k] & 0x7fffu) / PART_BUCKETS, code[k] != DIG_ZERO);
    __syncthreads();
    uint32_t v = 0, incl = 0;
    if (tid < MAXP) {
      v = cnt[tid];
      incl = v;
      const int lane = tid & 63;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, dd);
        if (lane >= dd) incl += t;
      }
      if (lane == 63) wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < MAXP) toff[tid] = (tid >= 64 ? wsum[0] : 0) + incl - v;
    __syncthreads();
    const uint32_t tile_n = wsum[0] + wsum[1];  // non-zero digits in this tile
#pragma unroll
    for (int k = 0; k < L1_EPT; ++k) {
      if (code[k] != DIG_ZERO) {
        const uint32_t b0 = code[k] & 0x7fffu;
        const uint32_t slot = toff[b0 / PART_BUCKETS] + rank[k];
        pay[slot] = entry_id(p, t0 + (size_t)k * SORT_BLK + tid) | ((code[k] >> 15) << 31);
        slo[slot] = (uint8_t)(b0 % PART_BUCKETS);
        sbin[slot] = (uint8_t)(b0 / PART_BUCKETS);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < L1_EPT; ++k) {
      const uint32_t sl = k * SORT_BLK + tid;
      if (sl < tile_n) {
        const uint32_t bin = sbin[sl];
        const uint32_t dst = gcur[bin] + (sl - toff[bin]);
        if constexpr (REC == 1) {
          out[dst] = (pay[sl] & 0x7fffffu) | ((pay[sl] >> 31) << 23) | ((uint32_t)slo[sl] << 24);
        } else {
          out[dst] = (uint64_t)pay[sl] | ((uint64_t)slo[sl] << 32);
        }
      }
    }
    __syncthreads();
    if (tid < MAXP) {
      gcur[tid] += v;
      cnt[tid] = 0;
    }
    __syncthreads();
  }
}

// Level 2: block (partition, window). The partition is processed in tiles of L2_TILE records: each tile is
// counting-sorted by the low bucket byte inside LDS and then written out slot by slot, so neighbouring lanes write
// neighbouring addresses (runs of ~L2_TILE/256 entries per bucket) instead of 64 unrelated 4-byte stores per wave.
constexpr int L2_EPT = 8;
constexpr int L2_TILE = L2_EPT * SORT_BLK;
template <int REC>
__global__ __launch_bounds__(SORT_BLK) void k_msm_scatter_l2(MsmParams p, const uint32_t* __restrict__ start, const void* __restrict__ inter,
                                                             uint32_t* __restrict__ sorted) {
  using Rec = typename std::conditional<REC != 0, uint32_t, uint64_t>::type;
  constexpr int BIN_SHIFT = REC != 0 ? 24 : 32;
  __shared__ uint32_t gcur[PART_BUCKETS];  // next free sorted slot per bucket
  __shared__ uint32_t cnt[PART_BUCKETS];   // tile histogram
  __shared__ uint32_t toff[PART_BUCKETS];  // tile-local exclusive offsets
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t pay[L2_TILE];
  __shared__ uint8_t sbin[L2_TILE];
  const uint32_t part = blockIdx.x, w = blockIdx.y;
  const uint32_t* st = start + (size_t)w * (p.NB + 2) + (size_t)part * PART_BUCKETS + 1;
  const uint32_t tid = threadIdx.x;
  if (tid < PART_BUCKETS) {
    gcur[tid] = st[tid];
    cnt[tid] = 0;
  }
  __syncthreads();
  const uint32_t lo = st[0], hi = st[PART_BUCKETS];
  const Rec* in = reinterpret_cast<const Rec*>(inter) + (size_t)w * p.n;
  uint32_t* so = sorted + (size_t)w * p.n;
  for (uint32_t t0 = lo; t0 < hi; t0 += L2_TILE) {
    Rec e[L2_EPT];
    uint32_t rank[L2_EPT];
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) {
      const uint32_t i = t0 + k * SORT_BLK + tid;
      e[k] = i < hi ? __builtin_nontemporal_load(in + i) : 0;
    }
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) rank[k] = lds_slot(cnt, (uint32_t)(e[k] >> BIN_SHIFT), t0 + k * SORT_BLK + tid < hi);
    __syncthreads();
    uint32_t v = 0, incl = 0;
    if (tid < PART_BUCKETS) {  // waves 0..3, fully active: wave scan + 4 wave totals
      v = cnt[tid];
      incl = v;
      const int lane = tid & 63;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
      }
      if (lane == 63) wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < PART_BUCKETS) {
      uint32_t base = 0;
      for (uint32_t q = 0; q < (tid >> 6); ++q) base += wsum[q];
      toff[tid] = base + incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) {
      if (t0 + k * SORT_BLK + tid < hi) {
        const uint32_t bin = (uint32_t)(e[k] >> BIN_SHIFT);
        const uint32_t slot = toff[bin] + rank[k];
        if constexpr (REC == 1) pay[slot] = ((uint32_t)e[k] & 0x7fffffu) | ((((uint32_t)e[k] >> 23) & 1u) << 31);
        else pay[slot] = (uint32_t)e[k];
        sbin[slot] = (uint8_t)bin;
      }
    }
    __syncthreads();
    const uint32_t tile_n = hi - t0 < (uint32_t)L2_TILE ? hi - t0 : (uint32_t)L2_TILE;
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) {
      const uint32_t sl = k * SORT_BLK + tid;
      if (sl < tile_n) {
        const uint32_t bin = sbin[sl];
        so[gcur[bin] + (sl - toff[bin])] = pay[sl];
      }
    }
    __syncthreads();
    if (tid < PART_BUCKETS) {
      gcur[tid] += v;
      cnt[tid] = 0;
    }
    __syncthreads();
  }
}

// Level 2, tile-parallel (the default since the end of round 3). The kernel above gives one block a whole partition, so a skewed
// digit distribution -- a 0/1-heavy witness puts every "1" into bucket 1 of window 0, a repeated value one bucket per window -- leaves
// ONE block to sort a quarter or half of a window's entries tile after tile while the rest of the chip is idle (scatter 0.11 -> 0.42 ms
// at 2^20 with half the scalars in {0, 1} and a quarter equal). Here every partition is cut into slices of about one tile, one block
// each: the block finds its (partition, slice) from the partitions' sizes (a 128-entry scan of `start`), counting-sorts its tiles by the
// low bucket byte in LDS as before, and reserves the output run of every bucket with ONE global atomicAdd per bucket and tile on the
// cursor array k_msm_scan left in `hist`. The order of the entries inside a bucket then depends on the order in which blocks reserve --
// the bucket sums are sums in a group, the results are bit-identical (parity suite), and `start` is untouched.
template <int REC>
__global__ __launch_bounds__(SORT_BLK) void k_msm_scatter_l2t(MsmParams p, const uint32_t* __restrict__ start, uint32_t* __restrict__ cursor,
                                                              const void* __restrict__ inter, uint32_t* __restrict__ sorted) {
  using Rec = typename std::conditional<REC != 0, uint32_t, uint64_t>::type;
  constexpr int BIN_SHIFT = REC != 0 ? 24 : 32;
  constexpr uint32_t MAXP = 128;  // NB <= 2^15
  __shared__ uint32_t gbase[PART_BUCKETS];  // reserved output run of every bucket for the current tile
  __shared__ uint32_t cnt[PART_BUCKETS];    // tile histogram
  __shared__ uint32_t toff[PART_BUCKETS];   // tile-local exclusive offsets
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t job[3];               // partition, first and one-past-last intermediate slot of this block's slice
  __shared__ uint32_t pay[L2_TILE];
  __shared__ uint8_t sbin[L2_TILE];
  const uint32_t x = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  const uint32_t P = p.NB / PART_BUCKETS;
  const uint32_t* stw = start + (size_t)w * (p.NB + 2);
  {  // which slice of which partition is block x: partitions get max(1, round(entries / tile)) blocks each (none when empty)
    uint32_t lo_p = 0, hi_p = 0, nb = 0, incl = 0;
    if (tid < MAXP) {
      if (tid < P) {
        lo_p = stw[(size_t)tid * PART_BUCKETS + 1];
        hi_p = stw[(size_t)(tid + 1) * PART_BUCKETS + 1];
        const uint32_t c = hi_p - lo_p;
        nb = c ? (c + L2_TILE / 2) / L2_TILE : 0;
        if (c && nb == 0) nb = 1;
      }
      incl = nb;
      const int lane = tid & 63;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
      }
      if (lane == 63) wsum[tid >> 6] = incl;
    }
    if (tid == 0) job[0] = 0xffffffffu;
    __syncthreads();
    if (tid < MAXP) {
      const uint32_t pre = (tid >= 64 ? wsum[0] : 0) + incl - nb;  // blocks of earlier partitions
      if (nb && x >= pre && x < pre + nb) {
        const uint32_t c = hi_p - lo_p, slice = (c + nb - 1) / nb, j = x - pre;
        uint32_t a = lo_p + j * slice, b = a + slice;
        if (b > hi_p) b = hi_p;
        if (a > hi_p) a = hi_p;
        job[0] = tid;
        job[1] = a;
        job[2] = b;
      }
    }
    __syncthreads();
  }
  const uint32_t part = job[0];
  if (part == 0xffffffffu) return;  // the grid is sized for the worst case (entries / tile + partitions blocks per window)
  const uint32_t lo = job[1], hi = job[2];
  if (tid < PART_BUCKETS) cnt[tid] = 0;
  __syncthreads();
  uint32_t* cur = cursor + (size_t)w * (p.NB + 2) + (size_t)part * PART_BUCKETS + 1;
  const Rec* in = reinterpret_cast<const Rec*>(inter) + (size_t)w * p.n;
  uint32_t* so = sorted + (size_t)w * p.n;
  for (uint32_t t0 = lo; t0 < hi; t0 += L2_TILE) {
    Rec e[L2_EPT];
    uint32_t rank[L2_EPT];
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) {
      const uint32_t i = t0 + k * SORT_BLK + tid;
      e[k] = i < hi ? __builtin_nontemporal_load(in + i) : 0;
    }
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) rank[k] = lds_slot(cnt, (uint32_t)(e[k] >> BIN_SHIFT), t0 + k * SORT_BLK + tid < hi);
    __syncthreads();
    uint32_t v = 0, incl = 0;
    if (tid < PART_BUCKETS) {  // waves 0..3, fully active: wave scan + 4 wave totals; and the output run of every bucket of the tile
      v = cnt[tid];
      gbase[tid] = v ? atomicAdd(&cur[tid], v) : 0u;
      incl = v;
      const int lane = tid & 63;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
      }
      if (lane == 63) wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < PART_BUCKETS) {
      uint32_t base = 0;
      for (uint32_t q = 0; q < (tid >> 6); ++q) base += wsum[q];
      toff[tid] = base + incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) {
      if (t0 + k * SORT_BLK + tid < hi) {
        const uint32_t bin = (uint32_t)(e[k] >> BIN_SHIFT);
        const uint32_t slot = toff[bin] + rank[k];
        if constexpr (REC == 1) pay[slot] = ((uint32_t)e[k] & 0x7fffffu) | ((((uint32_t)e[k] >> 23) & 1u) << 31);
        else pay[slot] = (uint32_t)e[k];
        sbin[slot] = (uint8_t)bin;
      }
    }
    __syncthreads();
    const uint32_t tile_n = hi - t0 < (uint32_t)L2_TILE ? hi - t0 : (uint32_t)L2_TILE;
#pragma unroll
    for (int k = 0; k < L2_EPT; ++k) {
      const uint32_t sl = k * SORT_BLK + tid;
      if (sl < tile_n) {
        const uint32_t bin = sbin[sl];
        so[gbase[bin] + (sl - toff[bin])] = pay[sl];
      }
    }
    __syncthreads();
    if (tid < PART_BUCKETS) cnt[tid] = 0;
    __syncthreads();
  }
}

// Two-level mode pays off once a (partition, window) block has enough records to fill its tiles: n >= 2^20
// (measured: 2^20 0.22 -> 0.12 ms, 2^24 6.2 -> 2.1 ms; slower at 2^18). csh_tune_set("sort_two_level", 0/1) forces a mode (tests).
bool msm_sort_two_level(const MsmParams& p) {
  if (p.NB < 4 * PART_BUCKETS) return false;
  if (const int f = tune().sort_two_level.load(std::memory_order_relaxed); f >= 0) return f != 0;
  return p.n >= (1u << 20);
}

size_t msm_sort_extra_bytes(const MsmParams& p) {
  if (!msm_sort_two_level(p)) return 0;
  return Arena::padded(sizeof(uint64_t) * (size_t)p.n * p.W) + Arena::padded(sizeof(uint32_t) * (size_t)(p.NB / PART_BUCKETS) * p.CH * p.W);
}

int msm_sort_launch(const MsmParams& p, const SortBuffers& b, hipStream_t st, hipEvent_t* ev) {
  const bool two_level = msm_sort_two_level(p);
  const uint32_t nparts = p.NB / PART_BUCKETS;
  const size_t sort_lds = sizeof(uint32_t) * p.NB;
  if (sort_lds > 48 * 1024) {
    CSH_TRY(raise_lds_limit((const void*)k_msm_hist_lds, 128 * 1024));
    CSH_TRY(raise_lds_limit((const void*)k_msm_scatter_lds, 128 * 1024));
  }
  hipLaunchKernelGGL(k_msm_hist_lds, dim3(p.CH, p.W), dim3(SORT_BLK), sort_lds, st, p, b.dig, b.blkcnt, two_level ? b.part_cnt : nullptr);
  hipLaunchKernelGGL(k_msm_colscan, dim3((p.NB + 255) / 256, p.W), dim3(256), 0, st, p, b.blkcnt, b.hist);
  if (ev) CSH_HIP(hipEventRecord(ev[1], st));
  {
    const size_t scan_lds = sizeof(uint32_t) * ((size_t)p.NB + 2 + 16);
    if (scan_lds > 48 * 1024) CSH_TRY(raise_lds_limit((const void*)k_msm_scan, 136 * 1024));
    hipLaunchKernelGGL(k_msm_scan, dim3(p.W), dim3(1024), scan_lds, st, p, b.hist, b.start, b.nlanes);
  }
  if (ev) CSH_HIP(hipEventRecord(ev[2], st));
  if (two_level) {
    hipLaunchKernelGGL(k_msm_part_offsets, dim3((nparts + 255) / 256, p.W), dim3(256), 0, st, p, b.start, b.part_cnt);
    // 4-byte intermediate records when every stored entry id fits 23 bits (n, or rows x bases with tables, <= 2^23); tune "msm_variant"
    // bit 3 forces the 8-byte records (A/B runs, tests). (4-byte records + a separate sign byte for ids up to 2^24 were
    // measured and lose to the 8-byte records: scatter 2.29 against 2.10 ms on BN254 G1 2^24 -- byte-granular scattered
    // writes cost more than the 3 bytes per entry they save, profiles/archive/r02_g_rec_stages.log.)
    const uint64_t max_id = p.remap_n ? (uint64_t)(p.n / p.remap_n) * p.remap_stride : (uint64_t)p.n;  // stored ids: table indices
    const bool wide_only = (tune().msm_variant.load(std::memory_order_relaxed) & 8) != 0;
    // level 2: one block per slice of about one tile of a partition (default), or -- tune "msm_variant" bit 5 -- one block per partition
    const bool by_partition = (tune().msm_variant.load(std::memory_order_relaxed) & 32) != 0;
    const uint32_t l2_blocks = (uint32_t)(p.n / L2_TILE) + nparts + 1;
    if (!wide_only && max_id <= (1u << 23)) {
      hipLaunchKernelGGL(k_msm_scatter_l1<1>, dim3(p.CH, p.W), dim3(SORT_BLK), 0, st, p, b.dig, b.part_cnt, (void*)b.inter);
      if (by_partition) hipLaunchKernelGGL(k_msm_scatter_l2<1>, dim3(nparts, p.W), dim3(SORT_BLK), 0, st, p, b.start, (const void*)b.inter, b.sorted);
      else hipLaunchKernelGGL(k_msm_scatter_l2t<1>, dim3(l2_blocks, p.W), dim3(SORT_BLK), 0, st, p, b.start, b.hist, (const void*)b.inter, b.sorted);
    } else {
      hipLaunchKernelGGL(k_msm_scatter_l1<0>, dim3(p.CH, p.W), dim3(SORT_BLK), 0, st, p, b.dig, b.part_cnt, (void*)b.inter);
      if (by_partition) hipLaunchKernelGGL(k_msm_scatter_l2<0>, dim3(nparts, p.W), dim3(SORT_BLK), 0, st, p, b.start, (const void*)b.inter, b.sorted);
      else hipLaunchKernelGGL(k_msm_scatter_l2t<0>, dim3(l2_blocks, p.W), dim3(SORT_BLK), 0, st, p, b.start, b.hist, (const void*)b.inter, b.sorted);
    }
  } else {
    hipLaunchKernelGGL(k_msm_scatter_lds, dim3(p.CH, p.W), dim3(SORT_BLK), sort_lds, st, p, b.dig, b.start, b.blkcnt, b.sorted);
  }
  if (ev) CSH_HIP(hipEventRecord(ev[3], st));
  return CSH_OK;
}

}  // namespace csh
