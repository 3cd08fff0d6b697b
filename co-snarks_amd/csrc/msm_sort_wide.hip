// Sort stage of the fixed-base MSM with ONE bucket set and windows of 17 .. 22 bits (round 6).
//
// With fixed-base tables of one row per window (table[w][i] = 2^(c w) P_i, msm_impl.hpp) every digit of every scalar lands in the same
// set of NB = 2^(c-1) buckets, so the number of mixed additions per point is W = ceil((bits + 1) / c) with no per-window bucket
// reduction to pay for a wide c: 13 at c = 20 instead of 17 (c = 15, 2^20 points) or 16 (c = 16, 2^24). The plain sort stage
// (msm_sort.hip: 16-bit digit codes, an LDS histogram of all buckets, a one-block scan, 128 level-1 partitions) stops at 2^15 buckets;
// this one sorts n W entries by an up-to-21-bit bucket key in two levels, MSD first:
//
//   level 1  k_wide_hist1     scalars -> signed digits (recomputed, never stored: 32 B per scalar instead of 4 B per digit twice) ->
//                             entries per (chunk of scalars, partition), partition = high bits of (bucket - 1); LDS counters
//            k_wide_colscan   exclusive prefix over the chunks per partition, partition totals
//            k_wide_partscan  partition starts, level-2 job counts (one job = one tile of one partition), totals for the bucket stage
//            k_wide_jobs      job -> partition map
//            k_wide_scatter1  scalars -> digits again; every tile (512 scalars x W digits) is counting-sorted by partition in LDS and
//                             leaves as runs of 8-byte records (table index | sign << 31 | low key bits << 32; 4-byte records when the
//                             ids are short enough) -- chunk-major order inside a partition, no atomics, deterministic
//   level 2  k_wide_hist2     per job: LDS histogram of the low key bits, flushed with one global atomic per non-empty bucket
//            k_wide_scan2     per partition: exclusive scan of its buckets on top of the partition start -> start[], cursors
//            k_wide_scatter2  per job: LDS counting sort by the low key bits, one global atomicAdd per non-empty bucket reserves the
//                             output run (as k_msm_scatter_l2t), coalesced write-out
//
// Output contract = the plain stage's for ONE window: start[NB + 2], nlanes[0], sorted[n W] (table index | sign << 31 in bucket order).
#include <stdlib.h>

#include <type_traits>

#include "field.hpp"
#include "msm_digits.hpp"
#include "msm_sort_wide.hpp"

namespace csh {

constexpr int WS_BLK = 512;    // level 1: one scalar per thread and tile
constexpr int WS_MAXW = 16;    // digits per scalar (c >= 16 and bits + 1 <= 256)
constexpr int W2_BLK = 1024;   // level 2
constexpr int W2_EPT = 8;
constexpr int W2_TILE = W2_BLK * W2_EPT;
constexpr uint32_t WCODE_ZERO = 0xFFFFFFFFu;

// exclusive prefix of one value per thread over a block of NW waves; *total = the block's sum. wsum: >= NW LDS words that nobody else
// touches until the next barrier after the call. One barrier inside.
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t t = wsum[i];
    if (i < wv) base += t;
    tot += t;
  }
  *total = tot;
  return base + incl - v;
}

// Signed-digit codes of scalar i: code[w] = (bucket - 1) | negative << 31, WCODE_ZERO for a zero digit. Same recoding as k_msm_digits /
// for_each_digit (uniform c-bit windows; the table rows are 2^(c w) P). The window width CB is a template parameter: digit w sits at
// the compile-time bit offset w CB, so it is ONE funnel shift of two limbs (v_alignbit_b32) instead of a shift of the whole scalar by a
// run-time count after every digit -- the recoding dropped from ~440 to ~130 instructions per scalar, and it runs twice (k_wide_hist1,
// k_wide_scatter1: 2^24 scalars, level 1 1.53 -> see profiles/r06_* for the measured effect).
template <class Fr, int CB>
struct WideWin {
  static constexpr int W = (Fr::Params::BITS + 1 + CB - 1) / CB;  // = windows_for(bits, CB)
  static_assert(W <= WS_MAXW, "too many digits per scalar");
};
template <class Fr, int CB>
__device__ __forceinline__ void wide_codes(const Fr* __restrict__ scalars, size_t i, bool valid, int mont, uint32_t* code) {
  constexpr int W = WideWin<Fr, CB>::W;
  uint32_t s[Fr::N + 1];
  if (valid) {
    Fr v = scalars[i];
    if (mont) v = v.from_mont();
#pragma unroll
    for (int k = 0; k < Fr::N; ++k) s[k] = v.l[k];
  } else {
#pragma unroll
    for (int k = 0; k < Fr::N; ++k) s[k] = 0;
  }
  s[Fr::N] = 0;
  constexpr uint32_t mask = (1u << CB) - 1, half = 1u << (CB - 1);
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const int bit = w * CB, limb = bit >> 5, off = bit & 31;
    uint32_t raw = 0;
    if (limb < Fr::N) raw = off ? __builtin_amdgcn_alignbit(s[limb + 1], s[limb], (uint32_t)off) : s[limb];
    const uint32_t v = (raw & mask) + carry;
    const uint32_t neg = v > half ? 1u : 0u;
    const uint32_t mag = neg ? (1u << CB) - v : v;  // 0 .. half
    carry = neg;
    code[w] = mag ? ((mag - 1) | (neg << 31)) : WCODE_ZERO;
  }
}

// ---- level 1 ------------------------------------------------------------------------------------------------------------------
template <class Fr, int CB>
__global__ __launch_bounds__(WS_BLK) void k_wide_hist1(const Fr* __restrict__ scalars, MsmParams pd, WidePlan wp, uint32_t* __restrict__ part_cnt,
                                                       uint32_t* __restrict__ cursor, uint32_t cursor_len) {
  constexpr int W = WideWin<Fr, CB>::W;
  extern __shared__ uint32_t wl_cnt[];
  for (uint32_t b = threadIdx.x; b < wp.P; b += WS_BLK) wl_cnt[b] = 0;
  // the bucket counters of level 2 start from zero: every block clears its slice (no memset launch)
  for (size_t b = (size_t)blockIdx.x * WS_BLK + threadIdx.x; b < cursor_len; b += (size_t)gridDim.x * WS_BLK) cursor[b] = 0;
  __syncthreads();
  const uint32_t ch = blockIdx.x;
  uint32_t t1 = (ch + 1) * wp.chunk_tiles;
  if (t1 > wp.n_tiles) t1 = wp.n_tiles;
  for (uint32_t t = ch * wp.chunk_tiles; t < t1; ++t) {
    const size_t i = (size_t)t * WS_BLK + threadIdx.x;
    uint32_t code[W];
    wide_codes<Fr, CB>(scalars, i, i < pd.n, pd.mont, code);
#pragma unroll
    for (int w = 0; w < W; ++w) (void)lds_slot(wl_cnt, (code[w] & 0x7fffffffu) >> wp.lb, code[w] != WCODE_ZERO);
  }
  __syncthreads();
  uint32_t* out = part_cnt + (size_t)ch * wp.P;
  for (uint32_t b = threadIdx.x; b < wp.P; b += WS_BLK) out[b] = wl_cnt[b];
}

// per partition: exclusive prefix over the chunks (in place) and the partition's total. Block = 32 partitions x 32 groups of chunks:
// a thread sums its group's chunks (rows of 32 adjacent partitions: 128-byte loads, eight in flight), the groups' sums are scanned
// through LDS, a second walk writes the prefixes. (First form, one thread per partition walking all CH chunks: 73 us at 2^20 with
// CH = 1024 -- 128 dependent rounds of loads -- for 512 KB of counters, profiles/r06_c_wide_c17_lb9_2p20_kernel_stats.csv.)
__global__ __launch_bounds__(1024) void k_wide_colscan(WidePlan wp, uint32_t* __restrict__ part_cnt, uint32_t* __restrict__ part_tot) {
  __shared__ uint32_t gsum[32][33];
  const uint32_t pl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const uint32_t part = blockIdx.x * 32 + pl;
  const bool live = part < wp.P;
  const uint32_t cpg = (wp.CH + 31) / 32;
  const uint32_t c0 = g * cpg;
  uint32_t c1 = c0 + cpg;
  if (c1 > wp.CH) c1 = wp.CH;
  uint32_t* col = part_cnt + part;
  uint32_t sum = 0;
  if (live) {
    for (uint32_t ch0 = c0; ch0 < c1; ch0 += 8) {
      uint32_t c[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) c[j] = ch0 + j < c1 ? col[(size_t)(ch0 + j) * wp.P] : 0u;
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) sum += c[j];
    }
  }
  gsum[g][pl] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll 8
  for (uint32_t k = 0; k < 32; ++k) {
    const uint32_t t = gsum[k][pl];
    if (k < g) run += t;
    total += t;
  }
  if (!live) return;
  for (uint32_t ch0 = c0; ch0 < c1; ch0 += 8) {
    uint32_t c[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) c[j] = ch0 + j < c1 ? col[(size_t)(ch0 + j) * wp.P] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      if (ch0 + j < c1) col[(size_t)(ch0 + j) * wp.P] = run;
      run += c[j];
    }
  }
  if (g == 0) part_tot[part] = total;
}

// One block: part_start[p] = first intermediate slot of partition p (in: the totals), job_first[p] = first level-2 job of partition p
// (a partition of t entries has ceil(t / W2_TILE) jobs), and the totals the bucket stage reads (start[NB + 1], nlanes[0]).
__global__ __launch_bounds__(1024) void k_wide_partscan(MsmParams p, WidePlan wp, uint32_t* part_start, uint32_t* job_first, uint32_t* start,
                                                        uint32_t* cursor, uint32_t* nlanes) {
  __shared__ uint32_t wsum[16];
  const uint32_t per = (wp.P + 1023) / 1024;  // <= 8
  const uint32_t b0 = threadIdx.x * per;
  uint32_t v[8], sum = 0, jobs = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    v[k] = (k < per && b0 + k < wp.P) ? part_start[b0 + k] : 0u;
    sum += v[k];
    jobs += (v[k] + W2_TILE - 1) / W2_TILE;
  }
  uint32_t total = 0, total_jobs = 0;
  uint32_t run = block_excl_scan<16>(sum, wsum, &total);
  __syncthreads();
  uint32_t jrun = block_excl_scan<16>(jobs, wsum, &total_jobs);
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    if (k < per && b0 + k < wp.P) {
      part_start[b0 + k] = run;
      job_first[b0 + k] = jrun;
      run += v[k];
      jrun += (v[k] + W2_TILE - 1) / W2_TILE;
    }
  }
  if (threadIdx.x == 0) {
    part_start[wp.P] = total;
    job_first[wp.P] = total_jobs;
    start[0] = 0;
    cursor[0] = 0;
    start[p.NB + 1] = total;
    cursor[p.NB + 1] = total;
    nlanes[0] = (total + p.L - 1) / p.L;
  }
}

// job -> partition: the last p with job_first[p] <= j (empty partitions have no jobs and are skipped by the search)
__global__ __launch_bounds__(1024) void k_wide_jobs(WidePlan wp, const uint32_t* __restrict__ job_first, uint32_t* __restrict__ job_part) {
  extern __shared__ uint32_t jf[];  // P + 1
  for (uint32_t b = threadIdx.x; b <= wp.P; b += 1024) jf[b] = job_first[b];
  __syncthreads();
  const uint32_t j = blockIdx.x * 1024 + threadIdx.x;
  if (j >= jf[wp.P]) return;
  uint32_t lo = 0, hi = wp.P;  // first index in [0, P] with jf[idx] > j, minus one
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jf[mid + 1] > j) hi = mid; else lo = mid + 1;
  }
  job_part[j] = lo;
}

template <class Fr, int CB, int REC>
__global__ __launch_bounds__(WS_BLK) void k_wide_scatter1(const Fr* __restrict__ scalars, MsmParams pd, MsmParams p, WidePlan wp,
                                                          const uint32_t* __restrict__ part_pre, const uint32_t* __restrict__ part_start,
                                                          void* __restrict__ inter) {
  using Rec = typename std::conditional<REC != 0, uint32_t, uint64_t>::type;
  extern __shared__ uint32_t wl[];
  const uint32_t P = wp.P, tid = threadIdx.x;
  constexpr int W = WideWin<Fr, CB>::W;
  const uint32_t cap = WS_BLK * (uint32_t)W;
  uint32_t* gcur = wl;          // next free intermediate slot per partition
  uint32_t* cnt = gcur + P;     // tile histogram, then the tile-local exclusive offsets
  uint32_t* wsum = cnt + P;     // 16
  uint32_t* pay = wsum + 16;    // cap
  uint32_t* key = pay + cap;    // cap
  const uint32_t ch = blockIdx.x;
  {
    const uint32_t* pre = part_pre + (size_t)ch * P;
    for (uint32_t b = tid; b < P; b += WS_BLK) {
      gcur[b] = part_start[b] + pre[b];
      cnt[b] = 0;
    }
  }
  __syncthreads();
  const uint32_t per = (P + WS_BLK - 1) / WS_BLK;  // counters per thread in the scan: <= 16
  const uint32_t b0 = tid * per;
  Rec* out = reinterpret_cast<Rec*>(inter);
  const uint32_t lbmask = (1u << wp.lb) - 1;
  uint32_t t1 = (ch + 1) * wp.chunk_tiles;
  if (t1 > wp.n_tiles) t1 = wp.n_tiles;
  for (uint32_t t = ch * wp.chunk_tiles; t < t1; ++t) {
    const size_t i = (size_t)t * WS_BLK + tid;
    uint32_t code[W], rank[W];
    wide_codes<Fr, CB>(scalars, i, i < pd.n, pd.mont, code);
#pragma unroll
    for (int w = 0; w < W; ++w) rank[w] = lds_slot(cnt, (code[w] & 0x7fffffffu) >> wp.lb, code[w] != WCODE_ZERO);
    __syncthreads();
    uint32_t v[16], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      v[k] = (k < per && b0 + k < P) ? cnt[b0 + k] : 0u;
      sum += v[k];
    }
    uint32_t tile_n = 0;
    uint32_t run = block_excl_scan<WS_BLK / 64>(sum, wsum, &tile_n);
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      if (k < per && b0 + k < P) {
        cnt[b0 + k] = run;
        run += v[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < W; ++w) {
      if (code[w] != WCODE_ZERO) {
        const uint32_t k0 = code[w] & 0x7fffffffu;
        const uint32_t slot = cnt[k0 >> wp.lb] + rank[w];
        pay[slot] = ((uint32_t)w * p.remap_stride + p.remap_off + (uint32_t)i) | (code[w] & 0x80000000u);
        key[slot] = k0;
      }
    }
    __syncthreads();
    for (uint32_t sl = tid; sl < tile_n; sl += WS_BLK) {
      const uint32_t k0 = key[sl], bin = k0 >> wp.lb;
      const uint32_t dst = gcur[bin] + (sl - cnt[bin]);
      if constexpr (REC == 1) {
        const uint32_t idb = 31 - wp.lb;
        out[dst] = (pay[sl] & ((1u << idb) - 1)) | ((pay[sl] >> 31) << idb) | ((k0 & lbmask) << (idb + 1));
      } else {
        out[dst] = (uint64_t)pay[sl] | ((uint64_t)(k0 & lbmask) << 32);
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      if (k < per && b0 + k < P) {
        gcur[b0 + k] += v[k];
        cnt[b0 + k] = 0;
      }
    }
    __syncthreads();
  }
}

// ---- level 2 ------------------------------------------------------------------------------------------------------------------
template <int REC>
__device__ __forceinline__ uint32_t rec_sub(typename std::conditional<REC != 0, uint32_t, uint64_t>::type e, uint32_t lb) {
  if constexpr (REC == 1) return e >> (32 - lb);
  else return (uint32_t)(e >> 32);
}
template <int REC>
__device__ __forceinline__ uint32_t rec_entry(typename std::conditional<REC != 0, uint32_t, uint64_t>::type e, uint32_t lb) {
  if constexpr (REC == 1) {
    const uint32_t idb = 31 - lb;
    return (e & ((1u << idb) - 1)) | (((e >> idb) & 1u) << 31);
  } else {
    return (uint32_t)e;
  }
}

struct WideJob {
  uint32_t part, lo, hi;
};
__device__ __forceinline__ WideJob wide_job(const uint32_t* __restrict__ job_first, const uint32_t* __restrict__ job_part,
                                            const uint32_t* __restrict__ part_start, uint32_t P, uint32_t j) {
  WideJob r;
  r.part = 0xffffffffu;
  r.lo = r.hi = 0;
  if (j >= job_first[P]) return r;
  r.part = job_part[j];
  const uint32_t a = part_start[r.part], b = part_start[r.part + 1];
  r.lo = a + (j - job_first[r.part]) * (uint32_t)W2_TILE;
  r.hi = r.lo + (uint32_t)W2_TILE;
  if (r.hi > b) r.hi = b;
  return r;
}

template <int REC>
__global__ __launch_bounds__(W2_BLK) void k_wide_hist2(WidePlan wp, const uint32_t* __restrict__ job_first, const uint32_t* __restrict__ job_part,
                                                       const uint32_t* __restrict__ part_start, const void* __restrict__ inter, uint32_t* cursor) {
  using Rec = typename std::conditional<REC != 0, uint32_t, uint64_t>::type;
  extern __shared__ uint32_t w2_cnt[];  // B2
  const uint32_t tid = threadIdx.x;
  const WideJob jb = wide_job(job_first, job_part, part_start, wp.P, blockIdx.x);
  if (jb.part == 0xffffffffu) return;
  for (uint32_t b = tid; b < wp.B2; b += W2_BLK) w2_cnt[b] = 0;
  __syncthreads();
  const Rec* in = reinterpret_cast<const Rec*>(inter);
  Rec e[W2_EPT];
#pragma unroll
  for (int k = 0; k < W2_EPT; ++k) {
    const uint32_t i = jb.lo + k * W2_BLK + tid;
    e[k] = i < jb.hi ? __builtin_nontemporal_load(in + i) : 0;
  }
#pragma unroll
  for (int k = 0; k < W2_EPT; ++k) (void)lds_slot(w2_cnt, rec_sub<REC>(e[k], wp.lb), jb.lo + k * W2_BLK + tid < jb.hi);
  __syncthreads();
  uint32_t* cur = cursor + 1 + (size_t)jb.part * wp.B2;
  for (uint32_t b = tid; b < wp.B2; b += W2_BLK) {
    const uint32_t v = w2_cnt[b];
    if (v) atomicAdd(&cur[b], v);
  }
}

// Block per partition: bucket counts (in the cursor array) -> start[1 + k] = cursor[1 + k] = first sorted slot of bucket k + 1
__global__ __launch_bounds__(256) void k_wide_scan2(WidePlan wp, const uint32_t* __restrict__ part_start, uint32_t* cursor, uint32_t* start) {
  __shared__ uint32_t wsum[4];
  const uint32_t part = blockIdx.x;
  const uint32_t per = wp.B2 / 256;  // 1 .. 8
  uint32_t* cur = cursor + 1 + (size_t)part * wp.B2 + threadIdx.x * per;
  uint32_t* st = start + 1 + (size_t)part * wp.B2 + threadIdx.x * per;
  uint32_t v[8], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    v[k] = k < per ? cur[k] : 0u;
    sum += v[k];
  }
  uint32_t total;
  uint32_t run = part_start[part] + block_excl_scan<4>(sum, wsum, &total);
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    if (k < per) {
      cur[k] = run;
      st[k] = run;
      run += v[k];
    }
  }
}

template <int REC>
__global__ __launch_bounds__(W2_BLK) void k_wide_scatter2(WidePlan wp, const uint32_t* __restrict__ job_first, const uint32_t* __restrict__ job_part,
                                                          const uint32_t* __restrict__ part_start, const void* __restrict__ inter, uint32_t* cursor,
                                                          uint32_t* __restrict__ sorted) {
  using Rec = typename std::conditional<REC != 0, uint32_t, uint64_t>::type;
  extern __shared__ uint32_t w2[];
  const uint32_t tid = threadIdx.x, B2 = wp.B2;
  uint32_t* cnt = w2;              // B2: tile histogram, then tile-local exclusive offsets
  uint32_t* gbase = cnt + B2;      // B2: reserved output run of every bucket
  uint32_t* wsum = gbase + B2;     // 16
  uint32_t* pay = wsum + 16;       // W2_TILE
  uint16_t* sbin = reinterpret_cast<uint16_t*>(pay + W2_TILE);  // W2_TILE
  const WideJob jb = wide_job(job_first, job_part, part_start, wp.P, blockIdx.x);
  if (jb.part == 0xffffffffu) return;
  for (uint32_t b = tid; b < B2; b += W2_BLK) cnt[b] = 0;
  __syncthreads();
  const Rec* in = reinterpret_cast<const Rec*>(inter);
  Rec e[W2_EPT];
  uint32_t rank[W2_EPT];
#pragma unroll
  for (int k = 0; k < W2_EPT; ++k) {
    const uint32_t i = jb.lo + k * W2_BLK + tid;
    e[k] = i < jb.hi ? __builtin_nontemporal_load(in + i) : 0;
  }
#pragma unroll
  for (int k = 0; k < W2_EPT; ++k) rank[k] = lds_slot(cnt, rec_sub<REC>(e[k], wp.lb), jb.lo + k * W2_BLK + tid < jb.hi);
  __syncthreads();
  {
    uint32_t* cur = cursor + 1 + (size_t)jb.part * B2;
    const uint32_t per = (B2 + W2_BLK - 1) / W2_BLK;  // 1 or 2
    const uint32_t b0 = tid * per;
    uint32_t v[2], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      v[k] = (k < per && b0 + k < B2) ? cnt[b0 + k] : 0u;
      if (v[k]) gbase[b0 + k] = atomicAdd(&cur[b0 + k], v[k]);
      sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_excl_scan<W2_BLK / 64>(sum, wsum, &total);
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      if (k < per && b0 + k < B2) {
        cnt[b0 + k] = run;
        run += v[k];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < W2_EPT; ++k) {
    if (jb.lo + k * W2_BLK + tid < jb.hi) {
      const uint32_t bin = rec_sub<REC>(e[k], wp.lb);
      const uint32_t slot = cnt[bin] + rank[k];
      pay[slot] = rec_entry<REC>(e[k], wp.lb);
      sbin[slot] = (uint16_t)bin;
    }
  }
  __syncthreads();
  const uint32_t tile_n = jb.hi - jb.lo;
#pragma unroll
  for (int k = 0; k < W2_EPT; ++k) {
    const uint32_t sl = k * W2_BLK + tid;
    if (sl < tile_n) {
      const uint32_t bin = sbin[sl];
      sorted[gbase[bin] + (sl - cnt[bin])] = pay[sl];
    }
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
WidePlan msm_wide_plan(const MsmParams& srt, const MsmParams& dig) {
  WidePlan wp;
  const uint32_t K = (uint32_t)srt.c - 1;  // key bits
  int lb = tune().msm_wide_lb.load(std::memory_order_relaxed);
  if (lb < 8 || lb > 11) lb = 10;  // measured: level 1 gains more from few partitions (long runs of records) than level 2 loses (profiles/r06_b .. r06_d)
  while (K - (uint32_t)lb > 13) ++lb;  // at most 8192 partitions
  wp.lb = (uint32_t)lb;
  wp.B2 = 1u << lb;
  wp.P = srt.NB >> lb;
  wp.n_tiles = (dig.n + WS_BLK - 1) / WS_BLK;
  int target = tune().msm_wide_chunks.load(std::memory_order_relaxed);
  if (target < 1 || target > 65536) target = 1024;
  wp.chunk_tiles = (wp.n_tiles + (uint32_t)target - 1) / (uint32_t)target;
  if (wp.chunk_tiles < 1) wp.chunk_tiles = 1;
  wp.CH = (wp.n_tiles + wp.chunk_tiles - 1) / wp.chunk_tiles;
  if (wp.CH < 1) wp.CH = 1;
  wp.jobs_max = (uint32_t)(((uint64_t)srt.n + W2_TILE - 1) / W2_TILE) + wp.P;
  const uint64_t max_id = (uint64_t)dig.W * srt.remap_stride + srt.remap_off;  // stored ids are table indices
  const bool wide_only = (tune().msm_variant.load(std::memory_order_relaxed) & 8) != 0;
  wp.rec4 = (!wide_only && max_id <= (uint64_t(1) << (31 - lb))) ? 1u : 0u;
  return wp;
}

size_t msm_sort_wide_bytes(const MsmParams& srt, const MsmParams& dig) {
  const WidePlan wp = msm_wide_plan(srt, dig);
  const size_t len = (size_t)srt.NB + 2;
  size_t need = 0;
  need += 2 * Arena::padded(sizeof(uint32_t) * len);                       // cursor, start
  need += Arena::padded(sizeof(uint32_t) * MAX_WINDOWS);                   // nlanes
  need += Arena::padded(sizeof(uint32_t) * (size_t)srt.n);                 // sorted
  need += Arena::padded((wp.rec4 ? sizeof(uint32_t) : sizeof(uint64_t)) * (size_t)srt.n);  // intermediate records
  need += Arena::padded(sizeof(uint32_t) * (size_t)wp.CH * wp.P);          // entries per (chunk, partition)
  need += 2 * Arena::padded(sizeof(uint32_t) * ((size_t)wp.P + 1));        // partition starts, first job per partition
  need += Arena::padded(sizeof(uint32_t) * (size_t)wp.jobs_max);           // job -> partition
  return need;
}

template <class Fr, int CB>
static int wide_launch_t(const MsmParams& p, const MsmParams& pd, const uint64_t* scalars_dev, hipStream_t st, Arena& ar, uint32_t** out_start,
                         uint32_t** out_nlanes, uint32_t** out_sorted, hipEvent_t* ev) {
  constexpr int WIN = WideWin<Fr, CB>::W;
  CSH_REQUIRE(pd.W == WIN, "wide sort: the plan's window count does not match the scalar field");
  const WidePlan wp = msm_wide_plan(p, pd);
  const size_t len = (size_t)p.NB + 2;
  uint32_t* cursor = ar.take<uint32_t>(len);
  uint32_t* start = ar.take<uint32_t>(len);
  uint32_t* nlanes = ar.take<uint32_t>(MAX_WINDOWS);
  uint32_t* sorted = ar.take<uint32_t>(p.n);
  void* inter = wp.rec4 ? (void*)ar.take<uint32_t>(p.n) : (void*)ar.take<uint64_t>(p.n);
  uint32_t* part_cnt = ar.take<uint32_t>((size_t)wp.CH * wp.P);
  uint32_t* part_start = ar.take<uint32_t>((size_t)wp.P + 1);
  uint32_t* job_first = ar.take<uint32_t>((size_t)wp.P + 1);
  uint32_t* job_part = ar.take<uint32_t>(wp.jobs_max);
  const Fr* sc = reinterpret_cast<const Fr*>(scalars_dev);
  const size_t lds1 = sizeof(uint32_t) * (2 * (size_t)wp.P + 16 + 2 * (size_t)WS_BLK * pd.W);
  const size_t lds2 = sizeof(uint32_t) * (2 * (size_t)wp.B2 + 16 + W2_TILE) + sizeof(uint16_t) * W2_TILE;
  CSH_TRY(raise_lds_limit((const void*)k_wide_scatter1<Fr, CB, 0>, 160 * 1024));
  CSH_TRY(raise_lds_limit((const void*)k_wide_scatter1<Fr, CB, 1>, 160 * 1024));
  CSH_TRY(raise_lds_limit((const void*)k_wide_scatter2<0>, 160 * 1024));
  CSH_TRY(raise_lds_limit((const void*)k_wide_scatter2<1>, 160 * 1024));
  CSH_REQUIRE(lds1 <= 160 * 1024, "wide sort: level-1 tile does not fit the LDS");
  hipLaunchKernelGGL((k_wide_hist1<Fr, CB>), dim3(wp.CH), dim3(WS_BLK), sizeof(uint32_t) * wp.P, st, sc, pd, wp, part_cnt, cursor, (uint32_t)len);
  hipLaunchKernelGGL(k_wide_colscan, dim3((wp.P + 31) / 32), dim3(1024), 0, st, wp, part_cnt, part_start);
  hipLaunchKernelGGL(k_wide_partscan, dim3(1), dim3(1024), 0, st, p, wp, part_start, job_first, start, cursor, nlanes);
  hipLaunchKernelGGL(k_wide_jobs, dim3((wp.jobs_max + 1023) / 1024), dim3(1024), sizeof(uint32_t) * ((size_t)wp.P + 1), st, wp, job_first, job_part);
  if (wp.rec4) hipLaunchKernelGGL((k_wide_scatter1<Fr, CB, 1>), dim3(wp.CH), dim3(WS_BLK), lds1, st, sc, pd, p, wp, part_cnt, part_start, inter);
  else hipLaunchKernelGGL((k_wide_scatter1<Fr, CB, 0>), dim3(wp.CH), dim3(WS_BLK), lds1, st, sc, pd, p, wp, part_cnt, part_start, inter);
  if (ev) CSH_HIP(hipEventRecord(ev[1], st));
  if (wp.rec4) hipLaunchKernelGGL(k_wide_hist2<1>, dim3(wp.jobs_max), dim3(W2_BLK), sizeof(uint32_t) * wp.B2, st, wp, job_first, job_part, part_start, inter, cursor);
  else hipLaunchKernelGGL(k_wide_hist2<0>, dim3(wp.jobs_max), dim3(W2_BLK), sizeof(uint32_t) * wp.B2, st, wp, job_first, job_part, part_start, inter, cursor);
  hipLaunchKernelGGL(k_wide_scan2, dim3(wp.P), dim3(256), 0, st, wp, part_start, cursor, start);
  if (ev) CSH_HIP(hipEventRecord(ev[2], st));
  if (wp.rec4) hipLaunchKernelGGL(k_wide_scatter2<1>, dim3(wp.jobs_max), dim3(W2_BLK), lds2, st, wp, job_first, job_part, part_start, inter, cursor, sorted);
  else hipLaunchKernelGGL(k_wide_scatter2<0>, dim3(wp.jobs_max), dim3(W2_BLK), lds2, st, wp, job_first, job_part, part_start, inter, cursor, sorted);
  if (ev) CSH_HIP(hipEventRecord(ev[3], st));
  CSH_HIP(hipGetLastError());
  *out_start = start;
  *out_nlanes = nlanes;
  *out_sorted = sorted;
  return CSH_OK;
}

template <class Fr>
static int wide_launch_c(const MsmParams& p, const MsmParams& pd, const uint64_t* scalars_dev, hipStream_t st, Arena& ar, uint32_t** out_start,
                         uint32_t** out_nlanes, uint32_t** out_sorted, hipEvent_t* ev) {
  switch (p.c) {
    case 17: return wide_launch_t<Fr, 17>(p, pd, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
    case 18: return wide_launch_t<Fr, 18>(p, pd, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
    case 19: return wide_launch_t<Fr, 19>(p, pd, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
    case 20: return wide_launch_t<Fr, 20>(p, pd, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
    case 21: return wide_launch_t<Fr, 21>(p, pd, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
    case 22: return wide_launch_t<Fr, 22>(p, pd, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
  }
  set_error("wide sort: window width %d out of range", p.c);
  return CSH_ERR_INVALID;
}

int msm_sort_wide_launch(int fr_id, const MsmParams& srt, const MsmParams& dig, const uint64_t* scalars_dev, hipStream_t st, Arena& ar,
                         uint32_t** out_start, uint32_t** out_nlanes, uint32_t** out_sorted, hipEvent_t* ev) {
  CSH_REQUIRE(srt.W == 1 && srt.c >= 17 && srt.c <= 22 && dig.W >= 1 && dig.W <= WS_MAXW && srt.remap_n == dig.n, "wide sort: bad plan");
  if (fr_id == 1) return wide_launch_c<Bls381Fr>(srt, dig, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
  if (fr_id == 2) return wide_launch_c<Bn254Fq>(srt, dig, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
  if (fr_id == 3) return wide_launch_c<Bls377Fr>(srt, dig, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
  return wide_launch_c<Bn254Fr>(srt, dig, scalars_dev, st, ar, out_start, out_nlanes, out_sorted, ev);
}

}  // namespace csh
