// Templates of the Pippenger MSM (kernels + host drivers), instantiated once per group configuration in its own
// translation unit (msm_inst_*.hip) so the five configurations compile in parallel; msm.hip holds the C ABI and dispatch.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "common.hpp"
#include "curve.hpp"
#include "curve_lazy.hpp"
#include "curve_quad.hpp"
#include "curve_pair.hpp"
#include "host_fp64.hpp"
#include "msm_digits.hpp"
#include "msm_sort.hpp"
#include "msm_sort_wide.hpp"


namespace csh {

// LAZY: bucket accumulation runs in the signed lazy field (field29.hpp); Bases then stores the coordinates
// re-encoded as canonical x*R' (same 32 bytes per coordinate), converted once at upload.
// PAIR (the G2 groups): accumulate and window reduction can run with two lanes per point, one Fp2 component each (curve_pair.hpp)
struct Bn254G1Cfg { using Fq = Bn254Fq;   using Fr = Bn254Fr; static constexpr bool LAZY = true;  using L = Fq29s; static constexpr bool INLINE_ADD = true; static constexpr bool PAIR = false; };
struct Bn254G2Cfg { using Fq = Bn254Fq2;  using Fr = Bn254Fr; static constexpr bool LAZY = true;  using L = Fq29s2; static constexpr bool INLINE_ADD = true; static constexpr bool PAIR = true; static constexpr bool PAIR_ACC_DEFAULT = false; using LP = Fp2Pair<Fq29s>; };
struct Bls381G1Cfg { using Fq = Bls381Fq;  using Fr = Bls381Fr; static constexpr bool LAZY = true;  using L = Fq28s; static constexpr bool INLINE_ADD = true; static constexpr bool PAIR = false; };
// Grumpkin: base field = BN254 Fr, scalars = BN254 Fq (the 2-cycle partner of BN254)
struct GrumpkinG1Cfg { using Fq = Bn254Fr;   using Fr = Bn254Fq; static constexpr bool LAZY = true;  using L = Fr29s; static constexpr bool INLINE_ADD = true; static constexpr bool PAIR = false; };
// BLS12-377 (the curve of the reference's LibSnarkReduction fixtures, co-circom/co-groth16/src/lib.rs:231-300): the 377-bit base field in the
// 14 x 28-bit limbs of BLS12-381's, Fq2 = Fq[u]/(u^2 + 5)
struct Bls377G1Cfg { using Fq = Bls377Fq;  using Fr = Bls377Fr; static constexpr bool LAZY = true;  using L = Fq28s377; static constexpr bool INLINE_ADD = true; static constexpr bool PAIR = false; };
struct Bls377G2Cfg { using Fq = Bls377Fq2; using Fr = Bls377Fr; static constexpr bool LAZY = true;  using L = Fq28s377x2; static constexpr bool INLINE_ADD = true; static constexpr bool PAIR = true; static constexpr bool PAIR_ACC_DEFAULT = true; using LP = Fp2Pair<Fq28s377, 5>; };
struct Bls381G2Cfg { using Fq = Bls381Fq2; using Fr = Bls381Fr; static constexpr bool LAZY = true;  using L = Fq28s2; static constexpr bool INLINE_ADD = true; static constexpr bool PAIR = true; static constexpr bool PAIR_ACC_DEFAULT = true; using LP = Fp2Pair<Fq28s>; };

struct Bases {
  csh_curve_t curve;
  csh_group_t group;
  int device;
  size_t n;
  size_t point_bytes;
  void* points;  // packed Affine<Fq>[n] on the device
  // fixed-base tables (csh_bases_precompute[_grouped]): table[k * n + i] = 2^(table_c * W' * k) * points[i], k < table_W rows
  // (W' = ceil(windows / table_W)), same encoding as `points`. Windows w and w + W' k of an MSM share one set of buckets
  // (merged-window mode); table_W = windows is the full merge (one bucket set, W' = 1).
  void* table = nullptr;
  int table_c = 0, table_W = 0;
};

// MsmParams, the digit-code constants and the sort stage live in msm_sort.hpp / msm_sort.hip

constexpr int MSM_BLK = 256;
#ifndef CSH_ACC_BLK
#define CSH_ACC_BLK 128
#endif
constexpr int ACC_BLK = CSH_ACC_BLK;

// canonical little-endian limbs of scalar i
template <class Fr>
__device__ __forceinline__ void load_scalar(const Fr* __restrict__ scalars, size_t i, int mont, uint32_t* s) {
  Fr v = scalars[i];
  if (mont) v = v.from_mont();
#pragma unroll
  for (int k = 0; k < Fr::N; ++k) s[k] = v.l[k];
}


// Signed-digit codes of every scalar, dig[row(w) * n + i] (row(w) = w, or the grouped-table order (w % W') * g + w / W'). Straight-line
// per window: the scalar is shifted right by c bits after every digit (constant register indices, no indexed register access), the code
// is selected without branches and every row gets exactly one 2-byte store per scalar (rows past W hold no digits). Same recoding
// as for_each_digit (msm_digits.hpp), which the host self-test and the oracle comparison pin. Round 3: stage "digits + histogram" 0.081-0.087 -> 0.071 ms at 2^20, 0.91 -> 0.68 ms at 2^24 (profiles/archive/r03_x_digits_stages.log)
// (the former loop over for_each_digit compiled to 116 basic blocks with indexed register moves and an integer division per row).
template <class Fr>
__global__ __launch_bounds__(MSM_BLK) void k_msm_digits(const Fr* __restrict__ scalars, MsmParams p, uint16_t* __restrict__ dig) {
  const uint32_t g = p.dig_g > 1 ? p.dig_g : 1u, wp = p.dig_g > 1 ? p.dig_wp : (uint32_t)p.W;
  const uint32_t rows = g * wp;  // >= W: the windows past W (grouped tables, g * W' > W) hold no digits
  const uint32_t W = (uint32_t)p.W, wide = (uint32_t)p.wide;
  for (size_t i = blockIdx.x * (size_t)MSM_BLK + threadIdx.x; i < p.n; i += (size_t)gridDim.x * MSM_BLK) {
    uint32_t s[Fr::N];
    load_scalar<Fr>(scalars, i, p.mont, s);
    uint32_t carry = 0, rq = 0, rr = 0;  // w = rq * wp + rr
    for (uint32_t w = 0; w < rows; ++w) {
      uint32_t code = DIG_ZERO;
      if (w < W) {
        const uint32_t c = (uint32_t)p.c - (w >= wide ? 1u : 0u);  // wave-uniform: balanced windows (MsmParams::wide)
        const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
        const uint32_t v = (s[0] & mask) + carry;
#pragma unroll
        for (int k = 0; k + 1 < Fr::N; ++k) s[k] = (s[k] >> c) | (s[k + 1] << (32 - c));
        s[Fr::N - 1] >>= c;
        const uint32_t neg = v > half ? 1u : 0u;
        const uint32_t mag = neg ? (1u << c) - v : v;  // 0 .. half
        carry = neg;
        code = mag ? ((mag - 1) | (neg << 15)) : (uint32_t)DIG_ZERO;
      }
      dig[(size_t)(rr * g + rq) * p.n + i] = (uint16_t)code;
      if (++rr == wp) {
        rr = 0;
        ++rq;
      }
    }
  }
}

// one-word necessary condition for the stored point at infinity (0, 0): the full 16..48-word test only runs behind it
template <class P>
__device__ __forceinline__ uint32_t top_word(const Fp<P>& f) { return f.l[Fp<P>::N - 1]; }
template <class F, int NR>
__device__ __forceinline__ uint32_t top_word(const Fp2T<F, NR>& f) { return top_word(f.c0) | top_word(f.c1); }
template <class Fq>
__device__ __forceinline__ bool stored_is_inf(const Affine<Fq>& pt) {
  return (top_word(pt.x) | top_word(pt.y)) == 0 && pt.is_inf();
}

// first index in [lo, hi) with a[idx] > v
__device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t v) {
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] > v) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// Everything between the accumulate kernel and the final window sums stays in the lazy field: a stored partial /
// bucket sum / segment sum is the XYZZLazy value as the arithmetic left it (no canonical form, no domain change).
template <class Cfg>
using LazyPt = XYZZLazy<typename Cfg::L>;

// Bucket accumulation. Lane k of window w owns sorted entries [k*L, (k+1)*L): it walks them in bucket order and
// emits one partial sum per bucket it touches into slot (bucket + k) -- unique and increasing in (bucket, lane), so
// the partials of one bucket are consecutive slots. A bucket that spans several lanes (skewed scalars, or simply
// > L entries) gets one partial per lane; k_msm_merge folds them into the dense per-bucket array.
template <class Cfg>
__global__ __launch_bounds__(ACC_BLK) void k_msm_accum(const Affine<typename Cfg::Fq>* __restrict__ bases, MsmParams p,
                                                       const uint32_t* __restrict__ start, const uint32_t* __restrict__ nlanes,
                                                       const uint32_t* __restrict__ sorted, LazyPt<Cfg>* partial, uint32_t* giant_count) {
  using Fq = typename Cfg::Fq;
  static_assert(Cfg::LAZY, "the bucket pipeline runs in the signed lazy field");
  const int w = blockIdx.y;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0 && w == 0) {  // the merge kernel's queue of oversized buckets and its count of sliced ones (stream-ordered after this kernel)
    giant_count[0] = 0;
    giant_count[1] = 0;
  }
  if (k >= nlanes[w]) return;
  const uint32_t* st = start + (size_t)w * (p.NB + 2);
  const uint32_t total = st[p.NB + 1];
  const uint32_t Lw = lane_len(p, w);
  const uint32_t lo = k * Lw;
  uint32_t hi = lo + Lw;
  if (hi > total) hi = total;
  uint32_t b = upper_bound_u32(st, 1, p.NB + 1, lo) - 1;  // bucket containing sorted position lo
  uint32_t next = st[b + 1];
  const uint32_t* so = sorted + (size_t)w * p.n;
  LazyPt<Cfg>* pw = partial + (size_t)w * p.tmax;
  {
    using L = typename Cfg::L;
    XYZZLazy<L> acc = XYZZLazy<L>::inf();
    uint32_t e_next = so[lo];
    Affine<Fq> pt_next = bases[e_next & 0x7fffffffu];
    // software prefetch: the gather of the next entry and, on the G1 groups, the sorted index two entries ahead, so that the gather's
    // address never waits for an index load issued in the same iteration: -0.5 .. -1 % on the G1 accumulation, nothing on BN254 G2 and
    // +8 % on BLS12-381 G2, where the extra live register adds spills (profiles/archive/r03_s_acc_prefetch.log, r03_u_acc_prefetch_groups.log)
    constexpr bool INDEX_AHEAD = !Cfg::PAIR;
    uint32_t e_next2 = (INDEX_AHEAD && lo + 1 < hi) ? so[lo + 1] : 0;
    for (uint32_t pos = lo; pos < hi; ++pos) {
      const uint32_t e = e_next;
      const Affine<Fq> pt = pt_next;
      if (pos + 1 < hi) {
        if constexpr (INDEX_AHEAD) {
          e_next = e_next2;
          pt_next = bases[e_next & 0x7fffffffu];
          if (pos + 2 < hi) e_next2 = so[pos + 2];
        } else {
          e_next = so[pos + 1];
          pt_next = bases[e_next & 0x7fffffffu];
        }
      }
      const bool inf = stored_is_inf(pt);
      const L x = L::unpack(pt.x);
      const L y = L::unpack(pt.y).cneg_unpacked(e >> 31);  // limbs stay in [0, 2^B]: lazy_madd subtracts acc.y limb-wise
      if (pos == next) {                            // crossed into the next non-empty bucket: flush, and the entry that
        pw[b + k] = acc;                            // opens the bucket becomes the accumulator (the coordinates of an
        do { ++b; next = st[b + 1]; } while (next <= pos);  // empty accumulator are never read)
        acc.x = x;
        acc.y = y;
        acc.zz = L::one();
        acc.zzz = L::one();
        acc.empty = inf;
        continue;
      }
      if (inf) continue;
      lazy_madd<L, Affine<Fq>>(acc, x, y, bases + (e & 0x7fffffffu), e >> 31);
    }
    pw[b + k] = acc;
  }
}

// The same accumulation with two lanes per run (G2: lane 2k holds the c0 components of the point, lane 2k+1 the c1 components,
// curve_pair.hpp). Everything that steers control flow (run bounds, bucket boundaries, the empty flag, the point-at-infinity and
// doubling tests) is identical on the two lanes of a pair.
template <class Cfg>
__global__ __launch_bounds__(ACC_BLK) void k_msm_accum_pair(const Affine<typename Cfg::Fq>* __restrict__ bases, MsmParams p,
                                                            const uint32_t* __restrict__ start, const uint32_t* __restrict__ nlanes,
                                                            const uint32_t* __restrict__ sorted, LazyPt<Cfg>* partial, uint32_t* giant_count) {
  using Fq = typename Cfg::Fq;
  using L = typename Cfg::LP;
  using LF = typename L::Base;
  using F32 = decltype(bases->x.c0);
  const int w = blockIdx.y;
  const int role = pair_role();
  const uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (blockIdx.x == 0 && threadIdx.x == 0 && w == 0) {
    giant_count[0] = 0;
    giant_count[1] = 0;
  }
  if (k >= nlanes[w]) return;
  const uint32_t* st = start + (size_t)w * (p.NB + 2);
  const uint32_t total = st[p.NB + 1];
  const uint32_t Lw = lane_len(p, w);
  const uint32_t lo = k * Lw;
  uint32_t hi = lo + Lw;
  if (hi > total) hi = total;
  uint32_t b = upper_bound_u32(st, 1, p.NB + 1, lo) - 1;  // bucket containing sorted position lo
  uint32_t next = st[b + 1];
  const uint32_t* so = sorted + (size_t)w * p.n;
  LazyPt<Cfg>* pw = partial + (size_t)w * p.tmax;
  const F32* half = reinterpret_cast<const F32*>(bases) + role;  // component `role` of x at [4 i], of y at [4 i + 2]
  XYZZLazy<L> acc = XYZZLazy<L>::inf();
  uint32_t e_next = so[lo];
  F32 x_next = half[4 * (size_t)(e_next & 0x7fffffffu)], y_next = half[4 * (size_t)(e_next & 0x7fffffffu) + 2];
  for (uint32_t pos = lo; pos < hi; ++pos) {
    const uint32_t e = e_next;
    const F32 xs = x_next, ys = y_next;
    if (pos + 1 < hi) {                           // software prefetch of the next gather
      e_next = so[pos + 1];
      x_next = half[4 * (size_t)(e_next & 0x7fffffffu)];
      y_next = half[4 * (size_t)(e_next & 0x7fffffffu) + 2];
    }
    const bool inf = pair_all((top_word(xs) | top_word(ys)) == 0 && xs.is_zero() && ys.is_zero());  // infinity is stored as (0, 0)
    const L x{LF::unpack(xs)};
    const L y{LF::unpack(ys).cneg_unpacked(e >> 31)};
    if (pos == next) {                            // crossed into the next non-empty bucket: flush; this entry opens the next sum
      pair_store(&pw[b + k], role, acc);
      do { ++b; next = st[b + 1]; } while (next <= pos);
      acc.x = x;
      acc.y = y;
      acc.zz = L::one();
      acc.zzz = L::one();
      acc.empty = inf;
      continue;
    }
    if (inf) continue;
    lazy_madd<L, Affine<Fq>>(acc, x, y, bases + (e & 0x7fffffffu), e >> 31);
  }
  pair_store(&pw[b + k], role, acc);
}

// ---- the latency-bound tail: merge -> reduce -> fold, four lanes per point (curve_quad.hpp) ---------------------------
// A "logical lane" (one bucket / one segment / one tree node) is a DPP quad; blocks of TAIL_BLK threads hold TAIL_BLK / 4 of
// them. Control flow depends on the logical index and on quad-broadcast flags only, so the four lanes stay converged.
constexpr int TAIL_BLK = 64;
constexpr int TAIL_Q = TAIL_BLK / 4;

// Bucket merge: the partials of bucket b sit in consecutive slots b + k0 .. b + k1 (k0, k1 = first / last lane that
// touched it). One quad per bucket folds them into the dense array dense[w][b]; buckets with more than MERGE_CAP
// partials (heavily repeated scalars) are queued for the block-wide tree kernel below. (One LANE per bucket with whole-point additions,
// compiled with the accumulate kernel's pinned multiplier, was measured in round 3: tail 0.34 -> 0.55 ms at 2^20, 0.53 -> 1.05 ms at 2^24,
// profiles/archive/r03_w_merge_lane.log -- 1-3 additions per lane under divergent trip counts and the empty / doubling branches of the full
// addition; not kept.)
constexpr uint32_t MERGE_CAP = 16;
// Oversized buckets of a witness-like scalar vector are few and huge (every "1" lands in bucket 1 of window 0: a quarter of a 2^20 vector
// in {1} = 4766 partials of 55 entries): one block walking such a bucket took 0.5 ms of a 2.2 ms MSM. Buckets with >= GIANT_BIG
// partials are therefore SLICED: their queue entry carries a flag, k_msm_giant_slices sums GIANT_SLICES slices of the partial run with
// one block each into a scratch array, and k_msm_merge_giant folds the slice sums. Up to GIANT_BIG_CAP sliced buckets per launch (more
// cannot hold a large share of the entries each); the rest take the one-block path.
constexpr uint32_t GIANT_BIG = 256, GIANT_SLICES = 16, GIANT_BIG_CAP = 128, GIANT_ROWS = 32, GIANT_SLICED_FLAG = 0x80000000u;
__device__ __forceinline__ void giant_enqueue(uint32_t* giant_count, uint32_t* giant_list, uint32_t* big_list, uint32_t w, uint32_t b, uint32_t nparts) {
  uint32_t tag = w;
  if (nparts >= GIANT_BIG) {
    const uint32_t q = atomicAdd(giant_count + 1, 1u);
    if (q < GIANT_BIG_CAP) {
      big_list[2 * q] = w;
      big_list[2 * q + 1] = b;
      tag |= GIANT_SLICED_FLAG;
    }
  }
  const uint32_t g = atomicAdd(giant_count, 1u);
  giant_list[2 * g] = tag;
  giant_list[2 * g + 1] = b;
}
template <class Cfg>
__global__ __launch_bounds__(TAIL_BLK) void k_msm_merge(MsmParams p, const uint32_t* __restrict__ start,
                                                        const LazyPt<Cfg>* __restrict__ partial, LazyPt<Cfg>* dense,
                                                        uint32_t* giant_count, uint32_t* giant_list, uint32_t* big_list) {
  using L = typename Cfg::L;
  const int w = blockIdx.y;
  const int role = threadIdx.x & 3;
  const uint32_t b = blockIdx.x * TAIL_Q + (threadIdx.x >> 2) + 1;
  if (b > p.NB) return;
  const uint32_t* st = start + (size_t)w * (p.NB + 2);
  const uint32_t lo = st[b], hi = st[b + 1];
  QPt<L> acc = qpt_inf<L>();
  if (hi > lo) {
    const uint32_t Lw = lane_len(p, w);
    const uint32_t k0 = lo / Lw, k1 = (hi - 1) / Lw;
    const LazyPt<Cfg>* pw = partial + (size_t)w * p.tmax + b;
    if (k1 - k0 >= MERGE_CAP) {
      if (role == 0) giant_enqueue(giant_count, giant_list, big_list, (uint32_t)w, b, k1 - k0 + 1);
    } else {
      acc = qpt_load<L>(&pw[k0], role);
      for (uint32_t k = k0 + 1; k <= k1; ++k) qadd<L>(acc, qpt_load<L>(&pw[k], role), role);
    }
  }
  qpt_store<L>(&dense[(size_t)w * (p.NB + 1) + b], role, acc);
}

// Tree over the quads of one block through LDS: on return quad 0 holds the sum of all nq quad values (nq a power of two).
template <class L>
__device__ __forceinline__ void quad_block_tree(QPt<L>& x, XYZZLazy<L>* sh, int q, int nq, int role) {
  for (int half = nq >> 1; half >= 1; half >>= 1) {
    if (q >= half && q < 2 * half) qpt_store<L>(&sh[q - half], role, x);
    __syncthreads();
    if (q < half) qadd<L>(x, qpt_load<L>(&sh[q], role), role);
    __syncthreads();
  }
}

// One 256-thread block (64 quads) per slice of a sliced bucket: strided private sums over the slice's partials, then the LDS tree.
template <class Cfg>
__global__ __launch_bounds__(256) void k_msm_giant_slices(MsmParams p, const uint32_t* __restrict__ start, const LazyPt<Cfg>* __restrict__ partial,
                                                          const uint32_t* __restrict__ giant_count, const uint32_t* __restrict__ big_list,
                                                          LazyPt<Cfg>* gscratch) {
  using L = typename Cfg::L;
  __shared__ XYZZLazy<L> sh[32];
  const uint32_t sl = blockIdx.x;
  const uint32_t nbig = giant_count[1] < GIANT_BIG_CAP ? giant_count[1] : GIANT_BIG_CAP;
  const int role = threadIdx.x & 3, qd = threadIdx.x >> 2;
  for (uint32_t q = blockIdx.y; q < nbig; q += gridDim.y) {  // usually none: the launch is GIANT_SLICES x GIANT_ROWS blocks that return at once
    const uint32_t w = big_list[2 * q], b = big_list[2 * q + 1];
    const uint32_t* st = start + (size_t)w * (p.NB + 2);
    const uint32_t Lw = lane_len(p, (int)w);
    const uint32_t k0 = st[b] / Lw, k1 = (st[b + 1] - 1) / Lw;
    const uint32_t per = (k1 - k0 + GIANT_SLICES) / GIANT_SLICES;  // ceil((k1 - k0 + 1) / slices)
    const uint32_t a = k0 + sl * per;
    uint32_t e = a + per;  // one past the slice's last partial
    if (e > k1 + 1) e = k1 + 1;
    const LazyPt<Cfg>* pw = partial + (size_t)w * p.tmax + b;
    QPt<L> acc = qpt_inf<L>();
    for (uint32_t k = a + qd; k < e; k += 64) qadd<L>(acc, qpt_load<L>(&pw[k], role), role);
    quad_block_tree<L>(acc, sh, qd, 64, role);
    if (qd == 0) qpt_store<L>(&gscratch[(size_t)q * GIANT_SLICES + sl], role, acc);
  }
}

// One 256-thread block (64 quads) per queued bucket: strided private sums, then the LDS tree; a sliced bucket folds its slice sums.
template <class Cfg>
__global__ __launch_bounds__(256) void k_msm_merge_giant(MsmParams p, const uint32_t* __restrict__ start,
                                                         const LazyPt<Cfg>* __restrict__ partial, LazyPt<Cfg>* dense,
                                                         const uint32_t* __restrict__ giant_count, const uint32_t* __restrict__ giant_list,
                                                         const uint32_t* __restrict__ big_list, const LazyPt<Cfg>* __restrict__ gscratch) {
  using L = typename Cfg::L;
  __shared__ XYZZLazy<L> sh[32];
  __shared__ uint32_t slot;
  const uint32_t count = giant_count[0];
  const uint32_t nbig = giant_count[1] < GIANT_BIG_CAP ? giant_count[1] : GIANT_BIG_CAP;
  const int role = threadIdx.x & 3, q = threadIdx.x >> 2;
  for (uint32_t g = blockIdx.x; g < count; g += gridDim.x) {
    const uint32_t tag = giant_list[2 * g], w = tag & ~GIANT_SLICED_FLAG, b = giant_list[2 * g + 1];
    QPt<L> acc = qpt_inf<L>();
    if (tag & GIANT_SLICED_FLAG) {
      __syncthreads();  // `slot` of the previous queue entry has been read by everyone
      if (threadIdx.x == 0) slot = GIANT_BIG_CAP;  // "not found": cannot happen for a flagged entry; then the bucket takes the walk below
      __syncthreads();
      if (threadIdx.x < nbig && big_list[2 * threadIdx.x] == w && big_list[2 * threadIdx.x + 1] == b) slot = threadIdx.x;
      __syncthreads();
      if (slot < GIANT_BIG_CAP) {
        if ((uint32_t)q < GIANT_SLICES) acc = qpt_load<L>(&gscratch[(size_t)slot * GIANT_SLICES + q], role);
      } else {
        const uint32_t* st = start + (size_t)w * (p.NB + 2);
        const uint32_t Lw = lane_len(p, (int)w);
        const uint32_t k0 = st[b] / Lw, k1 = (st[b + 1] - 1) / Lw;
        const LazyPt<Cfg>* pw = partial + (size_t)w * p.tmax + b;
        for (uint32_t k = k0 + q; k <= k1; k += 64) qadd<L>(acc, qpt_load<L>(&pw[k], role), role);
      }
    } else {
      const uint32_t* st = start + (size_t)w * (p.NB + 2);
      const uint32_t Lw = lane_len(p, (int)w);
      const uint32_t k0 = st[b] / Lw, k1 = (st[b + 1] - 1) / Lw;
      const LazyPt<Cfg>* pw = partial + (size_t)w * p.tmax + b;
      for (uint32_t k = k0 + q; k <= k1; k += 64) qadd<L>(acc, qpt_load<L>(&pw[k], role), role);
    }
    quad_block_tree<L>(acc, sh, q, 64, role);
    if (q == 0) qpt_store<L>(&dense[(size_t)w * (p.NB + 1) + b], role, acc);
  }
}

// ---- fused merge (round 3) -----------------------------------------------------------------------------------------
// Variant (msm_variant bit 4; measured slower on G1, see bucket_group): the window reduction folds a bucket's partial slots itself
// while it walks its segment (a bucket of ~32 entries spans 1.6 lanes of 55 on average: ~0.6 extra additions per bucket on the
// reduction's chain) instead of reading a dense array that a separate merge launch wrote. Buckets with more than MERGE_CAP partials
// still go through the block-wide tree (k_msm_merge_giant), which writes their sum to dense[]; k_msm_mark_giant queues them (what
// k_msm_merge does on the side).
template <class Cfg>
__global__ __launch_bounds__(256) void k_msm_mark_giant(MsmParams p, const uint32_t* __restrict__ start, uint32_t* giant_count, uint32_t* giant_list,
                                                        uint32_t* big_list) {
  const int w = blockIdx.y;
  const uint32_t b = blockIdx.x * 256 + threadIdx.x + 1;
  if (b > p.NB) return;
  const uint32_t* st = start + (size_t)w * (p.NB + 2);
  const uint32_t lo = st[b], hi = st[b + 1];
  const uint32_t Lw = lane_len(p, w);
  if (hi > lo && (hi - 1) / Lw - lo / Lw >= MERGE_CAP) giant_enqueue(giant_count, giant_list, big_list, (uint32_t)w, b, (hi - 1) / Lw - lo / Lw + 1);
}
// sum of bucket t of one window: four lanes per point / two lanes per Fp2 point
template <class Cfg>
__device__ __forceinline__ QPt<typename Cfg::L> qbucket_merged(uint32_t Lw, const uint32_t* __restrict__ st, const LazyPt<Cfg>* __restrict__ pw,
                                                               const LazyPt<Cfg>* __restrict__ dw, uint32_t t, int role) {
  using L = typename Cfg::L;
  const uint32_t lo = st[t], hi = st[t + 1];
  if (hi <= lo) return qpt_inf<L>();
  const uint32_t k0 = lo / Lw, k1 = (hi - 1) / Lw;
  if (k1 - k0 >= MERGE_CAP) return qpt_load<L>(&dw[t], role);
  QPt<L> acc = qpt_load<L>(&pw[t + k0], role);
  for (uint32_t k = k0 + 1; k <= k1; ++k) qadd<L>(acc, qpt_load<L>(&pw[t + k], role), role);
  return acc;
}
template <class Cfg>
__device__ __forceinline__ XYZZLazy<typename Cfg::LP> pbucket_merged(uint32_t Lw, const uint32_t* __restrict__ st, const LazyPt<Cfg>* __restrict__ pw,
                                                                     const LazyPt<Cfg>* __restrict__ dw, uint32_t t, int role) {
  using L = typename Cfg::LP;
  const uint32_t lo = st[t], hi = st[t + 1];
  if (hi <= lo) return XYZZLazy<L>::inf();
  const uint32_t k0 = lo / Lw, k1 = (hi - 1) / Lw;
  if (k1 - k0 >= MERGE_CAP) return dw[t].empty ? XYZZLazy<L>::inf() : pair_load(&dw[t], role);
  XYZZLazy<L> acc = pw[t + k0].empty ? XYZZLazy<L>::inf() : pair_load(&pw[t + k0], role);
  for (uint32_t k = k0 + 1; k <= k1; ++k) {
    if (pw[t + k].empty) continue;
    const XYZZLazy<L> q = pair_load(&pw[t + k], role);
    lazy_add_inl<L>(acc, q);
  }
  return acc;
}

// Segment k of window w folds dense buckets [t0, t1): returns sum_t t * B_t by the running-sum trick with explicit gaps
// (empty buckets are skipped; a gap of more than 4 empty buckets is bridged by one small scalar multiple).
template <class Cfg, bool FUSED>
__global__ __launch_bounds__(256) void k_msm_reduce(MsmParams p, const LazyPt<Cfg>* __restrict__ dense,
                                                    LazyPt<Cfg>* segres, const uint32_t* __restrict__ start, const LazyPt<Cfg>* __restrict__ partial) {
  using L = typename Cfg::L;
  const int w = blockIdx.y;
  const int role = threadIdx.x & 3;
  const uint32_t k = blockIdx.x * 64 + (threadIdx.x >> 2);
  if (k >= p.S) return;
  const uint32_t per = (p.NB + p.S - 1) / p.S;  // dense[w][b], b = 1..NB (index 0 unused)
  const uint32_t t0 = 1 + k * per;
  uint32_t t1 = t0 + per;
  if (t1 > p.NB + 1) t1 = p.NB + 1;
  QPt<L> running = qpt_inf<L>(), acc = qpt_inf<L>();
  uint32_t prev_b = 0;
  if (t0 < t1) {
    const LazyPt<Cfg>* dw = dense + (size_t)w * (p.NB + 1);
    const uint32_t* st = FUSED ? start + (size_t)w * (p.NB + 2) : nullptr;
    const LazyPt<Cfg>* pw = FUSED ? partial + (size_t)w * p.tmax : nullptr;
    for (uint32_t t = t1; t-- > t0;) {
      QPt<L> pt;
      if constexpr (FUSED) pt = qbucket_merged<Cfg>(lane_len(p, w), st, pw, dw, t, role);
      else pt = qpt_load<L>(&dw[t], role);
      if (pt.empty) continue;
      uint32_t gap = prev_b ? prev_b - t : 0;
      if (gap) {
        if (gap <= 4) {
          while (gap--) qadd<L>(acc, running, role);
        } else {
          const QPt<L> m = qmul_small<L>(running, gap, role);
          qadd<L>(acc, m, role);
        }
      }
      qadd<L>(running, pt, role);
      prev_b = t;
    }
    if (prev_b) {  // acc = sum (b - bmin) B_b ; add bmin * R
      const QPt<L> m = qmul_small<L>(running, prev_b, role);
      qadd<L>(acc, m, role);
    }
  }
  qpt_store<L>(&segres[(size_t)w * p.S + k], role, acc);
}

// The lane-serial reductions run in workgroups of four waves: a workgroup's waves spread over the four SIMDs of one CU and the
// <= 256 workgroups of a one-round launch over the CUs, so every SIMD gets exactly one wave (single-wave workgroups are packed
// unevenly: 870 of them took 0.38 ms where 544 took 0.26 ms, profiles/archive/r02_g_seg_stages.log).
constexpr int RED_BLK = 256;

// Lane-serial form of the same reduction (one lane per segment, whole points in registers): 1.75x less total work than the
// quad form but a 2-3x longer dependent chain per point operation. Round 2 first measured both at the SAME segment count
// (2048 segments: quad = 2176 waves = three rounds, 293 us against 263 us) and kept this one; sized to one round each
// (reduce_segments) and launched in four-wave workgroups the quad form wins by 17-25 %. Kept for A/B runs (msm_variant).
template <class Cfg>
__global__ __launch_bounds__(RED_BLK) void k_msm_reduce_serial(MsmParams p, const LazyPt<Cfg>* __restrict__ dense, LazyPt<Cfg>* segres) {
  using L = typename Cfg::L;
  const int w = blockIdx.y;
  const uint32_t k = blockIdx.x * RED_BLK + threadIdx.x;
  if (k >= p.S) return;
  const uint32_t per = (p.NB + p.S - 1) / p.S;
  const uint32_t t0 = 1 + k * per;
  uint32_t t1 = t0 + per;
  if (t1 > p.NB + 1) t1 = p.NB + 1;
  LazyPt<Cfg> running = LazyPt<Cfg>::inf(), acc = LazyPt<Cfg>::inf();
  uint32_t prev_b = 0;
  if (t0 < t1) {
    const LazyPt<Cfg>* dw = dense + (size_t)w * (p.NB + 1);
    for (uint32_t t = t1; t-- > t0;) {
      const LazyPt<Cfg> pt = dw[t];
      if (pt.empty) continue;
      uint32_t gap = prev_b ? prev_b - t : 0;
      if (gap) {
        if (gap <= 4) {
          while (gap--) lazy_add_inl<L>(acc, running);
        } else {
          const LazyPt<Cfg> m = lazy_mul_small<L, true>(running, gap);
          lazy_add_inl<L>(acc, m);
        }
      }
      lazy_add_inl<L>(running, pt);
      prev_b = t;
    }
    if (prev_b) {  // acc = sum (b - bmin) B_b ; add bmin * R
      const LazyPt<Cfg> m = lazy_mul_small<L, true>(running, prev_b);
      lazy_add_inl<L>(acc, m);
    }
  }
  segres[(size_t)w * p.S + k] = acc;
}

// The lane-serial reduction with two lanes per segment (G2, curve_pair.hpp): half the registers per lane (no scratch) and half
// the instructions on the dependent chain of every point operation.
template <class Cfg, bool FUSED>
__global__ __launch_bounds__(RED_BLK) void k_msm_reduce_pair(MsmParams p, const LazyPt<Cfg>* __restrict__ dense, LazyPt<Cfg>* segres,
                                                             const uint32_t* __restrict__ start, const LazyPt<Cfg>* __restrict__ partial) {
  using L = typename Cfg::LP;
  const int w = blockIdx.y;
  const int role = pair_role();
  const uint32_t k = blockIdx.x * (RED_BLK / 2) + (threadIdx.x >> 1);
  if (k >= p.S) return;
  const uint32_t per = (p.NB + p.S - 1) / p.S;
  const uint32_t t0 = 1 + k * per;
  uint32_t t1 = t0 + per;
  if (t1 > p.NB + 1) t1 = p.NB + 1;
  XYZZLazy<L> running = XYZZLazy<L>::inf(), acc = XYZZLazy<L>::inf();
  uint32_t prev_b = 0;
  if (t0 < t1) {
    const LazyPt<Cfg>* dw = dense + (size_t)w * (p.NB + 1);
    const uint32_t* st = FUSED ? start + (size_t)w * (p.NB + 2) : nullptr;
    const LazyPt<Cfg>* pw = FUSED ? partial + (size_t)w * p.tmax : nullptr;
    for (uint32_t t = t1; t-- > t0;) {
      XYZZLazy<L> pt;
      if constexpr (FUSED) {
        pt = pbucket_merged<Cfg>(lane_len(p, w), st, pw, dw, t, role);
        if (pt.empty) continue;
      } else {
        if (dw[t].empty) continue;
        pt = pair_load(&dw[t], role);
      }
      uint32_t gap = prev_b ? prev_b - t : 0;
      if (gap) {
        if (gap <= 4) {
          while (gap--) lazy_add_inl<L>(acc, running);
        } else {
          const XYZZLazy<L> m = lazy_mul_small<L, true>(running, gap);
          lazy_add_inl<L>(acc, m);
        }
      }
      lazy_add_inl<L>(running, pt);
      prev_b = t;
    }
    if (prev_b) {  // acc = sum (b - bmin) B_b ; add bmin * R
      const XYZZLazy<L> m = lazy_mul_small<L, true>(running, prev_b);
      lazy_add_inl<L>(acc, m);
    }
  }
  pair_store(&segres[(size_t)w * p.S + k], role, acc);
}

// Fold tree: block (j, w) sums in[w][j * 128 .. j * 128 + 127] (entries >= count are skipped) into out[w][j]; 256 threads =
// 64 quads, two loads per quad, then the LDS tree. Two launches take 2048 segment sums to one window sum.
template <class Cfg>
__global__ __launch_bounds__(256) void k_msm_fold_tree(const LazyPt<Cfg>* __restrict__ in, uint32_t stride_in, uint32_t count,
                                                       LazyPt<Cfg>* out, uint32_t stride_out, XYZZ<typename Cfg::Fq>* win_out) {
  using L = typename Cfg::L;
  using Fq = typename Cfg::Fq;
  __shared__ XYZZLazy<L> sh[32];
  const int w = blockIdx.y;
  const int role = threadIdx.x & 3, q = threadIdx.x >> 2;
  const LazyPt<Cfg>* a = in + (size_t)w * stride_in;
  const uint32_t i0 = blockIdx.x * 128 + q, i1 = i0 + 64;
  QPt<L> x = i0 < count ? qpt_load<L>(&a[i0], role) : qpt_inf<L>();
  if (i1 < count) qadd<L>(x, qpt_load<L>(&a[i1], role), role);
  quad_block_tree<L>(x, sh, q, 64, role);
  if (q == 0) {
    if (win_out) {  // last level: the window sum leaves in the arkworks encoding (lane r converts member r; no export launch)
      static_assert(sizeof(XYZZ<Fq>) == 4 * sizeof(Fq), "XYZZ members must be contiguous");
      (&win_out[w].x)[role] = x.empty ? Fq::zero() : x.v.to_fp();
    } else {
      qpt_store<L>(&out[(size_t)w * stride_out + blockIdx.x], role, x);
    }
  }
}

// In-place re-encoding of uploaded bases for LAZY curves: x*2^(32N) -> canonical x*R' (infinity stays 0,0)
template <class Cfg>
__global__ __launch_bounds__(256) void k_bases_repack(Affine<typename Cfg::Fq>* pts, size_t n) {
  if constexpr (Cfg::LAZY) {
    using L = typename Cfg::L;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      Affine<typename Cfg::Fq> p = pts[i];
      if (p.is_inf()) continue;
      p.x = L::repack_for_storage(p.x);
      p.y = L::repack_for_storage(p.y);
      pts[i] = p;
    }
  }
}

template <class Cfg>
__global__ void k_msm_gather_windows(const LazyPt<Cfg>* segres, uint32_t stride, int W, XYZZ<typename Cfg::Fq>* out) {
  const int w = threadIdx.x;
  if (w < W) out[w] = lazy_to_xyzz<typename Cfg::L, typename Cfg::Fq>(segres[(size_t)w * stride]);  // back to the arkworks encoding
}

// ---- host side -----------------------------------------------------------------------------------------
extern thread_local float tl_msm_timing[6];    // defined in msm.hip
extern thread_local uint32_t tl_msm_params[4];  // c, W, L, S of the last MSM


inline int choose_c(size_t n, int bits) {
  {
    const int c = tune().msm_c.load(std::memory_order_relaxed);
    if (c >= 2 && c <= 16) return c;
  }
  double best = 1e300;
  int best_c = 4;
  for (int c = 3; c <= 16; ++c) {  // digit codes are 15 bits + sign
    const double nb = double(size_t(1) << (c - 1));
    // per window: n mixed additions + the bucket stages, ~5 additions' worth per bucket (merge, running sums, segment multiple).
    // Re-checked after the one-round window reduction (profiles/archive/r02_g_csweep.log, r02_g_c1516.log): at 2^20 c = 15 and 16 tie
    // within 1-2 % (G1: 15 ahead, G2: 16 ahead), 2^17-2^19: 13 / 13 / 13-15, >= 2^21: 16.
    const double cost = windows_for(bits, c) * (double(n) + 5.0 * nb);
    if (cost < best) {
      best = cost;
      best_c = c;
    }
  }
  return best_c;
}

// L = sorted entries per accumulate lane. Every lane of every wave performs exactly L mixed additions and one wave of
// multiply-add code already saturates its SIMD's integer pipe, so the accumulate kernel takes ceil(waves / SIMDs) rounds of L
// additions: a sawtooth in L (measured, BN254 G1 2^22, 16 windows: L = 128 -> 8192 waves = 8.00 per SIMD, 4.82 ms; L = 112 ->
// 9.16 per SIMD = 10 rounds, 5.28 ms; L = 144 -> 5.36 ms; profiles/archive/r02_g_lsweep*.log). Longer lanes leave fewer partial sums
// to merge (n W / L of them, ~4.6e-5 addition rounds each); with few long rounds the last one is balanced less well (+~0.2 round).
// The plan takes the L in [16, 1024] with the smallest
//   (rounds(L) + 0.2) * L + 4.6e-5 * n * W / L.
// With 16 windows at the power-of-two sizes this lands on the former table (2^22 -> 128, 2^24 -> 256); it matters whenever
// n W / 64 is not a multiple of the SIMD count: 17 windows (BN254 at 2^20: L = 32 meant 8.5 waves per SIMD, 9 rounds of 32
// where 5 of 55 do, accumulate + merge 1.78 -> 1.73 ms; BN254 G2 5.48 -> 5.2 ms; 2^19: 1.33 -> 1.25 ms) and the arbitrary sizes of
// real proving keys.
// Small MSMs (round 4, profiles/archive/r04_zj_plan_sweep.log, r04_zk_short_lanes.log, interleaved): the round-count model above prices a
// SIMD with ONE wave on it, but a group whose accumulate kernel fits `occ` waves per SIMD (BN254 G1: 144 VGPRs -> 3) runs them
// interleaved, and with fewer than occ waves per SIMD in the whole launch the shorter lane wins: 2^15 c = 11 L = 16 -> 8 0.440 ->
// 0.371 ms, 2^16 c = 12 L = 23 -> 12 0.489 -> 0.426, 2^17 c = 13 L = 21 -> 12..16 0.569 -> 0.531..0.534; at 2^18 (L = 27 = exactly
// three waves per SIMD) and above the model's choice stands. Rule: never longer than the lane that fills occ waves per SIMD, down to 8
// entries (below ~6 10^5 entries the launch is latency, not throughput: left alone). occ = 1 (the G2 kernels, shared plans): unchanged.
inline uint32_t choose_lane_length(size_t n, int W, int occ = 1) {
  if (const int fl = tune().msm_l.load(std::memory_order_relaxed); fl > 0) return (uint32_t)fl;
  const double simds = (double)device_simds();
  const int wpb = ACC_BLK / 64;
  double best = 1e300;
  uint32_t best_L = 16;
  for (uint32_t L = 16; L <= 1024; ++L) {
    const uint64_t lanes = (n + L - 1) / L;
    const uint64_t waves = (uint64_t)W * ((lanes + ACC_BLK - 1) / ACC_BLK) * wpb;
    const double rounds = ceil((double)waves / simds);
    const double cost = (rounds + 0.2) * L + 4.6e-5 * (double)n * W / L;
    if (cost < best) {
      best = cost;
      best_L = L;
    }
    if (rounds <= 1) break;  // one round already: longer lanes only cost
  }
  // Round 5 (after balanced windows; profiles/archive/r05_f_ab_lane_floor.log, r05_g_ab_narrow_lane_length.log, r05_h_ab_lane_lengths_large.log,
  // interleaved, BN254 G1 / BLS12-381 G1 / Grumpkin): the lane that fills THREE waves per SIMD is the best or within 1 % of it on every G1
  // group at 2^15 .. 2^18, also where the kernel's registers only admit two (BLS12-381 G1 2^17: 14 against the former 20, -9.7 %; more,
  // shorter waves beat one full round), and from ~10^6 entries on a lane shorter than 12 entries loses to the partial sums it leaves the
  // merge kernel (2^16: 12 against 8, -4.3 % BN254 G1, -8.6 % BLS12-381 G1 against its former 11, -4.0 % Grumpkin); 2^15 stays at 8.
  const double entries = (double)n * W;
  if (occ >= 2 && entries >= 6e5) {
    uint32_t fill = (uint32_t)ceil(entries / (64.0 * simds * 3.0));
    const uint32_t floor_l = entries >= 1e6 ? 12 : 8;
    if (fill < floor_l) fill = floor_l;
    if (fill < best_L) best_L = fill;
  }
  return best_L;
}

// Window reduction: S segments of `per` consecutive buckets per window, `lanes_per_segment` lanes each (1, or 2 for the lane-pair
// form), a dependent chain of 2 per + ~21 point operations. One wave saturates its SIMD, so the stage takes
// ceil(W S lanes / 64 / SIMDs) rounds of that chain: as many segments as still fit ONE round (BN254 2^20: 17 windows x 3277
// segments of 5 buckets = 870 waves, chain 31 instead of 37 with the former 2048 x 8; a finer 4096 x 4 would need two rounds).
// tune "msm_seg_buckets" forces `per`.
inline uint32_t reduce_segments(uint32_t NB, int W, int lanes_per_segment) {
  int per = tune().msm_seg_buckets.load(std::memory_order_relaxed);
  if (per < 1 || per > 64) {
    const uint64_t s_max = (uint64_t)device_simds() * 64 / ((uint64_t)W * lanes_per_segment);
    per = (int)((NB + s_max - 1) / s_max);
    if (per < 2) per = 2;
  }
  const uint32_t S = (NB + per - 1) / per;
  return S < 1 ? 1 : S;
}

// Balanced windows (round 5). Uniform c-bit windows leave the top window whatever bits remain: 3 of 12 at 2^16 (c = 12, W = 22), 2 of 11 at
// 2^15, 8 of 13 at 2^17 / 2^18 -- a window that costs its n additions like every other, whose few buckets hold n / 4 .. n / 128 entries
// each (the oversized-bucket path: k_msm_giant_slices + k_msm_merge_giant 48 us of a 417 us MSM at 2^16, profiles/archive/r04_zp_msm_2p16_kernel_stats.csv)
// and whose bucket stage is sized like a full one. Here W windows share the bits + 1 bits evenly: c = ceil((bits + 1) / W), the low
// `wide` = bits + 1 - W (c - 1) windows take c bits, the others c - 1. c stays <= 16 (15-bit digit magnitudes). tune "msm_c" forces the
// uniform form (tests, A/B), "msm_balanced" = 0 turns this off, "msm_w" forces W.
struct WindowPlan {
  int c, W, wide;
};
inline WindowPlan choose_windows(size_t n, int bits) {
  const int forced_c = tune().msm_c.load(std::memory_order_relaxed);
  if ((forced_c >= 2 && forced_c <= 16) || tune().msm_balanced.load(std::memory_order_relaxed) == 0) {
    const int c = choose_c(n, bits);
    const int W = windows_for(bits, c);
    return {c, W, W};
  }
  const int total = bits + 1;  // one spare bit absorbs the final carry of the signed recoding
  auto balanced = [total](int W) {
    const int c = (total + W - 1) / W;
    return WindowPlan{c, W, total - W * (c - 1)};
  };
  const int forced_w = tune().msm_w.load(std::memory_order_relaxed);
  if (forced_w >= (total + 15) / 16 && forced_w <= MAX_WINDOWS && (total + forced_w - 1) / forced_w >= 3) return balanced(forced_w);
  // The number of windows is the one the uniform plan's width gives (choose_c: a cost model re-fitted by sweeps in rounds 2-4); the bits
  // are then spread evenly over them. Letting the cost model pick W freely was measured first and is worse where the model is least
  // exact: 2^19 took W = 18 (c = 15, 3 wide windows) for a modelled tie with the uniform W = 17 and ran 16 % slower, 2^18 W = 19 +1 %
  // (profiles/archive/r05_b_ab_balanced.log).
  return balanced(windows_for(bits, choose_c(n, bits)));
}
// width of window w / bit offset of window w in a plan
CSH_HD int window_bits(int c, int wide, int w) { return w < wide ? c : c - 1; }

struct PartialHeader {
  uint32_t magic, c, W, wide;  // wide == 0 (buffers of earlier builds) means W: uniform windows
  uint32_t pad[4];
};
constexpr uint32_t PARTIAL_MAGIC = 0x4d534d50u;  // "PMSM"

// ---- the pipeline in three pieces: plan, sort stage (depends on the scalars only), bucket stage (per set of bases) ----
inline MsmParams msm_plan(size_t n, int scalar_bits, int mont, int occ = 1) {
  MsmParams p;
  p.n = (uint32_t)n;
  const WindowPlan wp = choose_windows(n, scalar_bits);
  p.c = wp.c;
  p.W = wp.W;
  p.wide = wp.wide;
  p.NB = 1u << (p.c - 1);
  p.L = choose_lane_length(n, p.W, occ);
  // Narrow windows (balanced plan) hold twice the entries per bucket; giving their lanes 2 L entries would leave k_msm_merge the same
  // number of partial sums per bucket as in a wide window. Measured (profiles/archive/r05_g_ab_narrow_lane_length.log, interleaved, 2^14 .. 2^18,
  // three groups): +6 .. +26 % -- at these sizes the accumulate launch is a dependent chain per lane, and doubling it costs more than the
  // merge saves. One length is the default; tune "msm_variant" bit 6 (64) selects the doubled form (kept parity-tested for A/B).
  p.Ln = (p.wide < p.W && p.L <= 32 && (tune().msm_variant.load(std::memory_order_relaxed) & 64) != 0) ? 2 * p.L : p.L;
  const uint32_t max_lanes = (uint32_t)((n + p.L - 1) / p.L);
  p.tmax = p.NB + max_lanes + 2;  // partial slots per window: slot = bucket + lane
  p.S = reduce_segments(p.NB, p.W, 1);
  p.mont = mont;
  uint64_t ch = 512 / (uint64_t)p.W;
  const uint64_t by_size = n / (2ull * p.NB);
  if (ch > by_size) ch = by_size;
  if (ch < 1) ch = 1;
  p.CH = (uint32_t)ch;
  p.chunk_len = (uint32_t)((n + ch - 1) / ch);
  p.remap_n = p.remap_stride = p.remap_off = 0;
  p.dig_g = p.dig_wp = 0;
  tl_msm_params[0] = (uint32_t)p.c;
  tl_msm_params[1] = (uint32_t)p.W;
  tl_msm_params[2] = p.L;
  tl_msm_params[3] = p.S;
  return p;
}

// Merged-window mode (bases with fixed-base tables of g rows, table[k] = 2^(c W' k) P with W' = ceil(W / g)): the digit kernel
// still produces W windows of n codes, but window w is filed as row k = w / W' of sort window w' = w % W', every entry pointing
// at the precomputed multiple 2^(c W' k) P_i: the sort and bucket stages see W' windows of n g entries. g = W (one window, no
// Horner over windows afterwards) is the full merge; g = 2..4 keeps the sort in its efficient regime and halves / quarters the
// window reductions and the host Horner for g times the key memory.
struct MergedPlan {
  MsmParams dig;  // n points, W windows: digit kernel
  MsmParams srt;  // n g entries per window, W' windows: sort + bucket stages
};
inline MergedPlan msm_plan_merged(size_t n, int scalar_bits, int mont, int c, int groups, size_t table_stride, size_t offset, int occ = 1) {
  MergedPlan m;
  MsmParams& d = m.dig;
  d.n = (uint32_t)n;
  d.c = c;
  d.W = windows_for(scalar_bits, c);
  d.NB = 1u << (c - 1);
  d.L = d.tmax = d.S = d.CH = d.chunk_len = 0;
  d.mont = mont;
  d.remap_n = d.remap_stride = d.remap_off = 0;
  const int g = groups < 1 ? 1 : (groups > d.W ? d.W : groups);
  const int wp = (d.W + g - 1) / g;
  d.dig_g = (uint32_t)g;
  d.dig_wp = (uint32_t)wp;
  d.wide = d.W;  // table rows are 2^(c W' k) P: uniform windows
  d.Ln = 0;
  MsmParams& p = m.srt;
  const uint64_t n2 = (uint64_t)n * g;
  p.n = (uint32_t)n2;
  p.c = c;
  p.W = wp;
  p.NB = d.NB;
  p.L = choose_lane_length((size_t)n2, p.W, occ);
  const uint32_t max_lanes = (uint32_t)((n2 + p.L - 1) / p.L);
  p.tmax = p.NB + max_lanes + 2;
  p.S = reduce_segments(p.NB, p.W, 1);
  p.mont = mont;
  uint64_t ch = 512 / (uint64_t)p.W;
  const uint64_t by_size = n2 / (2ull * p.NB);
  if (ch > by_size) ch = by_size;
  if (ch < 1) ch = 1;
  p.CH = (uint32_t)ch;
  p.chunk_len = (uint32_t)((n2 + ch - 1) / ch);
  p.remap_n = (uint32_t)n;
  p.remap_stride = (uint32_t)table_stride;
  p.remap_off = (uint32_t)offset;
  p.dig_g = p.dig_wp = 0;
  p.wide = p.W;
  p.Ln = p.L;
  tl_msm_params[0] = (uint32_t)c;
  tl_msm_params[1] = (uint32_t)p.W;
  tl_msm_params[2] = p.L;
  tl_msm_params[3] = p.S;
  return m;
}

struct SortOut {  // what the bucket stage consumes
  uint32_t *start, *nlanes, *sorted;
};

// a merged plan with ONE bucket set and a window wider than the 16-bit digit codes of the plain sort stage: msm_sort_wide.hip
inline bool msm_sort_is_wide(const MsmParams& p) { return p.c > 16 && p.W == 1 && p.remap_n != 0; }
template <class Fr>
constexpr int fr_id_of() {
  return std::is_same<Fr, Bls381Fr>::value ? 1 : (std::is_same<Fr, Bn254Fq>::value ? 2 : (std::is_same<Fr, Bls377Fr>::value ? 3 : 0));
}

inline size_t msm_sort_bytes(const MsmParams& p, const MsmParams& pdig) {
  if (msm_sort_is_wide(p)) return msm_sort_wide_bytes(p, pdig);
  const size_t len = (size_t)p.NB + 2, n = p.n;
  size_t need = 0;
  need += 2 * Arena::padded(sizeof(uint32_t) * len * p.W);       // hist/cursor, start
  need += Arena::padded(sizeof(uint32_t) * MAX_WINDOWS);          // lanes per window
  need += Arena::padded(sizeof(uint32_t) * n * p.W);              // sorted
  need += Arena::padded(sizeof(uint16_t) * n * p.W);              // digit codes
  need += Arena::padded(sizeof(uint32_t) * (size_t)p.NB * p.CH * p.W);  // per-chunk bucket counts / prefixes
  need += msm_sort_extra_bytes(p);  // level-1 records + partition offsets (two-level scatter, large n)
  return need;
}

// digits + counting sort; takes its buffers from `ar` (already reserved). ev (nullable): records ev[1..3].
struct SortStageBufs {
  SortBuffers sb;
  bool two_level;
};
// window group [w0, w0 + nw) of the sort buffers: every array has a per-window stride
inline SortBuffers sort_group_view(const MsmParams& p, const SortBuffers& b, int w0) {
  const size_t len = (size_t)p.NB + 2, n = p.n;
  SortBuffers g = b;
  g.hist += len * w0;
  g.start += len * w0;
  g.nlanes += w0;
  g.sorted += n * w0;
  g.dig += n * w0;
  g.blkcnt += (size_t)p.NB * p.CH * w0;
  if (g.inter) g.inter += n * w0;
  if (g.part_cnt) g.part_cnt += (size_t)(p.NB / 256) * p.CH * w0;
  return g;
}
template <class Fr>
int msm_sort_prepare(const MsmParams& p, const MsmParams& pdig, const uint64_t* scalars_dev, hipStream_t st, Arena& ar, SortStageBufs* out) {
  const size_t len = (size_t)p.NB + 2, n = p.n;
  uint32_t* hist = ar.take<uint32_t>(len * p.W);
  uint32_t* start = ar.take<uint32_t>(len * p.W);
  uint32_t* nlanes = ar.take<uint32_t>(MAX_WINDOWS);
  uint32_t* sorted = ar.take<uint32_t>(n * p.W);
  uint16_t* dig = ar.take<uint16_t>(n * p.W);
  uint32_t* blkcnt = ar.take<uint32_t>((size_t)p.NB * p.CH * p.W);
  const bool two_level = msm_sort_two_level(p);
  uint64_t* inter = two_level ? ar.take<uint64_t>(n * p.W) : nullptr;
  uint32_t* part_cnt = two_level ? ar.take<uint32_t>((size_t)(p.NB / 256) * p.CH * p.W) : nullptr;
  const int g1 = grid_for(pdig.n, MSM_BLK, 65536);
  hipLaunchKernelGGL(k_msm_digits<Fr>, dim3(g1), dim3(MSM_BLK), 0, st, reinterpret_cast<const Fr*>(scalars_dev), pdig, dig);  // dig[w * n + i]
  out->sb = SortBuffers{hist, start, nlanes, sorted, dig, blkcnt, inter, part_cnt};
  out->two_level = two_level;
  return CSH_OK;
}
template <class Fr>
int msm_sort_stage(const MsmParams& p, const MsmParams& pdig, const uint64_t* scalars_dev, hipStream_t st, Arena& ar, SortOut* out, hipEvent_t* ev) {
  if (msm_sort_is_wide(p)) return msm_sort_wide_launch(fr_id_of<Fr>(), p, pdig, scalars_dev, st, ar, &out->start, &out->nlanes, &out->sorted, ev);
  SortStageBufs ss;
  CSH_TRY(msm_sort_prepare<Fr>(p, pdig, scalars_dev, st, ar, &ss));
  CSH_TRY(msm_sort_launch(p, ss.sb, st, ev));
  *out = SortOut{ss.sb.start, ss.sb.nlanes, ss.sb.sorted};
  return CSH_OK;
}

template <class Cfg>
size_t msm_bucket_bytes(const MsmParams* pp) {
  const MsmParams& p = *pp;
  const uint32_t max_lanes = (uint32_t)(((size_t)p.n + p.L - 1) / p.L);
  const uint32_t max_giant = (uint32_t)(((uint64_t)max_lanes * p.W) / MERGE_CAP + 1);
  const uint32_t giant_blocks = max_giant < 1024 ? max_giant : 1024;
  size_t need = 0;
  need += Arena::padded(sizeof(LazyPt<Cfg>) * (size_t)p.tmax * p.W);       // partials
  need += Arena::padded(sizeof(LazyPt<Cfg>) * (size_t)p.S * p.W);          // segment results
  need += Arena::padded(sizeof(LazyPt<Cfg>) * (size_t)(p.NB + 1) * p.W);   // dense bucket sums
  need += Arena::padded(sizeof(uint32_t) * (2 * (size_t)max_giant + 2) * 8 /* MAX_GROUPS */);
  need += Arena::padded(sizeof(uint32_t) * 2 * GIANT_BIG_CAP * 8) + Arena::padded(sizeof(LazyPt<Cfg>) * (size_t)GIANT_BIG_CAP * GIANT_SLICES * 8);  // sliced giants
  need += 2 * Arena::padded(sizeof(LazyPt<Cfg>) * (size_t)((p.S + 127) / 128) * p.W);  // fold-tree ping / pong
  (void)giant_blocks;
  return need;
}

// Scratch of the bucket stage, every array with a per-window stride (window groups run on offset views of it)
template <class Cfg>
struct BucketBufs {
  LazyPt<Cfg>*partial, *segres, *dense, *fold_a, *fold_b;
  uint32_t* giant;  // MAX_GROUPS x ([0] = count, [1] = count of sliced buckets, list from [2])
  uint32_t* big;    // MAX_GROUPS x GIANT_BIG_CAP x (window, bucket) of the sliced buckets
  LazyPt<Cfg>* gscratch;  // MAX_GROUPS x GIANT_BIG_CAP x GIANT_SLICES slice sums
  uint32_t fold_n1, max_lanes, max_giant, giant_blocks;
};
constexpr int MAX_GROUPS = 8;
template <class Cfg>
BucketBufs<Cfg> bucket_take(const MsmParams& p, Arena& ar) {
  BucketBufs<Cfg> b;
  b.max_lanes = (uint32_t)(((size_t)p.n + p.L - 1) / p.L);
  b.max_giant = (uint32_t)(((uint64_t)b.max_lanes * p.W) / MERGE_CAP + 1);
  b.giant_blocks = b.max_giant < 1024 ? b.max_giant : 1024;
  b.partial = ar.take<LazyPt<Cfg>>((size_t)p.tmax * p.W);
  b.segres = ar.take<LazyPt<Cfg>>((size_t)p.S * p.W);
  b.dense = ar.take<LazyPt<Cfg>>((size_t)(p.NB + 1) * p.W);
  b.giant = ar.take<uint32_t>((2 * (size_t)b.max_giant + 2) * MAX_GROUPS);
  b.big = ar.take<uint32_t>((size_t)2 * GIANT_BIG_CAP * MAX_GROUPS);
  b.gscratch = ar.take<LazyPt<Cfg>>((size_t)GIANT_BIG_CAP * GIANT_SLICES * MAX_GROUPS);
  b.fold_n1 = (p.S + 127) / 128;
  b.fold_a = ar.take<LazyPt<Cfg>>((size_t)b.fold_n1 * p.W);
  b.fold_b = ar.take<LazyPt<Cfg>>((size_t)b.fold_n1 * p.W);
  return b;
}

// accumulate -> merge -> reduce -> fold tree -> window sums for windows [w0, w0 + nw) on `st`; `group` picks the giant queue.
// ev (nullable): records ev[4] after the accumulation.
template <class Cfg>
int bucket_group(const void* points, const MsmParams& p_all, const SortOut& so, const BucketBufs<Cfg>& bb, int w0, int nw, int group, hipStream_t st,
                 void* win_out_dev, hipEvent_t* ev) {
  using Fq = typename Cfg::Fq;
  MsmParams p = p_all;  // the kernels index windows from 0: the wide / narrow boundary moves with the group's first window
  p.wide = p_all.wide > w0 ? p_all.wide - w0 : 0;
  const size_t len = (size_t)p.NB + 2;
  const uint32_t* start = so.start + len * w0;
  const uint32_t* nlanes = so.nlanes + w0;
  const uint32_t* sorted = so.sorted + (size_t)p.n * w0;
  LazyPt<Cfg>* partial = bb.partial + (size_t)p.tmax * w0;
  LazyPt<Cfg>* segres = bb.segres + (size_t)p.S * w0;
  LazyPt<Cfg>* dense = bb.dense + (size_t)(p.NB + 1) * w0;
  LazyPt<Cfg>* fold_a = bb.fold_a + (size_t)bb.fold_n1 * w0;
  LazyPt<Cfg>* fold_b = bb.fold_b + (size_t)bb.fold_n1 * w0;
  uint32_t* giant = bb.giant + (2 * (size_t)bb.max_giant + 2) * group;
  uint32_t* big = bb.big + (size_t)2 * GIANT_BIG_CAP * group;
  LazyPt<Cfg>* gscratch = bb.gscratch + (size_t)GIANT_BIG_CAP * GIANT_SLICES * group;
  const Affine<Fq>* bases = reinterpret_cast<const Affine<Fq>*>(points);
  {
    const int tb = tune().acc_blk.load(std::memory_order_relaxed);
    const int blk = (tb == 64 || tb == 128) ? tb : ACC_BLK;
    // tune "msm_variant" bit 1: two lanes per point on the G2 groups (curve_pair.hpp). Measured a wash against whole points
    // per lane (profiles/archive/r02_g_pair_stages.log): one wave of multiply-add code already saturates the SIMD's integer pipe, so
    // the second wave the smaller footprint buys has nothing to fill. Kept for A/B runs and covered by the GPU parity suite.
    // Round 3: on BLS12-381 G2 the lane pair IS the default (304 VGPRs, no VGPR spills, against 419 + 6 spills): the whole-point kernel
    // measured 7.13 .. 7.85 ms at 2^20 from box to box (its accumulation-register traffic makes it the more sensitive one), the pair
    // 7.24 .. 7.49, and 29.1 against 30.9 ms at 2^22 on the last box (profiles/archive/r03_c_g2_acc_forms.log, r03_v_g2_acc_forms.log). Bit 1
    // selects the other form of the group's default.
    bool pair = false;
    if constexpr (Cfg::PAIR) pair = Cfg::PAIR_ACC_DEFAULT != ((tune().msm_variant.load(std::memory_order_relaxed) & 2) != 0);
    if constexpr (Cfg::PAIR) {
      if (pair) {
        const dim3 ag((2 * bb.max_lanes + blk - 1) / blk, nw), ab(blk);
        hipLaunchKernelGGL(k_msm_accum_pair<Cfg>, ag, ab, 0, st, bases, p, start, nlanes, sorted, partial, giant);
      }
    }
    if (!pair) {
      const dim3 ag((bb.max_lanes + blk - 1) / blk, nw), ab(blk);
      hipLaunchKernelGGL(k_msm_accum<Cfg>, ag, ab, 0, st, bases, p, start, nlanes, sorted, partial, giant);
    }
  }
  if (ev) CSH_HIP(hipEventRecord(ev[4], st));
  const int variant = tune().msm_variant.load(std::memory_order_relaxed);
  int form;  // 0 lane-serial, 1 quad, 2 pair
  if constexpr (Cfg::PAIR) form = (variant & 1) ? 1 : ((variant & 4) ? 0 : 2);
  else form = (variant & 1) ? 0 : 1;
  // tune "msm_variant" bit 4 (16): the reduction merges a bucket's partial slots itself (fused merge above) instead of reading the
  // dense array of a separate merge launch. Measured (profiles/archive/r03_i_fused_merge.log): it LOSES on G1 -- tail 0.36 against 0.33 ms at
  // 2^20, 0.69 / 0.46 at 2^22, 1.00 / 0.54 at 2^24 (at L = 256 a bucket spans 2-3 lanes: the extra additions sit on the reduction's
  // dependent chain, while the merge launch does them with one quad per bucket, all buckets at once) and gains 5 % of the tail on
  // BLS12-381 G2 only. Kept as a parity-tested variant; the separate merge launch stays the default.
  const bool fused = form != 0 && (variant & 16) != 0;
  if (fused)
    hipLaunchKernelGGL(k_msm_mark_giant<Cfg>, dim3((p.NB + 255) / 256, nw), dim3(256), 0, st, p, start, giant, giant + 2, big);
  else
    hipLaunchKernelGGL(k_msm_merge<Cfg>, dim3((p.NB + TAIL_Q - 1) / TAIL_Q, nw), dim3(TAIL_BLK), 0, st, p, start, partial, dense, giant, giant + 2, big);
  // buckets with > MERGE_CAP partials (heavily repeated scalars): block-wide tree, grid-stride over the queue; the few with >= GIANT_BIG
  // partials (a witness's "1"s, a repeated value) are summed in GIANT_SLICES slices by one block each first (blocks past the count return)
  hipLaunchKernelGGL(k_msm_giant_slices<Cfg>, dim3(GIANT_SLICES, GIANT_ROWS), dim3(256), 0, st, p, start, partial, giant, big, gscratch);
  hipLaunchKernelGGL(k_msm_merge_giant<Cfg>, dim3(bb.giant_blocks), dim3(256), 0, st, p, start, partial, dense, giant, giant + 2, big, gscratch);
  // Window reduction, three forms of the same segment walk, each with as many segments as fit one round of its waves
  // (reduce_segments): four lanes per point (curve_quad.hpp; the default on the G1 groups: BN254 G1 2^20 tail 0.41 -> 0.33 ms,
  // BLS12-381 G1 0.96 -> 0.80 ms against the lane-serial form at equal launch width, profiles/archive/r02_g_seg_stages3.log), two lanes
  // per Fp2 point (curve_pair.hpp; the default on the G2 groups: half the registers per lane -- no spills where the whole-point
  // form needs 437-512 VGPRs + scratch -- and half the dependent chain per point operation; BN254 G2 2^20 tail 1.05 -> 0.81 ms,
  // BLS12-381 G2 3.2 -> 2.1 ms, profiles/archive/r02_g_seg_stages2.log), or one lane per segment. tune "msm_variant" picks another
  // form for A/B runs and tests: G1: bit 0 -> lane-serial; G2: bit 0 -> four lanes, bit 2 -> lane-serial.
  MsmParams pr = p;  // the reduction's own segmentation (never more segments than the plan sized the buffers for)
  if (form == 1) {
    pr.S = std::min(p.S, reduce_segments(p.NB, p.W, 4));
    if (fused) hipLaunchKernelGGL((k_msm_reduce<Cfg, true>), dim3((pr.S + 63) / 64, nw), dim3(256), 0, st, pr, dense, segres, start, partial);
    else hipLaunchKernelGGL((k_msm_reduce<Cfg, false>), dim3((pr.S + 63) / 64, nw), dim3(256), 0, st, pr, dense, segres, start, partial);
  } else if (form == 2) {
    pr.S = std::min(p.S, reduce_segments(p.NB, p.W, 2));
    if constexpr (Cfg::PAIR) {
      if (fused) hipLaunchKernelGGL((k_msm_reduce_pair<Cfg, true>), dim3((pr.S + RED_BLK / 2 - 1) / (RED_BLK / 2), nw), dim3(RED_BLK), 0, st, pr, dense, segres, start, partial);
      else hipLaunchKernelGGL((k_msm_reduce_pair<Cfg, false>), dim3((pr.S + RED_BLK / 2 - 1) / (RED_BLK / 2), nw), dim3(RED_BLK), 0, st, pr, dense, segres, start, partial);
    }
  } else {
    hipLaunchKernelGGL(k_msm_reduce_serial<Cfg>, dim3((pr.S + RED_BLK - 1) / RED_BLK, nw), dim3(RED_BLK), 0, st, pr, dense, segres);
  }
  // fold tree: S segment sums per window -> one, 128 per block and launch
  const LazyPt<Cfg>* cur = segres;
  uint32_t cur_n = pr.S, cur_stride = pr.S;
  LazyPt<Cfg>* nxt = fold_a;
  XYZZ<Fq>* win_out = reinterpret_cast<XYZZ<Fq>*>(win_out_dev) + w0;
  bool exported = false;
  while (cur_n > 1) {
    const uint32_t out_n = (cur_n + 127) / 128;
    exported = out_n == 1;  // the last level writes the window sums in the arkworks encoding itself
    hipLaunchKernelGGL(k_msm_fold_tree<Cfg>, dim3(out_n, nw), dim3(256), 0, st, cur, cur_stride, cur_n, nxt, bb.fold_n1,
                       exported ? win_out : (XYZZ<Fq>*)nullptr);
    cur = nxt;
    cur_n = out_n;
    cur_stride = bb.fold_n1;
    nxt = nxt == fold_a ? fold_b : fold_a;
  }
  if (!exported)  // a single segment per window: nothing to fold
    hipLaunchKernelGGL(k_msm_gather_windows<Cfg>, dim3(1), dim3(MAX_WINDOWS), 0, st, cur, cur_stride, nw, win_out);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}

// accumulate -> merge -> reduce -> fold for one set of bases; win_out_dev: W x XYZZ<Fq> on the device. ev (nullable):
// records ev[4] after the accumulation and ev[5] at the end.
template <class Cfg>
int msm_bucket_stage(const void* points, const MsmParams* pp, const SortOut* so, hipStream_t st, Arena* arp, void* win_out_dev, hipEvent_t* ev) {
  const MsmParams& p = *pp;
  const BucketBufs<Cfg> bb = bucket_take<Cfg>(p, *arp);
  CSH_TRY((bucket_group<Cfg>(points, p, *so, bb, 0, p.W, 0, st, win_out_dev, ev)));
  if (ev) CSH_HIP(hipEventRecord(ev[5], st));
  return CSH_OK;
}

// (A software pipeline over window groups on two streams -- group g's sort and tail under the accumulation of its
// neighbours -- was built and measured: BN254 G1 2^20 2.07 -> 2.19 / 2.49 / 2.61 ms with 2 / 3 / 4 groups, 2^24 23.4 -> 25.2 /
// 23.9 / 24.8 ms, worse on every group and size, profiles/archive/r02_c5_pipeline.log. The accumulate workgroups hold every SIMD's
// register file, so the other stream's kernels wait for them to retire: the stages serialise anyway and each group adds its
// own ramp-up / ramp-down. Not kept; bucket_group() keeps the window-offset form it needed.)

// merged-window mode applies when the handle carries tables, the call covers a good part of them (a tiny MSM against a
// 2^15-bucket table would pay the bucket reduction for nothing) and the entry ids fit 31 bits
inline bool msm_use_table(const Bases* B, size_t n) {
  if (!B->table || n == 0 || tune().msm_no_table.load(std::memory_order_relaxed)) return false;
  if ((uint64_t)n * (uint64_t)B->table_W >= (uint64_t(1) << 31) || (uint64_t)B->n * (uint64_t)B->table_W >= (uint64_t(1) << 31)) return false;
  return n * 8 >= B->n;
}

// accumulate waves of this group's kernel that fit one SIMD together (register-limited: 3 on BN254 G1 / Grumpkin, 2 on BLS12-381 G1,
// 1 on the G2 groups); 1 where it cannot be asked (no device)
template <class Cfg>
int accum_occupancy() {
  static std::atomic<int> cache{0};
  int v = cache.load(std::memory_order_relaxed);
  if (v) return v;
  v = 1;
  if constexpr (!Cfg::PAIR) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_msm_accum<Cfg>, ACC_BLK, 0) == hipSuccess && blocks > 0) {
      v = blocks * (ACC_BLK / 64) / 4;
      v = v < 1 ? 1 : (v > 8 ? 8 : v);
    }
  }
  cache.store(v, std::memory_order_relaxed);
  return v;
}

template <class Cfg>
int msm_windows_dev(const Bases* B, size_t offset, size_t n, const uint64_t* scalars_dev, int mont, hipStream_t st,
                    XYZZ<typename Cfg::Fq>* win_out_dev /* W entries, device */, MsmParams* p_out) {
  using Fr = typename Cfg::Fr;
  using Fq = typename Cfg::Fq;
  const bool merged = msm_use_table(B, n);
  MsmParams p, pdig;
  const void* points;
  if (merged) {
    const MergedPlan m = msm_plan_merged(n, Fr::Params::BITS, mont, B->table_c, B->table_W, B->n, offset, accum_occupancy<Cfg>());
    p = m.srt;
    pdig = m.dig;
    points = B->table;
  } else {
    p = pdig = msm_plan(n, Fr::Params::BITS, mont, accum_occupancy<Cfg>());
    points = reinterpret_cast<const Affine<Fq>*>(B->points) + offset;
  }
  *p_out = p;  // merged: W = 1 (the single window sum is the result)
  Arena& ar = arena_for(st);
  CSH_TRY(ar.reserve(msm_sort_bytes(p, pdig) + msm_bucket_bytes<Cfg>(&p)));
  const bool timing = tune().msm_timing.load(std::memory_order_relaxed) != 0;
  hipEvent_t ev[7];
  if (timing) {
    for (auto& e : ev) CSH_HIP(hipEventCreate(&e));
    CSH_HIP(hipEventRecord(ev[0], st));
  }
  SortOut so;
  CSH_TRY(msm_sort_stage<Fr>(p, pdig, scalars_dev, st, ar, &so, timing ? ev : nullptr));
  CSH_TRY(msm_bucket_stage<Cfg>(points, &p, &so, st, &ar, win_out_dev, timing ? ev : nullptr));
  if (timing) {
    CSH_HIP(hipEventSynchronize(ev[5]));
    for (int i = 0; i < 5; ++i) CSH_HIP(hipEventElapsedTime(&tl_msm_timing[i], ev[i], ev[i + 1]));
    CSH_HIP(hipEventElapsedTime(&tl_msm_timing[5], ev[0], ev[5]));
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  return CSH_OK;
}

// Fixed-base tables: table[k * n + i] = 2^(step k) P_i for k < W rows (step = c W' doublings; `c` below is that step). One lane per point:
// back to the arkworks encoding, c doublings per window in XYZZ, one inversion per window to return to affine, re-encoded
// for storage. One-off per set of bases (a proving key): ~8.4 k field multiplications per G1 point.
template <class Cfg>
__global__ __launch_bounds__(128) void k_bases_precompute(const Affine<typename Cfg::Fq>* __restrict__ pts, size_t n, int c, int W,
                                                          Affine<typename Cfg::Fq>* __restrict__ table) {
  using Fq = typename Cfg::Fq;
  using L = typename Cfg::L;
  const size_t i = blockIdx.x * (size_t)128 + threadIdx.x;
  if (i >= n) return;
  const Affine<Fq> stored = pts[i];
  table[i] = stored;
  if (stored.is_inf()) {
    for (int w = 1; w < W; ++w) table[(size_t)w * n + i] = stored;
    return;
  }
  Affine<Fq> q{L::unpack(stored.x).to_fp(), L::unpack(stored.y).to_fp()};
  for (int w = 1; w < W; ++w) {
    XYZZ<Fq> acc = XYZZ<Fq>::from_affine(q);
    for (int k = 0; k < c; ++k) acc = xyzz_dbl(acc);
    q = xyzz_to_affine(acc);
    Affine<Fq> out = q;
    if (!q.is_inf()) {
      out.x = L::repack_for_storage(q.x);
      out.y = L::repack_for_storage(q.y);
    }
    table[(size_t)w * n + i] = out;
  }
}

template <class Cfg>
int precompute_table_t(Bases* B, int c, int groups, hipStream_t st) {
  using Fq = typename Cfg::Fq;
  using Fr = typename Cfg::Fr;
  const int windows = windows_for(Fr::Params::BITS, c);
  const int asked = groups <= 0 || groups > windows ? windows : groups;
  CSH_REQUIRE(c <= 16 || asked == windows, "windows wider than 16 bits need one table row per window (groups = 0 or >= the window count)");
  const int wp = (windows + asked - 1) / asked;                    // W': windows per bucket set
  const int W = (windows + wp - 1) / wp;                           // table rows actually referenced (rows >= this would never be read)
  const int step = c * wp;                                         // doublings between rows: c * W'
  void* t = nullptr;
  hipError_t e = hipMalloc(&t, sizeof(Affine<Fq>) * B->n * (size_t)W);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) for the fixed-base tables failed: %s", sizeof(Affine<Fq>) * B->n * (size_t)W, hipGetErrorString(e));
    return CSH_ERR_OOM;
  }
  hipLaunchKernelGGL(k_bases_precompute<Cfg>, dim3((unsigned)((B->n + 127) / 128)), dim3(128), 0, st, reinterpret_cast<const Affine<Fq>*>(B->points),
                     B->n, step, W, reinterpret_cast<Affine<Fq>*>(t));
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    (void)hipFree(t);
    set_error("fixed-base table kernel failed: %s", hipGetErrorString(e));
    return CSH_ERR_HIP;
  }
  B->table = t;
  B->table_c = c;
  B->table_W = W;
  return CSH_OK;
}

// type-erased host fold (the multi-MSM entry dispatches per set of bases at run time)
template <class Cfg>
void fold_windows_erased(const void* wins_host, int W, int c, int wide, void* out_jacobian);

// Horner over window sums + affine normalisation -> arkworks Projective (x, y, 1) / (1, 1, 0). Runs on the host in
// 64-bit limbs (host_fp64.hpp): W*c sequential doublings are latency, not throughput, and a CPU core does them faster.
template <class F>
XYZZ<F> horner_windows(const XYZZ<F>* wins, int W, int c, int wide);
template <class Fq>
void fold_windows_host(const XYZZ<Fq>* wins32, int W, int c, int wide, void* out_jacobian) {
  using F = typename Host64<Fq>::type;
  static_assert(sizeof(XYZZ<F>) == sizeof(XYZZ<Fq>) && sizeof(Jac<F>) == sizeof(Jac<Fq>), "64-bit view must alias the device encoding");
  std::vector<XYZZ<F>> wins(W);
  memcpy((void*)wins.data(), wins32, sizeof(XYZZ<F>) * W);
  Affine<F> a = xyzz_to_affine(horner_windows<F>(wins.data(), W, c, wide));
  Jac<F> j = a.is_inf() ? Jac<F>::inf() : Jac<F>{a.x, a.y, F::one()};
  memcpy(out_jacobian, &j, sizeof(j));
}

template <class Cfg>
void fold_windows_erased(const void* wins_host, int W, int c, int wide, void* out_jacobian) {
  fold_windows_host<typename Cfg::Fq>(reinterpret_cast<const XYZZ<typename Cfg::Fq>*>(wins_host), W, c, wide, out_jacobian);
}

template <class Cfg>
int msm_t(const Bases* B, size_t offset, size_t n, const uint64_t* scalars_dev, int mont, void* out_host, hipStream_t st) {
  using Fq = typename Cfg::Fq;
  if (n == 0) {
    Jac<Fq> j = Jac<Fq>::inf();
    memcpy(out_host, &j, sizeof(j));
    return CSH_OK;
  }
  Arena& wa = arena_for((hipStream_t)((uintptr_t)st ^ 0x2));
  CSH_TRY(wa.reserve(sizeof(XYZZ<Fq>) * MAX_WINDOWS));
  XYZZ<Fq>* win_dev = wa.take<XYZZ<Fq>>(MAX_WINDOWS);
  MsmParams p;
  CSH_TRY((msm_windows_dev<Cfg>(B, offset, n, scalars_dev, mont, st, win_dev, &p)));
  // the W window sums come back through a page-locked buffer of the calling thread's lane (a DMA copy straight into it; a
  // pageable destination is staged: 26 us per call. Letting the export kernel write host memory itself was measured and is far
  // worse: its dword stores cross PCIe one by one, +0.1 ms on G1, +0.8 ms on G2)
  void *pin_host = nullptr, *pin_dev = nullptr;
  std::vector<XYZZ<Fq>> pageable;
  XYZZ<Fq>* wins = nullptr;
  if (pinned_for(st, sizeof(XYZZ<Fq>) * MAX_WINDOWS, &pin_host, &pin_dev)) {
    wins = reinterpret_cast<XYZZ<Fq>*>(pin_host);
  } else {
    pageable.resize(p.W);
    wins = pageable.data();
  }
  CSH_HIP(hipMemcpyAsync(wins, win_dev, sizeof(XYZZ<Fq>) * p.W, hipMemcpyDeviceToHost, st));
  CSH_HIP(hipStreamSynchronize(st));
  fold_windows_host<Fq>(wins, p.W, p.c, p.wide, out_host);
  return CSH_OK;
}

// header of a partial buffer, written on the stream (no host round trip: the partial stays asynchronous)
template <int UNUSED = 0>
__global__ void k_msm_partial_header(PartialHeader* out, uint32_t c, uint32_t W, uint32_t wide) {
  PartialHeader h;
  h.magic = PARTIAL_MAGIC;
  h.c = c;
  h.W = W;
  h.wide = wide;
  for (int i = 0; i < 4; ++i) h.pad[i] = 0;
  *out = h;
}

// One range of a split MSM: PartialHeader + W window sums (XYZZ, arkworks encoding) left on the device. Asynchronous on
// `st` unless `sync`.
template <class Cfg>
int msm_partial_t(const Bases* B, size_t offset, size_t n, const uint64_t* scalars_dev, int mont, void* out_dev, hipStream_t st, bool sync) {
  using Fq = typename Cfg::Fq;
  XYZZ<Fq>* wins = reinterpret_cast<XYZZ<Fq>*>(static_cast<char*>(out_dev) + sizeof(PartialHeader));
  CSH_HIP(hipMemsetAsync(out_dev, 0, sizeof(PartialHeader) + sizeof(XYZZ<Fq>) * MAX_WINDOWS, st));
  uint32_t c = 0, W = 0, wide = 0;
  if (n > 0) {
    MsmParams p;
    CSH_TRY((msm_windows_dev<Cfg>(B, offset, n, scalars_dev, mont, st, wins, &p)));
    c = (uint32_t)p.c;
    W = (uint32_t)p.W;
    wide = (uint32_t)p.wide;
  }
  hipLaunchKernelGGL(k_msm_partial_header<0>, dim3(1), dim3(1), 0, st, reinterpret_cast<PartialHeader*>(out_dev), c, W, wide);
  CSH_HIP(hipGetLastError());
  if (sync) CSH_HIP(hipStreamSynchronize(st));
  return CSH_OK;
}

// c doublings of an XYZZ point on the host, run in Jacobian coordinates: dbl-2009-l for a = 0 costs 2M + 5S against 6M + 3S for
// the XYZZ doubling, and the round trip is 4 + 2 multiplications: (X, Y, ZZ, ZZZ) = (X ZZ^2 : Y ZZ^3 : ZZZ) because ZZZ^2 = ZZ^3,
// back with (X : Y : Z) = (X, Y, Z^2, Z^3). ~16 % fewer field multiplications over the c (W - 1) sequential doublings of the fold.
template <class F>
XYZZ<F> xyzz_dbl_many_host(const XYZZ<F>& p, int c) {
  if (p.is_inf() || c <= 0) return p;
  const F zz2 = F::sqr(p.zz);
  F X = F::mul(p.x, zz2), Y = F::mul(p.y, F::mul(zz2, p.zz)), Z = p.zzz;
  for (int k = 0; k < c; ++k) {
    if (Y.is_zero()) return XYZZ<F>::inf();  // 2-torsion: not on these prime-order groups, kept for completeness
    const F A = F::sqr(X), B = F::sqr(Y), C = F::sqr(B);
    const F t = F::add(X, B);
    const F D = F::mul2(F::sub(F::sub(F::sqr(t), A), C));
    const F E = F::mul3(A);
    const F X3 = F::sub(F::sqr(E), F::mul2(D));
    const F Y3 = F::sub(F::mul(E, F::sub(D, X3)), F::mul8(C));
    Z = F::mul2(F::mul(Y, Z));
    X = X3;
    Y = Y3;
  }
  const F ZZ = F::sqr(Z);
  return {X, Y, ZZ, F::mul(Z, ZZ)};
}

// Horner value (not yet normalised) of one set of window sums, 64-bit host limbs
// (window w holds c bits for w < wide, c - 1 above: sum_w 2^(offset of w) S_w, the shift before adding S_w is the width of window w)
template <class F>
XYZZ<F> horner_windows(const XYZZ<F>* wins, int W, int c, int wide) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int w = W - 1; w >= 0; --w) {
    acc = xyzz_dbl_many_host(acc, window_bits(c, wide, w));
    acc = xyzz_add_inl(acc, wins[w]);
  }
  return acc;
}

template <class Cfg>
int fold_partials_t(const void* partials_host, size_t nparts, void* out_jacobian) {
  using Fq = typename Cfg::Fq;
  using F = typename Host64<Fq>::type;
  const size_t stride = sizeof(PartialHeader) + sizeof(XYZZ<Fq>) * MAX_WINDOWS;
  XYZZ<F> total = XYZZ<F>::inf();
  // partials with the same window layout (the usual case: equal shares per rank) are summed window by window first,
  // so the W*c doublings are paid once
  std::vector<XYZZ<F>> sum;
  uint32_t sum_c = 0, sum_W = 0, sum_wide = 0;
  for (size_t k = 0; k < nparts; ++k) {
    const char* base = static_cast<const char*>(partials_host) + k * stride;
    PartialHeader h;
    memcpy(&h, base, sizeof h);
    CSH_REQUIRE(h.magic == PARTIAL_MAGIC, "fold_partials: bad partial header");
    if (h.W == 0) continue;
    CSH_REQUIRE(h.W <= (uint32_t)MAX_WINDOWS && h.c >= 2 && h.c <= 22 && h.wide <= h.W, "fold_partials: bad window parameters");
    const uint32_t wide = h.wide ? h.wide : h.W;  // 0: a buffer written before balanced windows existed
    std::vector<XYZZ<F>> wins(h.W);
    memcpy((void*)wins.data(), base + sizeof h, sizeof(XYZZ<F>) * h.W);
    if (sum.empty()) {
      sum.swap(wins);
      sum_c = h.c;
      sum_W = h.W;
      sum_wide = wide;
    } else if (h.c == sum_c && h.W == sum_W && wide == sum_wide) {
      for (uint32_t w = 0; w < h.W; ++w) sum[w] = xyzz_add_inl(sum[w], wins[w]);
    } else {
      total = xyzz_add_inl(total, horner_windows<F>(wins.data(), (int)h.W, (int)h.c, (int)wide));
    }
  }
  if (!sum.empty()) total = xyzz_add_inl(total, horner_windows<F>(sum.data(), (int)sum_W, (int)sum_c, (int)sum_wide));
  Affine<F> a = xyzz_to_affine(total);
  Jac<F> j = a.is_inf() ? Jac<F>::inf() : Jac<F>{a.x, a.y, F::one()};
  memcpy(out_jacobian, &j, sizeof(j));
  return CSH_OK;
}

template <class Cfg>
int repack_bases_t(Bases* B, hipStream_t st) {
  if constexpr (Cfg::LAZY) {
    if (B->n) {
      hipLaunchKernelGGL(k_bases_repack<Cfg>, dim3(grid_for(B->n, 256)), dim3(256), 0, st, (Affine<typename Cfg::Fq>*)B->points, B->n);
      CSH_HIP(hipGetLastError());
      CSH_HIP(hipStreamSynchronize(st));
    }
  }
  return CSH_OK;
}

// The accumulate kernels are instantiated in their own translation units (msm_accum_*.hip, which define CSH_PIN_MADS 3: the
// product-scanning Montgomery multiplication with its multiply-add order pinned); the per-configuration units below see them
// as explicit-instantiation declarations, so their own copy of the field code stays unpinned for the tail kernels.
#define CSH_MSM_ACCUM_INSTANTIATE(KW, CFG)                                                                                               \
  KW template __global__ void k_msm_accum<CFG>(const Affine<typename CFG::Fq>* __restrict__, MsmParams, const uint32_t* __restrict__,  \
                                               const uint32_t* __restrict__, const uint32_t* __restrict__, LazyPt<CFG>*, uint32_t*);
#define CSH_MSM_ACCUM_PAIR_INSTANTIATE(KW, CFG)                                                                                               \
  KW template __global__ void k_msm_accum_pair<CFG>(const Affine<typename CFG::Fq>* __restrict__, MsmParams, const uint32_t* __restrict__,  \
                                                    const uint32_t* __restrict__, const uint32_t* __restrict__, LazyPt<CFG>*, uint32_t*);

// one explicit instantiation set per configuration (msm_inst_*.hip); everyone else only sees the declarations
#define CSH_MSM_INSTANTIATE(KW, CFG)                                                                                            \
  KW template int msm_t<CFG>(const Bases*, size_t, size_t, const uint64_t*, int, void*, hipStream_t);                           \
  KW template int msm_partial_t<CFG>(const Bases*, size_t, size_t, const uint64_t*, int, void*, hipStream_t, bool);                 \
  KW template int fold_partials_t<CFG>(const void*, size_t, void*);                                                             \
  KW template int repack_bases_t<CFG>(Bases*, hipStream_t);                                                                     \
  KW template size_t msm_bucket_bytes<CFG>(const MsmParams*);                                                                   \
  KW template int msm_bucket_stage<CFG>(const void*, const MsmParams*, const SortOut*, hipStream_t, Arena*, void*, hipEvent_t*);          \
  KW template int precompute_table_t<CFG>(Bases*, int, int, hipStream_t);                                                            \
  KW template void fold_windows_erased<CFG>(const void*, int, int, int, void*);

// the seven group configurations, by run-time (curve, group)
#define CURVE_DISPATCH(curve, group, CALL)                                                      \
  do {                                                                                          \
    if ((curve) == CSH_BN254 && (group) == CSH_G1) { using Cfg = csh::Bn254G1Cfg; return CALL; }     \
    if ((curve) == CSH_BN254 && (group) == CSH_G2) { using Cfg = csh::Bn254G2Cfg; return CALL; }     \
    if ((curve) == CSH_BLS12_381 && (group) == CSH_G1) { using Cfg = csh::Bls381G1Cfg; return CALL; } \
    if ((curve) == CSH_BLS12_381 && (group) == CSH_G2) { using Cfg = csh::Bls381G2Cfg; return CALL; } \
    if ((curve) == CSH_GRUMPKIN && (group) == CSH_G1) { using Cfg = csh::GrumpkinG1Cfg; return CALL; }  \
    if ((curve) == CSH_BLS12_377 && (group) == CSH_G1) { using Cfg = csh::Bls377G1Cfg; return CALL; } \
    if ((curve) == CSH_BLS12_377 && (group) == CSH_G2) { using Cfg = csh::Bls377G2Cfg; return CALL; } \
    csh::set_error("unknown curve/group %d/%d", (int)(curve), (int)(group));                     \
    return CSH_ERR_INVALID;                                                                     \
  } while (0)

inline int scalar_bits_of(csh_curve_t c) {
  return c == CSH_BLS12_381 ? Bls381FrParams::BITS : c == CSH_GRUMPKIN ? Bn254FqParams::BITS : c == CSH_BLS12_377 ? Bls377FrParams::BITS : Bn254FrParams::BITS;
}
inline size_t point_bytes_of(csh_curve_t c, csh_group_t g) {
  const size_t fq = (c == CSH_BLS12_381 || c == CSH_BLS12_377) ? 48 : 32;
  return 2 * fq * (g == CSH_G2 ? 2 : 1);
}
inline size_t partial_bytes_of(csh_curve_t c, csh_group_t g) { return sizeof(PartialHeader) + 2 * point_bytes_of(c, g) * MAX_WINDOWS; }

}  // namespace csh
