// Radix-2 NTT over Fr (BN254 / BLS12-381) on gfx950, LDS-tiled multi-pass.
//
// Conventions (the reference's taceo_ark_algebra::fft::Domain, see include/cosnarks_hip.h):
//   fft_out_to_in : bit-reversed coefficients -> natural-order evaluations   (decimation in time)
//   ifft_in_to_out: natural-order evaluations -> bit-reversed coefficients/n (decimation in frequency)
// Stage s (half-size m = 2^s) pairs index i (bit s clear) with i + m, twiddle w^((i mod m) * n/(2m)).
//
// A pass executes stages [s0, s0+k) for one tile entirely in LDS: the tile is the 2^k stage-bit values
// x 2^cb low "column" entries x ncomp components (ncomp = 2 for Rep3 shares), so every HBM access is a
// contiguous run of 2^cb * ncomp * 32 bytes (>= 256 B) and a 2^22 transform needs 3 read+write sweeps.
// Default path: butterflies in the signed lazy 9 x 29-bit field (k_ntt_pass_lazy, limb-major LDS arrays); the 32-bit
// CIOS pass (k_ntt_pass: two 16-byte halves per element, conflict-free ds_read/write_b128) remains selectable with
// CSH_NTT_LAZY=0 for A/B measurements.
// the butterflies' products run with the pinned multiply-add order (field29.hpp): -5.5 % per transform, 52-61 VGPRs as before
#define CSH_PIN_MADS 3
#include <stdlib.h>

#include <mutex>
#include <string.h>
#include <type_traits>

#include "common.hpp"
#include "field.hpp"
#include "field29.hpp"

namespace csh {

constexpr int NTT_MAX_THREADS = 1024;
constexpr int NTT_TILE_LOG = 11;  // 2^11 field elements = 64 KiB of LDS per workgroup

struct Domain {
  csh_curve_t curve;
  int device;
  uint32_t log_n;
  size_t n;
  void* tw_fwd;  // w^j, j < n/2
  void* tw_inv;  // w^-j
  void* tw_fwd_lazy;  // the powers the lazy butterflies use, as packed canonical w^j * R' (R' = 2^261), one contiguous run per stage
  void* tw_inv_lazy;
  uint32_t n_inv_lazy[8];  // (1/n) * R', packed
  uint32_t gen[8], gen_inv[8], n_inv[8];  // Montgomery
  // The scaled coset table of the witness map (shift^i / n, ntt_coset_table_scaled) for the FIRST shift the domain is asked for: a reduction
  // always uses the same coset of its domain (reduction.rs:100), so every witness map after the first reads it instead of rebuilding it
  // (91 us of a 2^20 witness map). Another shift on the same domain is computed into the caller's scratch as before and is not cached.
  std::mutex cs_mu;
  uint64_t cs_shift[4] = {0, 0, 0, 0};
  void* cs_table = nullptr;
  hipEvent_t cs_ready = nullptr;
};

template <class F>
__device__ __forceinline__ F lds_get(const uint4* lo, const uint4* hi, int e) {
  union { F f; uint4 q[2]; } u;
  u.q[0] = lo[e];
  u.q[1] = hi[e];
  return u.f;
}
template <class F>
__device__ __forceinline__ void lds_put(uint4* lo, uint4* hi, int e, const F& f) {
  union { F f; uint4 q[2]; } u;
  u.f = f;
  lo[e] = u.q[0];
  hi[e] = u.q[1];
}

// One pass = stages [s0, s0+k) of a size-2^L transform. CC = 2^cb * ncomp contiguous elements.
template <class F, bool DIF>
__global__ __launch_bounds__(NTT_MAX_THREADS) void k_ntt_pass(F* __restrict__ data, const F* __restrict__ tw, int L, int s0, int k,
                                                          int cb, int ncomp_log, F scale, int do_scale) {
  static_assert(sizeof(F) == 32, "Fr is 8 x u32");
  extern __shared__ uint4 lds_raw[];
  const int cc_log = cb + ncomp_log;
  const int CC = 1 << cc_log;
  const int E = 1 << (k + cc_log);
  uint4* lo = lds_raw;
  uint4* hi = lds_raw + E;
  const int mid_bits = s0 - cb;
  const size_t tile = blockIdx.x;
  const size_t mid = tile & ((size_t(1) << mid_bits) - 1);
  const size_t hi_idx = tile >> mid_bits;
  const int tid = threadIdx.x;
  const int NTT_THREADS = blockDim.x;

  // gather tile: lds index e = t * CC + cc ; global element = (((hi << k | t) << mid_bits | mid) * CC + cc
  for (int e = tid; e < E; e += NTT_THREADS) {
    const int cc = e & (CC - 1);
    const size_t t = e >> cc_log;
    const size_t g = ((((hi_idx << k) | t) << mid_bits) | mid) * CC + cc;
    lds_put<F>(lo, hi, e, data[g]);
  }
  __syncthreads();

  const int half_E = E >> 1;
  for (int qq = 0; qq < k; ++qq) {
    const int q = DIF ? (k - 1 - qq) : qq;  // local stage; global stage s = s0 + q
    const int half = 1 << q;
    const int tw_shift = L - 1 - (s0 + q);
    for (int bidx = tid; bidx < half_E; bidx += NTT_THREADS) {
      const int cc = bidx & (CC - 1);
      const int tb = bidx >> cc_log;
      const int t_lo = tb & (half - 1);
      const int t0 = ((tb >> q) << (q + 1)) | t_lo;
      const int e0 = (t0 << cc_log) | cc;
      const int e1 = e0 + (half << cc_log);
      // i mod 2^s with s = s0+q: (t_lo << s0) | (mid << cb) | col
      const size_t imod = ((size_t)t_lo << s0) | (mid << cb) | (size_t)(cc >> ncomp_log);
      const F w = tw[imod << tw_shift];
      F u = lds_get<F>(lo, hi, e0);
      F v = lds_get<F>(lo, hi, e1);
      if (DIF) {
        F s = F::add(u, v);
        F d = F::mul(F::sub(u, v), w);
        lds_put<F>(lo, hi, e0, s);
        lds_put<F>(lo, hi, e1, d);
      } else {
        F x = F::mul(v, w);
        lds_put<F>(lo, hi, e0, F::add(u, x));
        lds_put<F>(lo, hi, e1, F::sub(u, x));
      }
    }
    __syncthreads();
  }

  for (int e = tid; e < E; e += NTT_THREADS) {
    const int cc = e & (CC - 1);
    const size_t t = e >> cc_log;
    const size_t g = ((((hi_idx << k) | t) << mid_bits) | mid) * CC + cc;
    F f = lds_get<F>(lo, hi, e);
    if (do_scale) f = F::mul(f, scale);
    data[g] = f;
  }
}

// ---- lazy-field pass ---------------------------------------------------------------------------------------------
// Same tiling, butterflies in the signed 9 x 29-bit representation (field29.hpp): an element x*R (arkworks Montgomery,
// R = 2^256) is re-sliced into 29-bit limbs as is, the twiddle comes in the R' = 2^261 domain, so the lazy Montgomery
// product  (x R)(w R') / R' = (x w) R  stays in the arkworks domain with no conversion multiplication. A product costs
// 162 v_mad_i64_i32 instead of ~136 mads + ~300 carry instructions; sums are limb-wise + one parallel carry step. Values
// drift to at most ~(1 + k) p over the k <= 11 stages of a decimation-in-time pass (u' = u + v w: limbs stay normalised,
// the top limb absorbs the growth); in a decimation-in-frequency pass the sum output feeds the next sum and would double
// per stage, so it is folded back below 2p each stage (fold_top: a top-limb quotient estimate, no multiplication). The
// tile is brought back to [0, p) once per pass when it is stored (canonical_wide).
// limb-major LDS tile: limbs (2i, 2i+1) of element e live in an 8-byte slot of pair-array i (ds_read/write_b64, lane
// stride 8 bytes: conflict-free), the odd last limb in a 4-byte array behind them -- 5 LDS accesses per element instead of 9
// Entry idx of a staged twiddle table: limbs 0..7 of the canonical w * R' as the 8 words of slot idx (two 16-byte loads), limb 8
// in an int32 array behind the `slots` slots -- ready to multiply by, no re-slicing per butterfly.
template <class F, class LZ>
__device__ __forceinline__ LZ load_sliced_twiddle(const F* __restrict__ twl, size_t idx, size_t slots) {
  static_assert(LZ::NL == F::N + 1, "one limb more than 32-bit words");
  const F f = twl[idx];
  LZ r;
#pragma unroll
  for (int i = 0; i < F::N; ++i) r.l[i] = (int32_t)f.l[i];
  r.l[F::N] = reinterpret_cast<const int32_t*>(twl + slots)[idx];
  return r;
}

// Bank swizzle of the tile index: entry e lives at slot e ^ ((e >> 2) & 31). A radix-4 round at local stage q reads / writes, per lane, the
// entries t0 + j 2^q: for q = 0 the lanes of a wave touch every FOURTH 8-byte slot (8-way bank conflicts on ds_read_b64, whose 32-lane
// groups need 32 distinct slots mod 32, and on ds_write_b64, whose 16-lane groups need 16 distinct slots mod 16), for q = 1..4 runs of
// 2^q slots 2^(q+2) apart. XOR-ing bits 2..6 of the index into bits 0..4 makes every one of those patterns a bijection onto the slots of
// its lane group (checked exhaustively over GF(2) for all rounds q = 0..9, tile and strided passes, with one or two components per entry;
// only the 32-lane read of the single left-over radix-2 stage at q = 0 keeps a 2-way conflict), while runs of >= 32 consecutive entries
// (tile load / store, late rounds) stay conflict-free: the map permutes entries inside aligned blocks of 128. Two instructions per address.
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of k_ntt_pass_r4 before: 0.28-0.30 (profiles/archive/r03_f_vecops_ntt_pmc_sq.csv).
template <class LZ>
struct LazyLds {
  int32_t* base;
  int E;
  int swz;  // 31, or 0 to switch the swizzle off (A/B runs: tune "ntt_variant" bit 11)
  static constexpr int NP = LZ::NL / 2;
  __device__ __forceinline__ int slot(int e) const { return e ^ ((e >> 2) & swz); }
  __device__ __forceinline__ LZ get(int e0) const {
    const int e = slot(e0);
    LZ r;
    const int2* pairs = reinterpret_cast<const int2*>(base);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int2 v = pairs[i * E + e];
      r.l[2 * i] = v.x;
      r.l[2 * i + 1] = v.y;
    }
    if (LZ::NL & 1) r.l[LZ::NL - 1] = base[2 * NP * E + e];
    return r;
  }
  __device__ __forceinline__ void put(int e0, const LZ& v) const {
    const int e = slot(e0);
    int2* pairs = reinterpret_cast<int2*>(base);
#pragma unroll
    for (int i = 0; i < NP; ++i) pairs[i * E + e] = make_int2(v.l[2 * i], v.l[2 * i + 1]);
    if (LZ::NL & 1) base[2 * NP * E + e] = v.l[LZ::NL - 1];
  }
};

template <class F, class LZ, bool DIF>
__global__ __launch_bounds__(NTT_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(8))) void k_ntt_pass_lazy(F* __restrict__ data, const F* __restrict__ twl, int L, int s0, int k, int cb,
                                                                    int ncomp_log, F scale_lazy, int do_scale, const F* __restrict__ scale_tbl) {
  extern __shared__ uint4 lds_raw[];
  const int cc_log = cb + ncomp_log;
  const int CC = 1 << cc_log;
  const int E = 1 << (k + cc_log);
  const LazyLds<LZ> lds{reinterpret_cast<int32_t*>(lds_raw), E, (do_scale & 0x1000) ? 0 : 31};
  const bool unit_skip = !(do_scale & 0x2000);  // stage 0 of the whole transform has twiddle 1 everywhere: no multiplication (see k_ntt_pass_r4)
  do_scale &= 1;
  const int mid_bits = s0 - cb;
  const size_t tile = blockIdx.x;
  const size_t mid = tile & ((size_t(1) << mid_bits) - 1);
  const size_t hi_idx = tile >> mid_bits;
  const int tid = threadIdx.x;
  const int NT = blockDim.x;

  // Tile load, four entries per lane with all four global loads issued before the first is consumed: written as one rolled loop
  // (load, unpack, put per iteration) every lane paid four SERIAL trips to HBM per tile -- the 0.04-0.05 ms per sweep that no butterfly
  // stage accounts for (profiles/archive/r04_e_ntt_per_pass.log; shortening the store tail by a third changed nothing, r04_l_ntt.log).
  for (int base = 0; base < E; base += 4 * NT) {
    F raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = base + tid + i * NT;
      if (e < E) {
        const int cc = e & (CC - 1);
        const size_t t = e >> cc_log;
        raw[i] = data[((((hi_idx << k) | t) << mid_bits) | mid) * CC + cc];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = base + tid + i * NT;
      if (e < E) lds.put(e, LZ::unpack(raw[i]));
    }
  }
  __syncthreads();

  const int half_E = E >> 1;
  // One butterfly stage over the tile. NORM: run the parallel carry step on the decimation-in-time outputs. x is a fresh
  // product (limbs < 2^B) and u +- x of a normalised u is a two-term sum, still an admissible `a` operand of the next
  // stage's product, so the carry step is only needed every second stage (and not after the last one: the tile leaves
  // through mul / canonical_wide, which take two-term limbs).
  auto stage = [&](int q, auto norm_tag) {
    constexpr bool NORM = decltype(norm_tag)::value;
    const int half = 1 << q;
    const size_t stage_base = (size_t(1) << (s0 + q)) - 1;
    for (int bidx = tid; bidx < half_E; bidx += NT) {
      const int cc = bidx & (CC - 1);
      const int tb = bidx >> cc_log;
      const int t_lo = tb & (half - 1);
      const int t0 = ((tb >> q) << (q + 1)) | t_lo;
      const int e0 = (t0 << cc_log) | cc;
      const int e1 = e0 + (half << cc_log);
      const size_t imod = ((size_t)t_lo << s0) | (mid << cb) | (size_t)(cc >> ncomp_log);
#if defined(CSH_EXPERIMENTS) && defined(CSH_NTT_ABLATE_TW)  // ablation builds only (tools/experiments/gpu_r3_h.sh): one of 64 table entries -- wrong results, no gather
      const LZ w = load_sliced_twiddle<F, LZ>(twl, imod & 63, (size_t(1) << L) - 1);
#else
      const LZ w = load_sliced_twiddle<F, LZ>(twl, stage_base + imod, (size_t(1) << L) - 1);  // staged table: stage s0 + q starts at 2^(s0 + q) - 1
#endif
      const LZ u = lds.get(e0);
      const LZ v = lds.get(e1);
      if (unit_skip && s0 + q == 0) {  // w = 1 for every butterfly of global stage 0 (wave-uniform branch)
        if (DIF) {
          lds.put(e0, LZ::add(u, v).fold_top());
          lds.put(e1, LZ::sub(u, v));            // the pass's last stage: leaves through mul / canonical_wide, which take two-term limbs
        } else {
          lds.put(e0, LZ::add(u, v));            // the pass's first stage: u, v are unpack() outputs, the sums two-term operands as after a product
          lds.put(e1, LZ::sub(u, v));
        }
        continue;
      }
      if (DIF) {
        lds.put(e0, LZ::add(u, v).fold_top());   // sums feed sums here: keep the value within (-p, 2p) every stage
#if defined(CSH_EXPERIMENTS) && defined(CSH_NTT_ABLATE_MUL)
        lds.put(e1, LZ::add(LZ::sub(u, v), w));
#else
        lds.put(e1, LZ::mul(LZ::sub(u, v), w));  // the two-term difference is an admissible product operand as is
#endif
      } else {
#if defined(CSH_EXPERIMENTS) && defined(CSH_NTT_ABLATE_MUL)  // ablation builds only: no multiplication (wrong results): the memory / LDS / barrier floor of the pass
        const LZ x = LZ::add(v, w);
#else
        const LZ x = LZ::mul(v, w);
#endif
        if (NORM) {
          lds.put(e0, LZ::add(u, x).normalized());
          lds.put(e1, LZ::sub(u, x).normalized());
        } else {
          lds.put(e0, LZ::add(u, x));
          lds.put(e1, LZ::sub(u, x));
        }
      }
    }
    __syncthreads();
  };
  for (int qq = 0; qq < k; ++qq) {
    const int q = DIF ? (k - 1 - qq) : qq;
    if (DIF || (qq & 1))
      stage(q, std::true_type{});
    else
      stage(q, std::false_type{});
  }

  const LZ sc = LZ::unpack(scale_lazy);
  for (int e = tid; e < E; e += NT) {
    const int cc = e & (CC - 1);
    const size_t t = e >> cc_log;
    const size_t g = ((((hi_idx << k) | t) << mid_bits) | mid) * CC + cc;
    LZ f = lds.get(e);
    // the 1/n of the inverse transform, or -- witness maps -- a per-entry table that already carries it (1/n times the coset
    // power of the entry, in the pass's output order): the coset shift costs no sweep and no multiplication of its own
    if (do_scale) f = LZ::mul(f, scale_tbl ? LZ::unpack(scale_tbl[g >> ncomp_log]) : sc);
    data[g] = f.canonical_wide().pack();
  }
}

// ---- radix-4 form of the lazy pass ---------------------------------------------------------------------------------
// Two butterfly stages per LDS round trip: a lane owns the four tile entries that stages (q, q + 1) connect, keeps them in
// registers across both stages and touches LDS once per pair of stages (half the ds traffic, address arithmetic, waits and
// barriers of the radix-2 loop; three twiddle loads per four butterflies instead of four: the second stage's two twiddles are
// w and w * omega^(n/4), both straight from the table). 512 lanes per 2048-entry tile (one radix-4 unit per lane and round), two
// tiles per CU, up to 128 VGPRs per lane. The arithmetic per element is the radix-2 pass's, operation for operation (same
// operand classes: a round starts from normalised values, the second stage multiplies two-term sums, the outputs are normalised
// once; in decimation in frequency every sum is folded), so the limb-bound contracts checked by the host self-test carry over.
// An odd stage count leaves one radix-2 stage (last in DIT, last = local stage 0 in DIF).
// PAIR (round 6; DIF must be true, s0 = 0): the LAST pass of an inverse transform (decimation in frequency, table `twl`) and the FIRST
// pass of the forward transform that follows it in a witness map (decimation in time, table `twl_fwd`) work on the same contiguous
// tiles -- ifft_in_to_out, coset table, fft_out_to_in of reduction.rs:141-174 -- so one launch keeps the tile in LDS across both:
// DIF rounds, the scale (1/n, or the per-entry table carrying 1/n times the coset power) and the canonical form a store + load would
// have left, DIT rounds, one store. One HBM sweep, one launch and one LDS fill less per pair; the field elements are the same.
template <class F, class LZ, bool DIF, bool PAIR = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ntt_pass_r4(F* __restrict__ data, const F* __restrict__ twl, int L, int s0, int k,
                                                                                             int cb, int ncomp_log, F scale_lazy, int do_scale, const F* __restrict__ scale_tbl,
                                                                                             const F* __restrict__ twl_fwd = nullptr) {
  static_assert(!PAIR || DIF, "the pair kernel starts with the decimation-in-frequency half");
  extern __shared__ uint4 lds_raw[];
  const int cc_log = cb + ncomp_log;
  const int CC = 1 << cc_log;
  const int E = 1 << (k + cc_log);
  const LazyLds<LZ> lds{reinterpret_cast<int32_t*>(lds_raw), E, (do_scale & 0x1000) ? 0 : 31};
  const int mid_bits = s0 - cb;
  const size_t tile = blockIdx.x;
  const size_t mid = tile & ((size_t(1) << mid_bits) - 1);
  const size_t hi_idx = tile >> mid_bits;
  const int tid = threadIdx.x;
  const int NT = blockDim.x;
  // timing experiments only (tune "ntt_variant" bits 16-19, wrong results): 1 = no butterfly rounds, 2 = no global loads, 4 = no stores,
  // 8 = no canonicalisation before the store
  const int ablate = kExperiments ? (do_scale >> 8) & 0xf : 0;  // product builds: constant 0, the branches below fold away
  const bool unit_skip = !(do_scale & 0x2000);  // tune "ntt_variant" bit 20 switches the unit-twiddle rounds off (A/B runs)
  do_scale &= 1;

  // Tile load, four entries per lane with all four global loads issued before the first is consumed: written as one rolled loop
  // (load, unpack, put per iteration) every lane paid four SERIAL trips to HBM per tile -- the 0.04-0.05 ms per sweep that no butterfly
  // stage accounts for (profiles/archive/r04_e_ntt_per_pass.log; shortening the store tail by a third changed nothing, r04_l_ntt.log).
  for (int base = 0; base < E; base += 4 * NT) {
    F raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = base + tid + i * NT;
      if (e < E) {
        const int cc = e & (CC - 1);
        const size_t t = e >> cc_log;
        if (ablate & 2) raw[i] = scale_lazy; else
        raw[i] = data[((((hi_idx << k) | t) << mid_bits) | mid) * CC + cc];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = base + tid + i * NT;
      if (e < E) lds.put(e, LZ::unpack(raw[i]));
    }
  }
  __syncthreads();

  // twiddle of global stage s0 + q for the pair whose lower index has local stage bits t_lo (< 2^q)
  auto twiddle = [&](const F* __restrict__ tw, int q, int t_lo, int cc) {
    const size_t imod = ((size_t)t_lo << s0) | (mid << cb) | (size_t)(cc >> ncomp_log);
    return load_sliced_twiddle<F, LZ>(tw, ((size_t(1) << (s0 + q)) - 1) + imod, (size_t(1) << L) - 1);  // staged table
  };
  // stages (q, q + 1) on the four entries t0 + {0, 1, 2, 3} * 2^q of every unit
  auto round4 = [&](auto dif_c, const F* __restrict__ tw, int q) {
    constexpr bool D = decltype(dif_c)::value;
    const int quarter_E = E >> 2;
    for (int u = tid; u < quarter_E; u += NT) {
      const int cc = u & (CC - 1);
      const int tb = u >> cc_log;
      const int t_lo = tb & ((1 << q) - 1);
      const int t0 = ((tb >> q) << (q + 2)) | t_lo;
      const int e0 = (t0 << cc_log) | cc;
      const int st = 1 << (q + cc_log);
      LZ x0 = lds.get(e0), x1 = lds.get(e0 + st), x2 = lds.get(e0 + 2 * st), x3 = lds.get(e0 + 3 * st);
      if constexpr (D) {
        // Value / limb bounds of a round (inputs: normalised limbs, values within (-p, 2.2 p)): the first stage's sums stay below
        // 4.4 p and only take the parallel carry step (their differences, < 6.5 p in magnitude with two-term limbs, are admissible
        // product operands: the contract is |value| < 8 p); the second stage's sum of sums (< 8.8 p) is folded below 2 p; the sum of
        // the two fresh products (< 2.2 p) again only takes the carry step. Two folds per round instead of four.
        {  // stage q + 1: (x0, x2) with w, (x1, x3) with w * omega^(n/4)
          const LZ w2 = twiddle(tw, q + 1, t_lo, cc), w3 = twiddle(tw, q + 1, t_lo + (1 << q), cc);
          LZ d02, d13;  // the two products of a fold are independent: their multiply-adds alternate (FpS::mul2)
          LZ::mul2(LZ::sub(x0, x2), w2, LZ::sub(x1, x3), w3, d02, d13);
          x0 = LZ::add(x0, x2).normalized();
          x1 = LZ::add(x1, x3).normalized();
          x2 = d02;
          x3 = d13;
        }
        {  // stage q: (x0, x1), (x2, x3), one twiddle
          const LZ w1 = twiddle(tw, q, t_lo, cc);
          LZ d01, d23;
          LZ::mul2(LZ::sub(x0, x1), w1, LZ::sub(x2, x3), w1, d01, d23);
          x0 = LZ::add(x0, x1).fold_top();
          x2 = LZ::add(x2, x3).normalized();
          x1 = d01;
          x3 = d23;
        }
      } else {
        {  // stage q: inputs normalised, outputs two-term sums (admissible product operands as they are)
          const LZ w1 = twiddle(tw, q, t_lo, cc);
          LZ p1, p3;
          LZ::mul2(x1, w1, x3, w1, p1, p3);
          x1 = LZ::sub(x0, p1);
          x0 = LZ::add(x0, p1);
          x3 = LZ::sub(x2, p3);
          x2 = LZ::add(x2, p3);
        }
        {  // stage q + 1: the carry step runs once per round, on the outputs
          const LZ w2 = twiddle(tw, q + 1, t_lo, cc), w3 = twiddle(tw, q + 1, t_lo + (1 << q), cc);
          LZ p2, p3;
          LZ::mul2(x2, w2, x3, w3, p2, p3);
          x2 = LZ::sub(x0, p2).normalized();
          x0 = LZ::add(x0, p2).normalized();
          x3 = LZ::sub(x1, p3).normalized();
          x1 = LZ::add(x1, p3).normalized();
        }
      }
      lds.put(e0, x0);
      lds.put(e0 + st, x1);
      lds.put(e0 + 2 * st, x2);
      lds.put(e0 + 3 * st, x3);
    }
    __syncthreads();
  };
  // Global stages 0 and 1 of the whole transform (the pass with s0 = 0, local round q = 0): stage 0 has twiddle 1 everywhere, stage 1 has
  // 1 on its first pair and omega^(n/4) -- one table entry, the same for every lane -- on its second: ONE multiplication per unit
  // instead of four (3 of the 22 x 2 multiplications a 4-entry unit meets in a 2^22 transform). Limb / value bounds: decimation in time
  // meets this round first, on unpack() outputs (limbs 0..NL-2 in [0, 2^B), value in [0, p)): the four-term sum x0 + x1 + x2 + x3 stays
  // below 2^(B+2) <= 2^31 - 4 per limb and below 4 p, what the general round reaches with three fresh products. Decimation in frequency
  // meets it last, on normalised values within (-p, 2.2 p): sums are carried / folded stage by stage as in the general round; the two
  // differences that are no longer passed through a multiplication stay two- or (carried) three-term sums within (-6.4 p, 6.4 p) and
  // leave through the tail's mul / canonical_wide, whose contracts (two-term limbs, |value| < 8 p resp. 32 p) they meet.
  auto round4_unit = [&](auto dif_c, const F* __restrict__ tw) {
    constexpr bool D = decltype(dif_c)::value;
    const int quarter_E = E >> 2;
    const LZ w4 = twiddle(tw, 1, 1, 0);  // omega^(n/4) (forward table) / its inverse (inverse table): staged entry 2
    for (int u = tid; u < quarter_E; u += NT) {
      const int cc = u & (CC - 1);
      const int tb = u >> cc_log;
      const int e0 = ((tb << 2) << cc_log) | cc;
      const int st = 1 << cc_log;
      LZ x0 = lds.get(e0), x1 = lds.get(e0 + st), x2 = lds.get(e0 + 2 * st), x3 = lds.get(e0 + 3 * st);
      if constexpr (D) {
        const LZ d13 = LZ::mul(LZ::sub(x1, x3), w4);   // stage 1, second pair
        const LZ d02 = LZ::sub(x0, x2);                // stage 1, first pair: times 1
        x0 = LZ::add(x0, x2).normalized();
        x1 = LZ::add(x1, x3).normalized();
        const LZ o1 = LZ::sub(x0, x1);                 // stage 0: times 1
        x0 = LZ::add(x0, x1).fold_top();
        x1 = o1;
        x2 = LZ::add(d02, d13).normalized();
        x3 = LZ::sub(d02, d13).normalized();
      } else {
        const LZ a0 = LZ::add(x0, x1), a1 = LZ::sub(x0, x1), a2 = LZ::add(x2, x3), a3 = LZ::sub(x2, x3);  // stage 0: times 1
        const LZ p3 = LZ::mul(a3, w4);                                                                  // stage 1, second pair
        x0 = LZ::add(a0, a2).normalized();
        x2 = LZ::sub(a0, a2).normalized();
        x1 = LZ::add(a1, p3).normalized();
        x3 = LZ::sub(a1, p3).normalized();
      }
      lds.put(e0, x0);
      lds.put(e0 + st, x1);
      lds.put(e0 + 2 * st, x2);
      lds.put(e0 + 3 * st, x3);
    }
    __syncthreads();
  };
  // one radix-2 stage (odd stage counts): the radix-2 pass's butterfly; in DIT it is the pass's last stage (outputs leave
  // through mul / canonical_wide, which take two-term limbs), in DIF its inputs come normalised out of the last round
  auto stage2 = [&](auto dif_c, const F* __restrict__ tw, int q) {
    constexpr bool D = decltype(dif_c)::value;
    const int half_E = E >> 1;
    const int half = 1 << q;
    for (int bidx = tid; bidx < half_E; bidx += NT) {
      const int cc = bidx & (CC - 1);
      const int tb = bidx >> cc_log;
      const int t_lo = tb & (half - 1);
      const int t0 = ((tb >> q) << (q + 1)) | t_lo;
      const int e0 = (t0 << cc_log) | cc;
      const int e1 = e0 + (half << cc_log);
      const LZ u = lds.get(e0);
      const LZ v = lds.get(e1);
      if (unit_skip && s0 + q == 0) {  // global stage 0: twiddle 1 everywhere (wave-uniform branch)
        lds.put(e0, D ? LZ::add(u, v).fold_top() : LZ::add(u, v));
        lds.put(e1, LZ::sub(u, v));
        continue;
      }
      const LZ w = twiddle(tw, q, t_lo, cc);
      if constexpr (D) {
        lds.put(e0, LZ::add(u, v).fold_top());
        lds.put(e1, LZ::mul(LZ::sub(u, v), w));
      } else {
        const LZ x = LZ::mul(v, w);
        lds.put(e0, LZ::add(u, x));
        lds.put(e1, LZ::sub(u, x));
      }
    }
    __syncthreads();
  };
  const bool unit_round = unit_skip && s0 == 0 && k >= 2;
  auto rounds = [&](auto dif_c, const F* __restrict__ tw) {
    constexpr bool D = decltype(dif_c)::value;
    if constexpr (D) {
      int q = k;
      for (; q >= 2; q -= 2) {
        if (q == 2 && unit_round) round4_unit(dif_c, tw);
        else round4(dif_c, tw, q - 2);
      }
      if (q == 1) stage2(dif_c, tw, 0);
    } else {
      int q = 0;
      for (; q + 2 <= k; q += 2) {
        if (q == 0 && unit_round) round4_unit(dif_c, tw);
        else round4(dif_c, tw, q);
      }
      if (q < k) stage2(dif_c, tw, q);
    }
  };
  const LZ sc = LZ::unpack(scale_lazy);
  if constexpr (PAIR) {
    rounds(std::true_type{}, twl);
    // between the two transforms: what the inverse transform's last pass would have stored (scaled, canonical) and the forward
    // transform's first pass would have loaded and unpacked -- each lane on the entries it will store, no traffic
    for (int e = tid; e < E; e += NT) {
      const size_t g = (size_t)tile * (size_t)E + (size_t)e;  // s0 = 0, cb = 0: the tile is contiguous
      const LZ f = LZ::mul(lds.get(e), scale_tbl ? LZ::unpack(scale_tbl[g >> ncomp_log]) : sc);
      lds.put(e, LZ::unpack(f.canonical_wide().pack()));
    }
    __syncthreads();
    rounds(std::false_type{}, twl_fwd);
    for (int e = tid; e < E; e += NT) data[(size_t)tile * (size_t)E + (size_t)e] = lds.get(e).canonical_wide().pack();
    return;
  }
  if (ablate & 1) {
  } else {
    rounds(std::integral_constant<bool, DIF>{}, twl);
  }

  for (int e = tid; e < E; e += NT) {
    const int cc = e & (CC - 1);
    const size_t t = e >> cc_log;
    const size_t g = ((((hi_idx << k) | t) << mid_bits) | mid) * CC + cc;
    LZ f = lds.get(e);
    // the 1/n of the inverse transform, or -- witness maps -- a per-entry table that already carries it (1/n times the coset
    // power of the entry, in the pass's output order): the coset shift costs no sweep and no multiplication of its own
    if (do_scale) f = LZ::mul(f, scale_tbl ? LZ::unpack(scale_tbl[g >> ncomp_log]) : sc);
    if (ablate & 4) continue;
    data[g] = (ablate & 8) ? f.pack() : f.canonical_wide().pack();
  }
}

// out[i] = in[i] re-encoded from x * 2^256 to packed canonical x * R'
// Twiddle table of the lazy passes, STAGED: stage s (half-size m = 2^s) reads w^(j * n / 2m), j < 2^s, i.e. every 2^(L-1-s)-th
// entry of the natural table -- for the late stages of a tile, 64 lanes of a wave gathered 64 cache lines up to 64 KiB apart. Here
// every stage has its own contiguous run: out[(2^s - 1) + j] = storage form of in[j << (L - 1 - s)], 2^L - 1 entries in all (twice
// the natural table), so that neighbouring lanes (neighbouring j) read neighbouring 32-byte entries.
template <class F, class LZ>
__global__ __launch_bounds__(256) void k_to_lazy_table(const F* __restrict__ in, F* __restrict__ out, int L) {
  const size_t total = (size_t(1) << L) - 1;
  int32_t* top = reinterpret_cast<int32_t*>(out + total);  // limb 8 of every entry, behind the slots of limbs 0..7
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int s = 63 - __clzll((unsigned long long)(i + 1));  // stage: 2^s - 1 <= i < 2^(s+1) - 1
    const size_t j = i + 1 - (size_t(1) << s);
    const LZ v = LZ::from_fp(in[j << (L - 1 - s)]).canonical();
    F f;
#pragma unroll
    for (int k = 0; k < F::N; ++k) f.l[k] = (uint32_t)v.l[k];
    out[i] = f;
    top[i] = v.l[F::N];
  }
}

__device__ __forceinline__ uint32_t bitrev_n(uint32_t i, int log_n) { return log_n == 0 ? 0 : (__brev(i) >> (32 - log_n)); }

// in-place bit-reversal permutation of entries (ncomp elements each)
template <class F>
__global__ __launch_bounds__(256) void k_bit_reverse(F* data, int log_n, int ncomp) {
  const size_t n = size_t(1) << log_n;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t j = bitrev_n((uint32_t)i, log_n);
    if (i < j) {
      for (int c = 0; c < ncomp; ++c) {
        F a = data[i * ncomp + c];
        F b = data[j * ncomp + c];
        data[i * ncomp + c] = b;
        data[j * ncomp + c] = a;
      }
    }
  }
}

constexpr int POW_CHUNK = 32;
// out[perm(i)] = base^i, i < n; each thread produces POW_CHUNK consecutive powers
template <class F, bool BITREV>
__global__ __launch_bounds__(256) void k_powers(F* out, F base, size_t n, int log_n) {
  const size_t chunks = (n + POW_CHUNK - 1) / POW_CHUNK;
  for (size_t c = blockIdx.x * (size_t)256 + threadIdx.x; c < chunks; c += (size_t)gridDim.x * 256) {
    const size_t i0 = c * POW_CHUNK;
    F cur = F::pow_u64(base, (uint64_t)i0);
    for (int k = 0; k < POW_CHUNK; ++k) {
      const size_t i = i0 + k;
      if (i >= n) break;
      const size_t o = BITREV ? (size_t)bitrev_n((uint32_t)i, log_n) : i;
      out[o] = cur;
      cur = F::mul(cur, base);
    }
  }
}

// out[bitrev(i)] = storage form (packed canonical x * R', what the lazy passes multiply by) of scale * base^i: the coset table of
// a witness map with the inverse transform's 1/n folded in
template <class F, class LZ>
__global__ __launch_bounds__(256) void k_powers_lazy_scaled(F* out, F base, F scale, size_t n, int log_n) {
  const size_t chunks = (n + POW_CHUNK - 1) / POW_CHUNK;
  for (size_t c = blockIdx.x * (size_t)256 + threadIdx.x; c < chunks; c += (size_t)gridDim.x * 256) {
    const size_t i0 = c * POW_CHUNK;
    F cur = F::mul(F::pow_u64(base, (uint64_t)i0), scale);
    for (int k = 0; k < POW_CHUNK; ++k) {
      const size_t i = i0 + k;
      if (i >= n) break;
      out[bitrev_n((uint32_t)i, log_n)] = LZ::repack_for_storage(cur);
      cur = F::mul(cur, base);
    }
  }
}

// ---- host-side helpers -----------------------------------------------------------------------------
template <class F>
static F f_from_words(const void* p) {
  F f;
  memcpy(&f, p, sizeof(F));
  return f;
}

struct Pass {
  int s0, k, cb;
};

// log2(field elements per LDS tile) for a transform of 2^L entries of 2^ncomp_log elements. The tile is the unit of parallelism (one
// workgroup each): 2^11-element tiles give a 2^16-point transform 32 workgroups for 256 CUs and make every workgroup walk 11 dependent
// stages, so small transforms were latency-bound on a handful of CUs. Measured, interleaved (profiles/archive/r04_o_ntt_small.log, ms per
// transform, 2^11-element tiles -> best): 2^12 0.038 -> 0.015 (2^8), 2^14 0.043 -> 0.017 (2^8), 2^16 0.048 -> 0.021 (2^9), 2^17
// 0.052 -> 0.028 (2^9), 2^18 0.058 -> 0.039 (2^10); 2^19 .. 2^21 are best at 2^11 (more passes cost more than the parallelism gains);
// from 2^22 elements on 2^10-element tiles (four independent tiles per CU instead of two: their load / compute / store phases
// interleave) gain 1-4 % (r04_d_ntt_tile_ab.log, r04_k_ntt_tiles.log). tune "ntt_variant" bits 8-10 = v forces 2^(12-v) (A/B runs).
static int ntt_tile_log(int L, int ncomp_log) {
  const int v = (tune().ntt_variant.load(std::memory_order_relaxed) >> 8) & 7;
  if (v >= 1 && v <= 4) return 12 - v;
  const int total = L + ncomp_log;
  if (total <= 15) return 8;
  if (total <= 17) return 9;
  if (total == 18) return 10;
  if (total >= 22) return 10;
  return NTT_TILE_LOG;
}

static int plan_passes(int L, int ncomp_log, Pass* out) {
  const int TE = ntt_tile_log(L, ncomp_log) - ncomp_log;  // log2(entries per tile)
  // Contiguous run of a strided pass: 2^cb entries of ncomp x 32 bytes. 64-byte runs cost nothing against 256-byte ones and save a whole
  // sweep where the stages then fit two passes: 2^20 points 11 + 9 stages (128-byte runs) instead of 11 + 5 + 4, inverse / forward 0.143 / 0.135 ->
  // 0.132 / 0.123 ms, share pairs 0.258 / 0.259 -> 0.246 / 0.242 ms; 2^21 points 11 + 10 (64-byte runs) 0.268 / 0.285 -> 0.261 / 0.270
  // (profiles/archive/r03_q_ntt_runs.log, r03_r_ntt_tiles.log, r03_s_ntt_runs64.log).
  // 32-byte runs do not pay (2^22 as 11 + 11: 0.53 -> 0.60 ms), nor do 2^12-entry tiles at one tile per CU (2^22 as 12 + 10: 0.53 -> 0.55,
  // 2^24 2.2 -> 2.4 ms). tune "ntt_variant" bits 4-6 = log2(run entries) + 1 overrides the run length for A/B runs (0: this default).
  const int nv_run = (tune().ntt_variant.load(std::memory_order_relaxed) >> 4) & 7;
  const int cb_min = nv_run ? (nv_run - 1 > ncomp_log ? nv_run - 1 - ncomp_log : 0) : (1 - ncomp_log);
  int np = 0;
  int k0 = L < TE ? L : TE;
  out[np++] = {0, k0, 0};
  int rem = L - k0;
  if (rem > 0) {
    const int kmax = TE - cb_min;
    const int npass = (rem + kmax - 1) / kmax;
    int s0 = k0;
    for (int i = 0; i < npass; ++i) {
      int k = rem / (npass - i);
      if (rem % (npass - i)) ++k;
      int cb = TE - k;
      if (cb > s0) cb = s0;
      out[np++] = {s0, k, cb};
      s0 += k;
      rem -= k;
    }
  }
  return np;
}

// the natural-order twiddle tables of the 32-bit pass, (re)built when that pass is asked for on a domain that has released them
template <class F>
static int plain_tables(const Domain* cd) {
  // A Domain is shared across host threads (the mirror's DomainCache, the shim's Arc<HipDomain>): the pointers are only ever read or
  // written under this mutex (the former unlocked "already built?" test raced with the thread building them). Only the
  // CSH_NTT_LAZY = 0 A/B path comes here.
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  Domain* d = const_cast<Domain*>(cd);
  if (d->tw_fwd && d->tw_inv) return CSH_OK;
  const size_t half = d->n / 2 ? d->n / 2 : 1;
  int prev_dev = -1;
  (void)hipGetDevice(&prev_dev);
  const bool switch_dev = prev_dev != d->device;
  if (switch_dev && hipSetDevice(d->device) != hipSuccess) {  // the tables live where the domain lives, whoever asks first
    set_error("hipSetDevice(%d) for the domain's twiddle tables failed", d->device);
    return CSH_ERR_HIP;
  }
  void *f = nullptr, *i = nullptr;
  hipStream_t st = nullptr;
  hipError_t e = hipMalloc(&f, half * sizeof(F));
  if (e == hipSuccess) e = hipMalloc(&i, half * sizeof(F));
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);  // a stream of the domain's device, not the caller's lane
  if (e == hipSuccess) {
    hipLaunchKernelGGL((k_powers<F, false>), dim3(grid_for((half + POW_CHUNK - 1) / POW_CHUNK, 256)), dim3(256), 0, st, (F*)f, f_from_words<F>(d->gen), half, 0);
    hipLaunchKernelGGL((k_powers<F, false>), dim3(grid_for((half + POW_CHUNK - 1) / POW_CHUNK, 256)), dim3(256), 0, st, (F*)i, f_from_words<F>(d->gen_inv), half, 0);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  if (st) (void)hipStreamDestroy(st);
  if (e != hipSuccess) {
    if (f) (void)hipFree(f);
    if (i) (void)hipFree(i);
  }
  if (switch_dev && prev_dev >= 0) (void)hipSetDevice(prev_dev);
  if (e != hipSuccess) {
    set_error("building the natural-order twiddle tables failed: %s", hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? CSH_ERR_OOM : CSH_ERR_HIP;
  }
  d->tw_fwd = f;
  d->tw_inv = i;
  return CSH_OK;
}
// the tables of the 32-bit pass, read under plain_tables' rules: (fwd, inv) once they exist
template <class F>
static int plain_tables_get(const Domain* d, const void** fwd, const void** inv) {
  CSH_TRY(plain_tables<F>(d));  // takes the mutex: returns at once when both exist; the pointers never change afterwards (freed with the domain)
  *fwd = d->tw_fwd;
  *inv = d->tw_inv;
  return CSH_OK;
}

template <class F>
static int run_ntt(const Domain* d, F* data, uint32_t ncomp, bool dif, hipStream_t st, const F* scale_tbl = nullptr, bool skip_pass0 = false) {
  const int L = (int)d->log_n;
  if (L == 0) return CSH_OK;  // size-1 transform is the identity (1/n = 1)
  const int ncomp_log = ncomp == 2 ? 1 : 0;
  Pass passes[8];
  const int np = plan_passes(L, ncomp_log, passes);
  const bool use_lazy = tune().ntt_lazy.load(std::memory_order_relaxed) != 0;
  if (scale_tbl && (!use_lazy || !dif)) {
    set_error("a scale table needs the lazy decimation-in-frequency passes");
    return CSH_ERR_INVALID;
  }
  using LZ = typename LazyOf<F>::type;
  const void *pt_fwd = nullptr, *pt_inv = nullptr;
  if (!use_lazy) CSH_TRY(plain_tables_get<F>(d, &pt_fwd, &pt_inv));
  const F* tw = reinterpret_cast<const F*>(use_lazy ? (dif ? d->tw_inv_lazy : d->tw_fwd_lazy) : (dif ? pt_inv : pt_fwd));
  F scale = f_from_words<F>(use_lazy ? d->n_inv_lazy : d->n_inv);
  const int NTT_THREADS_MAX = [] {
    const int v = tune().ntt_threads.load(std::memory_order_relaxed);  // 16 waves per 2^11-element tile: measured best for both field representations
    return (v == 256 || v == 512 || v == 1024) ? v : 1024;
  }();
  const int only_pass = kExperiments ? (tune().ntt_variant.load(std::memory_order_relaxed) >> 12) & 3 : 0;  // timing experiments (-DCSH_EXPERIMENTS only): run ONE pass of the plan (wrong results)
  for (int pi = 0; pi < np; ++pi) {
    const Pass& p = dif ? passes[np - 1 - pi] : passes[pi];
    if (only_pass && (dif ? np - 1 - pi : pi) != only_pass - 1) continue;
    if (skip_pass0 && p.s0 == 0) continue;  // run_ntt_pair: the pass over the contiguous tiles runs fused with its neighbour of the other direction
    const int tile_log = p.k + p.cb;  // entries
    const size_t tiles = d->n >> tile_log;
    // radix-2 passes: one lane per butterfly of a stage, at least one wave
    const size_t bfly = (size_t(1) << (tile_log + ncomp_log)) >> 1;
    const int NTT_THREADS = bfly >= (size_t)NTT_THREADS_MAX ? NTT_THREADS_MAX : (bfly >= 64 ? (int)bfly : 64);
    const size_t lds_bytes = use_lazy ? (size_t(4 * LZ::NL) << (tile_log + ncomp_log)) : (size_t(32) << (tile_log + ncomp_log));
    const int do_scale = dif && (p.s0 == 0);
    const int noswz = ((tune().ntt_variant.load(std::memory_order_relaxed) & 0x800) ? 0x1000 : 0) |        // LDS bank swizzle off (A/B runs)
                      ((tune().ntt_variant.load(std::memory_order_relaxed) & 0x100000) ? 0x2000 : 0);   // unit-twiddle rounds off (A/B runs)
    if (use_lazy) {
      if (lds_bytes > 48 * 1024)
        CSH_TRY(raise_lds_limit(dif ? (const void*)k_ntt_pass_lazy<F, LZ, true> : (const void*)k_ntt_pass_lazy<F, LZ, false>, (size_t)(4 * LZ::NL) << NTT_TILE_LOG));
      // Radix-4 pass (two stages per LDS round trip, 512 lanes per tile) or radix-2 pass. With the staged twiddle tables and two
      // folds per decimation-in-frequency round the radix-4 pass wins from 2^20 points on (one box, inverse / forward ms: 2^22
      // 0.583 / 0.541 -> 0.553 / 0.526, 2^24 2.36 / 2.04 -> 2.31 / 2.06, 2^20 0.153 / 0.142 -> 0.150 / 0.141; another box 2^22 forward
      // 0.549 -> 0.510) and ties or loses below (profiles/archive/r03_k_ntt_staged.log, r03_l_ntt_mix.log). Default: transforms of >= 2^20
      // points take it; tune "ntt_variant" 1 forces it everywhere, 2 forces the radix-2 pass.
      const int nv = tune().ntt_variant.load(std::memory_order_relaxed);
      const bool r4_default = L >= 20;
      const bool r4 = ((nv & 1) != 0 || (r4_default && (nv & 2) == 0)) && p.k >= 2 && tile_log + ncomp_log >= 2;
      if (r4) {
        if (lds_bytes > 48 * 1024)
          CSH_TRY(raise_lds_limit(dif ? (const void*)k_ntt_pass_r4<F, LZ, true> : (const void*)k_ntt_pass_r4<F, LZ, false>, (size_t)(4 * LZ::NL) << NTT_TILE_LOG));
        const size_t units = (size_t(1) << (tile_log + ncomp_log)) >> 2;
        const int nt = units >= 512 ? 512 : (units >= 64 ? (int)units : 64);
        if (dif)
          hipLaunchKernelGGL((k_ntt_pass_r4<F, LZ, true>), dim3((unsigned)tiles), dim3(nt), lds_bytes, st, data, tw, L, p.s0, p.k, p.cb, ncomp_log, scale, do_scale | (((nv >> 16) & 0xf) << 8) | noswz, scale_tbl);
        else
          hipLaunchKernelGGL((k_ntt_pass_r4<F, LZ, false>), dim3((unsigned)tiles), dim3(nt), lds_bytes, st, data, tw, L, p.s0, p.k, p.cb, ncomp_log, scale, (((nv >> 16) & 0xf) << 8) | noswz, (const F*)nullptr);
        CSH_HIP(hipGetLastError());
        continue;
      }
      if (dif)
        hipLaunchKernelGGL((k_ntt_pass_lazy<F, LZ, true>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, st, data, tw, L, p.s0, p.k, p.cb,
                           ncomp_log, scale, do_scale | noswz, scale_tbl);
      else
        hipLaunchKernelGGL((k_ntt_pass_lazy<F, LZ, false>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, st, data, tw, L, p.s0, p.k, p.cb,
                           ncomp_log, scale, noswz, (const F*)nullptr);
      CSH_HIP(hipGetLastError());
      continue;
    }
    if (lds_bytes > 48 * 1024)
      CSH_TRY(raise_lds_limit(dif ? (const void*)k_ntt_pass<F, true> : (const void*)k_ntt_pass<F, false>, (size_t)32 << NTT_TILE_LOG));
    if (dif)
      hipLaunchKernelGGL((k_ntt_pass<F, true>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, st, data, tw, L, p.s0, p.k, p.cb,
                         ncomp_log, scale, do_scale);
    else
      hipLaunchKernelGGL((k_ntt_pass<F, false>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds_bytes, st, data, tw, L, p.s0, p.k, p.cb,
                         ncomp_log, scale, 0);
    CSH_HIP(hipGetLastError());
  }
  return CSH_OK;
}

template <class F>
static int run_bit_reverse(F* data, uint32_t log_n, uint32_t ncomp, hipStream_t st) {
  if (log_n == 0) return CSH_OK;
  hipLaunchKernelGGL(k_bit_reverse<F>, dim3(grid_for(size_t(1) << log_n, 256)), dim3(256), 0, st, data, (int)log_n, (int)ncomp);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}

template <class F>
static int create_domain_t(csh_curve_t curve, uint32_t log_n, const uint64_t* gen_words, uint64_t ark_generator, Domain** out) {
  if (log_n > (uint32_t)F::Params::TWO_ADICITY) {
    set_error("Polynomial Degree too large");
    return CSH_ERR_DOMAIN;
  }
  F gen;
  if (gen_words) {
    gen = f_from_words<F>(gen_words);
  } else {
    // GENERATOR^TRACE has order 2^TWO_ADICITY; TRACE = (p-1) >> TWO_ADICITY
    uint32_t tr[F::N];
    uint32_t pm1[F::N];
    for (int i = 0; i < F::N; ++i) pm1[i] = F::Params::MOD[i];
    pm1[0] -= 1;  // p is odd
    const int s = F::Params::TWO_ADICITY;
    for (int i = 0; i < F::N; ++i) {
      const int wi = i + s / 32;
      const uint64_t lo = wi < F::N ? pm1[wi] : 0;
      const uint64_t hi2 = wi + 1 < F::N ? pm1[wi + 1] : 0;
      tr[i] = (uint32_t)((lo | (hi2 << 32)) >> (s % 32));
    }
    F g = F::from_u64(ark_generator);
    gen = F::pow_limbs(g, tr, F::N);
    for (uint32_t i = log_n; i < (uint32_t)s; ++i) gen = F::sqr(gen);
  }
  // validate order: gen^(n) == 1 and gen^(n/2) != 1
  {
    F t = gen;
    for (uint32_t i = 0; i + 1 < log_n; ++i) t = F::sqr(t);
    F one = F::one();
    if (log_n >= 1) {
      if (t == one) {
        set_error("group_gen does not have order 2^%u", log_n);
        return CSH_ERR_DOMAIN;
      }
      t = F::sqr(t);
    }
    if (!(t == one)) {
      set_error("group_gen does not have order 2^%u", log_n);
      return CSH_ERR_DOMAIN;
    }
  }
  Domain* d = new Domain();
  d->curve = curve;
  if (hipGetDevice(&d->device) != hipSuccess) d->device = 0;
  d->log_n = log_n;
  d->n = size_t(1) << log_n;
  F gen_inv = F::inv(gen);
  F n_inv = F::inv(F::from_u64((uint64_t)d->n));
  memcpy(d->gen, &gen, 32);
  memcpy(d->gen_inv, &gen_inv, 32);
  memcpy(d->n_inv, &n_inv, 32);
  const size_t half = d->n / 2 ? d->n / 2 : 1;
  d->tw_fwd = d->tw_inv = nullptr;
  hipError_t e1 = hipMalloc(&d->tw_fwd, half * sizeof(F));
  hipError_t e2 = hipMalloc(&d->tw_inv, half * sizeof(F));
  if (e1 != hipSuccess || e2 != hipSuccess) {
    if (d->tw_fwd) (void)hipFree(d->tw_fwd);
    if (d->tw_inv) (void)hipFree(d->tw_inv);
    delete d;
    set_error("hipMalloc of twiddle tables failed");
    return CSH_ERR_OOM;
  }
  hipStream_t st = resolve_stream(nullptr);
  hipLaunchKernelGGL((k_powers<F, false>), dim3(grid_for((half + POW_CHUNK - 1) / POW_CHUNK, 256)), dim3(256), 0, st, (F*)d->tw_fwd, gen, half, 0);
  hipLaunchKernelGGL((k_powers<F, false>), dim3(grid_for((half + POW_CHUNK - 1) / POW_CHUNK, 256)), dim3(256), 0, st, (F*)d->tw_inv, gen_inv, half, 0);
  using LZ = typename LazyOf<F>::type;
  d->tw_fwd_lazy = d->tw_inv_lazy = nullptr;
  const size_t staged = d->n > 1 ? d->n - 1 : 1;  // one contiguous run per stage: 2^L - 1 entries
  hipError_t e4 = hipMalloc(&d->tw_fwd_lazy, staged * (sizeof(F) + sizeof(int32_t)));  // pre-sliced: 8 + 1 limbs per entry
  hipError_t e5 = hipMalloc(&d->tw_inv_lazy, staged * (sizeof(F) + sizeof(int32_t)));
  if (e4 == hipSuccess && e5 == hipSuccess && d->log_n >= 1) {
    hipLaunchKernelGGL((k_to_lazy_table<F, LZ>), dim3(grid_for(staged, 256)), dim3(256), 0, st, (const F*)d->tw_fwd, (F*)d->tw_fwd_lazy, (int)d->log_n);
    hipLaunchKernelGGL((k_to_lazy_table<F, LZ>), dim3(grid_for(staged, 256)), dim3(256), 0, st, (const F*)d->tw_inv, (F*)d->tw_inv_lazy, (int)d->log_n);
  }
  {
    const F nl = LZ::repack_for_storage(n_inv);  // host evaluation of the same template
    memcpy(d->n_inv_lazy, &nl, 32);
  }
  hipError_t e3 = hipStreamSynchronize(st);
  // Footprint: the staged, pre-sliced tables are (2^L - 1) x 36 bytes per direction (2^24: 1.2 GB for both, 2^26: 4.8 GB);
  // the natural tables they were made from (2^(L-1) x 32 bytes per direction) are only read by the 32-bit pass (CSH_NTT_LAZY=0, A/B
  // runs), so they are released here and rebuilt on demand (plain_tables) -- a third less per cached domain.
  if (e3 == hipSuccess && e4 == hipSuccess && e5 == hipSuccess && tune().ntt_lazy.load(std::memory_order_relaxed) != 0) {
    (void)hipFree(d->tw_fwd);
    (void)hipFree(d->tw_inv);
    d->tw_fwd = d->tw_inv = nullptr;
  }
  if (e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
    set_error("twiddle generation failed: %s", hipGetErrorString(e3 != hipSuccess ? e3 : (e4 != hipSuccess ? e4 : e5)));
    (void)hipFree(d->tw_fwd);
    (void)hipFree(d->tw_inv);
    if (d->tw_fwd_lazy) (void)hipFree(d->tw_fwd_lazy);
    if (d->tw_inv_lazy) (void)hipFree(d->tw_inv_lazy);
    delete d;
    return e3 != hipSuccess ? CSH_ERR_HIP : CSH_ERR_OOM;
  }
  *out = d;
  return CSH_OK;
}

template <class F>
static int coset_table_t(const Domain* d, const uint64_t* shift, uint64_t* out_dev, hipStream_t st) {
  F sh = f_from_words<F>(shift);
  hipLaunchKernelGGL((k_powers<F, true>), dim3(grid_for((d->n + POW_CHUNK - 1) / POW_CHUNK, 256)), dim3(256), 0, st, (F*)out_dev, sh, d->n,
                     (int)d->log_n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}

// exposed to other translation units (fused Groth16 h pipeline)
// ifft_in_to_out (its last pass multiplying entry i by scale_table[i], or by 1/n when the table is NULL) followed by fft_out_to_in
// on the same vector, the two passes over the contiguous tiles fused into ONE launch (k_ntt_pass_r4<.., PAIR>): reduction.rs:141-174 per
// vector. Falls back to the two transforms when the radix-4 pass does not apply to the plan's first pass (or tune "ntt_pair" = 0).
template <class F>
static int run_ntt_pair(const Domain* d, F* data, uint32_t ncomp, hipStream_t st, const F* scale_tbl) {
  const int L = (int)d->log_n;
  if (L == 0) return CSH_OK;
  using LZ = typename LazyOf<F>::type;
  const int ncomp_log = ncomp == 2 ? 1 : 0;
  Pass passes[8];
  const int np = plan_passes(L, ncomp_log, passes);
  const Pass& p = passes[0];
  const int tile_log = p.k + p.cb;
  const bool use_lazy = tune().ntt_lazy.load(std::memory_order_relaxed) != 0;
  const int nv = tune().ntt_variant.load(std::memory_order_relaxed);
  const bool fusable = use_lazy && tune().ntt_pair.load(std::memory_order_relaxed) != 0 && p.s0 == 0 && p.cb == 0 && p.k >= 2 && tile_log + ncomp_log >= 2 &&
                       (nv & ~0x100800) == 0 && L >= tune().ntt_pair_min_log.load(std::memory_order_relaxed);
  if (!fusable) {
    CSH_TRY(run_ntt<F>(d, data, ncomp, true, st, scale_tbl));
    return run_ntt<F>(d, data, ncomp, false, st);
  }
  if (np > 1) CSH_TRY(run_ntt<F>(d, data, ncomp, true, st, scale_tbl, true));   // the strided passes of the inverse transform
  const size_t tiles = d->n >> tile_log;
  const size_t lds_bytes = size_t(4 * LZ::NL) << (tile_log + ncomp_log);
  if (lds_bytes > 48 * 1024) CSH_TRY(raise_lds_limit((const void*)k_ntt_pass_r4<F, LZ, true, true>, (size_t)(4 * LZ::NL) << NTT_TILE_LOG));
  const size_t units = (size_t(1) << (tile_log + ncomp_log)) >> 2;
  const int nt = units >= 512 ? 512 : (units >= 64 ? (int)units : 64);
  const int noswz = ((nv & 0x800) ? 0x1000 : 0) | ((nv & 0x100000) ? 0x2000 : 0);
  const F scale = f_from_words<F>(d->n_inv_lazy);
  hipLaunchKernelGGL((k_ntt_pass_r4<F, LZ, true, true>), dim3((unsigned)tiles), dim3(nt), lds_bytes, st, data, reinterpret_cast<const F*>(d->tw_inv_lazy), L, p.s0, p.k,
                     p.cb, ncomp_log, scale, 1 | noswz, scale_tbl, reinterpret_cast<const F*>(d->tw_fwd_lazy));
  CSH_HIP(hipGetLastError());
  if (np > 1) CSH_TRY(run_ntt<F>(d, data, ncomp, false, st, nullptr, true));   // the strided passes of the forward transform
  return CSH_OK;
}
int ntt_run_pair_table(const Domain* d, uint64_t* data, uint32_t ncomp, const uint64_t* scale_table, hipStream_t st) {
  if (d->curve == CSH_BN254) return run_ntt_pair<Bn254Fr>(d, (Bn254Fr*)data, ncomp, st, (const Bn254Fr*)scale_table);
  if (d->curve == CSH_BLS12_377) return run_ntt_pair<Bls377Fr>(d, (Bls377Fr*)data, ncomp, st, (const Bls377Fr*)scale_table);
  return run_ntt_pair<Bls381Fr>(d, (Bls381Fr*)data, ncomp, st, (const Bls381Fr*)scale_table);
}

int ntt_run(const Domain* d, uint64_t* data, uint32_t ncomp, bool dif, hipStream_t st) {
  if (d->curve == CSH_BN254) return run_ntt<Bn254Fr>(d, (Bn254Fr*)data, ncomp, dif, st);
  if (d->curve == CSH_BLS12_377) return run_ntt<Bls377Fr>(d, (Bls377Fr*)data, ncomp, dif, st);
  return run_ntt<Bls381Fr>(d, (Bls381Fr*)data, ncomp, dif, st);
}
// ifft_in_to_out whose last pass multiplies entry i by table[i] instead of 1/n (table from ntt_coset_table_scaled); false when
// the lazy passes are switched off (the caller then runs the unfused sequence)
bool ntt_scale_table_supported(const Domain* d) { return tune().ntt_lazy.load(std::memory_order_relaxed) != 0 && d->log_n >= 1; }
int ntt_run_dif_table(const Domain* d, uint64_t* data, uint32_t ncomp, const uint64_t* scale_table, hipStream_t st) {
  if (d->curve == CSH_BN254) return run_ntt<Bn254Fr>(d, (Bn254Fr*)data, ncomp, true, st, (const Bn254Fr*)scale_table);
  if (d->curve == CSH_BLS12_377) return run_ntt<Bls377Fr>(d, (Bls377Fr*)data, ncomp, true, st, (const Bls377Fr*)scale_table);
  return run_ntt<Bls381Fr>(d, (Bls381Fr*)data, ncomp, true, st, (const Bls381Fr*)scale_table);
}
template <class F>
static int coset_table_scaled_t(const Domain* d, const uint64_t* shift, uint64_t* out_dev, hipStream_t st) {
  using LZ = typename LazyOf<F>::type;
  hipLaunchKernelGGL((k_powers_lazy_scaled<F, LZ>), dim3(grid_for((d->n + POW_CHUNK - 1) / POW_CHUNK, 256)), dim3(256), 0, st, (F*)out_dev, f_from_words<F>(shift),
                     f_from_words<F>(d->n_inv), d->n, (int)d->log_n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
// table[bitrev(i)] = shift^i / n in the storage form of the lazy passes
int ntt_coset_table_scaled(const Domain* d, const uint64_t* shift, uint64_t* out_dev, hipStream_t st) {
  if (d->curve == CSH_BN254) return coset_table_scaled_t<Bn254Fr>(d, shift, out_dev, st);
  if (d->curve == CSH_BLS12_377) return coset_table_scaled_t<Bls377Fr>(d, shift, out_dev, st);
  return coset_table_scaled_t<Bls381Fr>(d, shift, out_dev, st);
}
// The same table, kept with the domain (see Domain::cs_table): *table = the cached copy, valid for work queued on `st` after this call
// (the builder's stream records an event every other stream waits for), or `scratch` freshly filled when the shift is not the cached one,
// the domain is too large to keep a copy (> 2^25 points = 1 GiB) or the allocation fails.
int ntt_coset_table_scaled_cached(const Domain* dc, const uint64_t* shift, uint64_t* scratch, hipStream_t st, const uint64_t** table) {
  Domain* d = const_cast<Domain*>(dc);
  *table = scratch;
  int cur = -1;
  if (d->log_n <= 25 && tune().h_table_cache.load(std::memory_order_relaxed) && hipGetDevice(&cur) == hipSuccess && cur == d->device) {  // (the copy is allocated on the domain's GPU)
    std::lock_guard<std::mutex> g(d->cs_mu);
    if (!d->cs_table) {
      void* buf = nullptr;
      hipEvent_t ev = nullptr;
      if (hipMalloc(&buf, 32 * d->n) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        const int rc = ntt_coset_table_scaled(d, shift, static_cast<uint64_t*>(buf), st);
        if (rc == CSH_OK && hipEventRecord(ev, st) == hipSuccess) {
          memcpy(d->cs_shift, shift, 32);
          d->cs_table = buf;
          d->cs_ready = ev;
          *table = static_cast<const uint64_t*>(buf);
          return CSH_OK;
        }
      }
      (void)hipGetLastError();
      if (ev) (void)hipEventDestroy(ev);
      if (buf) (void)hipFree(buf);
    } else if (memcmp(d->cs_shift, shift, 32) == 0) {
      CSH_HIP(hipStreamWaitEvent(st, d->cs_ready, 0));
      *table = static_cast<const uint64_t*>(d->cs_table);
      return CSH_OK;
    }
  }
  return ntt_coset_table_scaled(d, shift, scratch, st);
}
int ntt_coset_table(const Domain* d, const uint64_t* shift, uint64_t* out_dev, hipStream_t st) {
  if (d->curve == CSH_BN254) return coset_table_t<Bn254Fr>(d, shift, out_dev, st);
  if (d->curve == CSH_BLS12_377) return coset_table_t<Bls377Fr>(d, shift, out_dev, st);
  return coset_table_t<Bls381Fr>(d, shift, out_dev, st);
}
int ntt_bit_reverse(csh_curve_t c, uint64_t* data, uint32_t log_n, uint32_t ncomp, hipStream_t st) {
  if (c == CSH_BN254) return run_bit_reverse<Bn254Fr>((Bn254Fr*)data, log_n, ncomp, st);
  if (c == CSH_BLS12_377) return run_bit_reverse<Bls377Fr>((Bls377Fr*)data, log_n, ncomp, st);
  return run_bit_reverse<Bls381Fr>((Bls381Fr*)data, log_n, ncomp, st);
}

size_t domain_size_of(const Domain* d) { return d->n; }
csh_curve_t domain_curve_of(const Domain* d) { return d->curve; }

}  // namespace csh

using namespace csh;

static int check_dom(csh_domain_t dom, uint32_t ncomp) {
  CSH_REQUIRE(dom, "domain is NULL");
  CSH_REQUIRE(ncomp == 1 || ncomp == 2, "ncomp must be 1 or 2");
  int cur = -1;
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  if (hipGetDevice(&cur) == hipSuccess && cur != d->device) {
    set_error("the domain's twiddle tables live on device %d but the calling thread is bound to device %d (csh_init)", d->device, cur);
    return CSH_ERR_INVALID;
  }
  return CSH_OK;
}

extern "C" {

int csh_domain_create(csh_curve_t field_of, uint32_t log_n, const uint64_t group_gen[4], csh_domain_t* out) {
  CSH_REQUIRE(out, "out is NULL");
  CSH_REQUIRE(log_n <= 31, "log_n too large");
  CSH_TRY(ensure_device());
  Domain* d = nullptr;
  int rc;
  if (field_of == CSH_BN254)
    rc = create_domain_t<Bn254Fr>(field_of, log_n, group_gen, 5, &d);
  else if (field_of == CSH_BLS12_381)
    rc = create_domain_t<Bls381Fr>(field_of, log_n, group_gen, 7, &d);
  else if (field_of == CSH_BLS12_377)
    rc = create_domain_t<Bls377Fr>(field_of, log_n, group_gen, 22, &d);  // ark_bls12_377::Fr::GENERATOR
  else {
    set_error("unknown curve %d", (int)field_of);
    return CSH_ERR_INVALID;
  }
  if (rc != CSH_OK) return rc;
  *out = reinterpret_cast<csh_domain_t>(d);
  return CSH_OK;
}
int csh_domain_size(csh_domain_t dom, size_t* n) {
  CSH_REQUIRE(dom && n, "NULL argument");
  *n = reinterpret_cast<Domain*>(dom)->n;
  return CSH_OK;
}
int csh_domain_free(csh_domain_t dom) {
  if (!dom) return CSH_OK;
  Domain* d = reinterpret_cast<Domain*>(dom);
  if (d->tw_fwd) (void)hipFree(d->tw_fwd);
  if (d->tw_inv) (void)hipFree(d->tw_inv);
  if (d->tw_fwd_lazy) (void)hipFree(d->tw_fwd_lazy);
  if (d->tw_inv_lazy) (void)hipFree(d->tw_inv_lazy);
  if (d->cs_table) (void)hipFree(d->cs_table);
  if (d->cs_ready) (void)hipEventDestroy(d->cs_ready);
  delete d;
  return CSH_OK;
}

int csh_ifft_in_to_out_dev(csh_domain_t dom, uint64_t* data, uint32_t ncomp, void* stream) {
  CSH_TRY(check_dom(dom, ncomp));
  CSH_TRY(ensure_device());
  return ntt_run(reinterpret_cast<Domain*>(dom), data, ncomp, true, resolve_stream(stream));
}
int csh_fft_out_to_in_dev(csh_domain_t dom, uint64_t* data, uint32_t ncomp, void* stream) {
  CSH_TRY(check_dom(dom, ncomp));
  CSH_TRY(ensure_device());
  return ntt_run(reinterpret_cast<Domain*>(dom), data, ncomp, false, resolve_stream(stream));
}
int csh_fft_dev(csh_domain_t dom, uint64_t* data, uint32_t ncomp, void* stream) {
  CSH_TRY(check_dom(dom, ncomp));
  CSH_TRY(ensure_device());
  Domain* d = reinterpret_cast<Domain*>(dom);
  hipStream_t st = resolve_stream(stream);
  CSH_TRY(ntt_bit_reverse(d->curve, data, d->log_n, ncomp, st));
  return ntt_run(d, data, ncomp, false, st);
}
int csh_ifft_dev(csh_domain_t dom, uint64_t* data, uint32_t ncomp, void* stream) {
  CSH_TRY(check_dom(dom, ncomp));
  CSH_TRY(ensure_device());
  Domain* d = reinterpret_cast<Domain*>(dom);
  hipStream_t st = resolve_stream(stream);
  CSH_TRY(ntt_run(d, data, ncomp, true, st));
  return ntt_bit_reverse(d->curve, data, d->log_n, ncomp, st);
}
int csh_bit_reverse_dev(csh_curve_t field_of, uint64_t* data, uint32_t log_n, uint32_t ncomp, void* stream) {
  CSH_REQUIRE(ncomp == 1 || ncomp == 2, "ncomp must be 1 or 2");
  CSH_REQUIRE(field_of == CSH_BN254 || field_of == CSH_BLS12_381 || field_of == CSH_BLS12_377, "unknown curve");
  CSH_REQUIRE(log_n <= 31, "log_n too large");
  CSH_TRY(ensure_device());
  return ntt_bit_reverse(field_of, data, log_n, ncomp, resolve_stream(stream));
}
int csh_coset_table_dev(csh_domain_t dom, const uint64_t shift[4], uint64_t* out, void* stream) {
  CSH_REQUIRE(dom && shift && out, "NULL argument");
  CSH_TRY(ensure_device());
  return ntt_coset_table(reinterpret_cast<Domain*>(dom), shift, out, resolve_stream(stream));
}

// ---- host-pointer wrappers ---------------------------------------------------------------------------
static int host_transform(csh_domain_t dom, uint64_t* data, uint32_t ncomp, int which) {
  CSH_TRY(check_dom(dom, ncomp));
  Domain* d = reinterpret_cast<Domain*>(dom);
  HostStage h;
  const size_t bytes = d->n * ncomp * 32;
  CSH_TRY(h.begin(Arena::padded(bytes)));
  uint64_t* dd;
  CSH_TRY(h.up(dd, data, bytes));
  int rc = CSH_OK;
  switch (which) {
    case 0: rc = csh_ifft_in_to_out_dev(dom, dd, ncomp, h.st); break;
    case 1: rc = csh_fft_out_to_in_dev(dom, dd, ncomp, h.st); break;
    case 2: rc = csh_fft_dev(dom, dd, ncomp, h.st); break;
    default: rc = csh_ifft_dev(dom, dd, ncomp, h.st); break;
  }
  if (rc != CSH_OK) return rc;
  return h.down(data, dd, bytes);
}
int csh_ifft_in_to_out(csh_domain_t dom, uint64_t* data, uint32_t ncomp) { return host_transform(dom, data, ncomp, 0); }
int csh_fft_out_to_in(csh_domain_t dom, uint64_t* data, uint32_t ncomp) { return host_transform(dom, data, ncomp, 1); }
int csh_fft(csh_domain_t dom, uint64_t* data, uint32_t ncomp) { return host_transform(dom, data, ncomp, 2); }
int csh_ifft(csh_domain_t dom, uint64_t* data, uint32_t ncomp) { return host_transform(dom, data, ncomp, 3); }

int csh_bit_reverse(csh_curve_t field_of, uint64_t* data, uint32_t log_n, uint32_t ncomp) {
  CSH_REQUIRE(log_n <= 31, "log_n too large");
  HostStage h;
  const size_t bytes = (size_t(1) << log_n) * ncomp * 32;
  CSH_TRY(h.begin(Arena::padded(bytes)));
  uint64_t* dd;
  CSH_TRY(h.up(dd, data, bytes));
  CSH_TRY(csh_bit_reverse_dev(field_of, dd, log_n, ncomp, h.st));
  return h.down(data, dd, bytes);
}
int csh_coset_table(csh_domain_t dom, const uint64_t shift[4], uint64_t* out) {
  CSH_REQUIRE(dom && shift && out, "NULL argument");
  Domain* d = reinterpret_cast<Domain*>(dom);
  HostStage h;
  const size_t bytes = d->n * 32;
  CSH_TRY(h.begin(Arena::padded(bytes)));
  uint64_t* dd;
  CSH_TRY(h.up(dd, nullptr, bytes));
  CSH_TRY(csh_coset_table_dev(dom, shift, dd, h.st));
  return h.down(out, dd, bytes);
}

}  // extern "C"
