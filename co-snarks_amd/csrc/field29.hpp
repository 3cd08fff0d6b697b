// Lazy-reduction base-field representation for gfx950: NL limbs of B (<32) bits in u32 containers.
//
// Why: v_mad_u64_u32 accumulates a 32x32 product into a 64-bit register pair but has no carry-in, so a
// saturated 32-bit-limb CIOS spends ~3 extra VALU ops per product on carry handling. With 29-bit limbs
// (BN254 Fq: 9 limbs, R' = 2^261) a whole product column (<= 18 products < 2^58) fits a 64-bit accumulator:
// every partial product is exactly one v_mad_u64_u32 and carries are resolved once per column.
// Additions are limb-wise with no carry chain. Montgomery domain is R' = 2^(NL*B); conversion to/from the
// arkworks 32-bit Montgomery encoding costs one multiplication by a constant each way.
//
// Bounds contract (checked by tests/host): mul/sqr operands may have limbs < 2^30 (i.e. a sum of two
// normalised values) and values < 8p; outputs are normalised (limbs < 2^B) with value < 2p.
#pragma once
#include <math.h>
#include "field.hpp"

namespace csh {

template <class LP, class F32>
struct FpLazy {
  static constexpr int NL = LP::NL;
  static constexpr int B = LP::B;
  uint32_t l[NL];

  CSH_HD static FpLazy zero() {
    FpLazy r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = 0;
    return r;
  }
  CSH_HD static FpLazy one() {
    FpLazy r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = LP::ONE[i];
    return r;
  }

  // limb-wise, no carry propagation: result limbs < sum of operand limb bounds
  CSH_HD static FpLazy add(const FpLazy& a, const FpLazy& b) {
    FpLazy r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
  }

  // one parallel carry step: limbs < 2^B + 2^(32-B)
  CSH_HD FpLazy normalized() const {
    FpLazy r;
    r.l[0] = l[0] & LP::MASK;
#pragma unroll
    for (int i = 1; i < NL - 1; ++i) r.l[i] = (l[i] & LP::MASK) + (l[i - 1] >> B);
    r.l[NL - 1] = l[NL - 1] + (l[NL - 2] >> B);
    return r;
  }

  CSH_HD static FpLazy mul(const FpLazy& a, const FpLazy& b) {
    uint64_t t[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) t[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int j = 0; j < NL; ++j) t[i + j] = (uint64_t)a.l[i] * b.l[j] + t[i + j];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const uint32_t m = ((uint32_t)t[i] * LP::INV) & LP::MASK;
#pragma unroll
      for (int j = 0; j < NL; ++j) t[i + j] = (uint64_t)m * LP::MOD[j] + t[i + j];
      t[i + 1] += t[i] >> B;  // low B bits of t[i] are now zero
    }
    FpLazy r;
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) {
      r.l[k] = (uint32_t)t[NL + k] & LP::MASK;
      t[NL + k + 1] += t[NL + k] >> B;
    }
    r.l[NL - 1] = (uint32_t)t[2 * NL - 1];
    return r;
  }
  CSH_HD static FpLazy sqr(const FpLazy& a) { return mul(a, a); }

  // full carry propagation + conditional subtractions -> canonical value in [0, p), normalised limbs
  CSH_HD FpLazy canonical() const {
    FpLazy r = *this;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      uint32_t v = r.l[i] + c;
      r.l[i] = v & LP::MASK;
      c = v >> B;
    }
    r.l[NL - 1] += c;
    for (int rep = 0; rep < 8; ++rep) {
      // r >= p ?  (top limb may exceed B bits: compare as integers limb by limb from the top)
      uint32_t d[NL];
      int32_t borrow = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        int64_t v = (int64_t)r.l[i] - (int64_t)LP::MOD[i] + borrow;
        if (i < NL - 1) {
          d[i] = (uint32_t)v & LP::MASK;
          borrow = (int32_t)(v >> B);
        } else {
          d[i] = (uint32_t)v;
          borrow = v < 0 ? -1 : 0;
        }
      }
      if (borrow < 0) break;
#pragma unroll
      for (int i = 0; i < NL; ++i) r.l[i] = d[i];
    }
    return r;
  }

  // arkworks 32-bit Montgomery element (x * 2^(32*N)) -> lazy Montgomery element (x * R')
  CSH_HD static FpLazy from_fp(const F32& f) {
    FpLazy s = reslice_in(f);
    FpLazy c;
#pragma unroll
    for (int i = 0; i < NL; ++i) c.l[i] = LP::TO_LAZY[i];
    return mul(s, c);
  }
  CSH_HD F32 to_fp() const {
    FpLazy c;
#pragma unroll
    for (int i = 0; i < NL; ++i) c.l[i] = LP::FROM_LAZY[i];
    FpLazy v = mul(this->normalized(), c).canonical();
    return reslice_out(v);
  }

  CSH_HD static FpLazy reslice_in(const F32& f) {
    FpLazy r;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int bit = i * B;
      const int w = bit >> 5, off = bit & 31;
      uint64_t two = w < F32::N ? f.l[w] : 0;
      if (w + 1 < F32::N) two |= (uint64_t)f.l[w + 1] << 32;
      r.l[i] = (uint32_t)(two >> off) & LP::MASK;
    }
    return r;
  }
  CSH_HD static F32 reslice_out(const FpLazy& v) {
    F32 f = F32::zero();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int bit = i * B;
      const int w = bit >> 5, off = bit & 31;
      const uint64_t sh = (uint64_t)v.l[i] << off;
      if (w < F32::N) f.l[w] |= (uint32_t)sh;
      if (w + 1 < F32::N) f.l[w + 1] |= (uint32_t)(sh >> 32);
    }
    return f;
  }
};

using Fq29 = FpLazy<Bn254Fq29Params, Bn254Fq>;

// ================================================================================================
// Signed lazy representation (the one the MSM bucket kernels use).
//
// Limbs are int32 in radix 2^B; a value is sum l[i] 2^(B i) and may be NEGATIVE or exceed p: every element
// is only defined mod p. Subtraction / negation are limb-wise with no modular correction, products use
// v_mad_i64_i32 into 64-bit column accumulators, and Montgomery reduction (R' = 2^(NL*B)) returns a value in
// (-p/32, p + p/32) whose limbs 0..NL-2 are in [0, 2^B) and whose top limb carries the sign.
// Contract: |limb| of mul operands <= 2^B + 2 for both, or <= 2^(B+1) for one of them; operand values within
// (-8p, 8p). `normalized()` (one parallel signed carry step) restores |limb| <= 2^B + 1 after 2-3 term sums.
// Host-only contract checks (enabled by the self-test translation unit): every product routine verifies the
// limb magnitudes its 63-bit column bound was derived for, so a caller that forgets a normalized() fails loudly
// in `pytest -m "not gpu"` instead of producing a one-in-10^5 wrong bucket on the device.
#if defined(CSH_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <stdio.h>
#include <stdlib.h>
#define CSH_LIMB_BOUND(x, bound, what)                                                        \
  do {                                                                                        \
    for (int i__ = 0; i__ < NL; ++i__) {                                                      \
      const int64_t v__ = (x).l[i__];                                                         \
      if (v__ > (int64_t)(bound) || v__ < -(int64_t)(bound)) {                                \
        fprintf(stderr, "FpS bound violation in %s: limb %d = %lld exceeds %lld\n", what, i__, (long long)v__, (long long)(bound)); \
        abort();                                                                              \
      }                                                                                       \
    }                                                                                         \
  } while (0)
#else
#define CSH_LIMB_BOUND(x, bound, what) do { } while (0)
#endif

// the column-term lambdas of reduce_scan must be inlined whatever their size (the column index has to be a constant)
#define CSH_LAMBDA_INLINE __attribute__((always_inline))
#ifndef CSH_REDUCE_SCAN
#define CSH_REDUCE_SCAN 1
#endif
// CSH_PIN_MADS: per translation unit. 0 (default): the compiler is free to reassociate a column's terms -- it moves the incoming
// carry to the end of the chain and pays a separate 64-bit addition per column, i.e. the instruction count of the row-wise form.
// 3: every partial sum gets a second (empty, input-only asm) use, which keeps the chain in the written order: the carry is the
// addend of the column's first multiply-add. Worth -4 % on the accumulate kernels and -5.5 % on the NTT passes (profiles/archive/r03_a_*);
// set by the translation units of those kernels only (msm_accum_*.hip, ntt.hip): the volatile asm statements are ordered among
// themselves, and in the four-lane tail kernels (long independent chains the scheduler wants to interleave) that ordering blew the
// register allocation up to 512 VGPRs + spills (tails +13 % on G1, x2 on G2 / BLS12-381).
#ifndef CSH_DUAL_CHAIN
#define CSH_DUAL_CHAIN 1  // FpS::mul2 runs its two multiplications in lockstep (reduce_scan2); 0 = one after the other (A/B builds)
#endif
#ifndef CSH_PIN_MADS
#define CSH_PIN_MADS 0
#endif
// acc = x * y + acc as ONE v_mad_i64_i32 in the order written: the empty asm makes every partial sum opaque to the compiler's
// reassociation pass, which otherwise sorts a column's terms by rank, moves the incoming carry (the latest value) to the end of
// the chain and pays a separate 64-bit addition for it.
CSH_HD inline int64_t mad_pinned(int32_t x, int32_t y, int64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__) && CSH_PIN_MADS == 2
  uint64_t sd;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sd) : "v"(x), "v"(y));
#else
  acc = (int64_t)x * (int64_t)y + acc;
#if defined(__HIP_DEVICE_COMPILE__) && CSH_PIN_MADS == 1
  asm("" : "+v"(acc));
#elif defined(__HIP_DEVICE_COMPILE__) && CSH_PIN_MADS == 3
  asm volatile("" ::"v"(acc));  // a second use of the partial sum: the reassociation pass only linearises single-use chains
#elif defined(__HIP_DEVICE_COMPILE__) && CSH_PIN_MADS == 4
  // a second use that exists only in the optimiser: the comparison feeding llvm.assume keeps the partial sum out of the
  // reassociation pass's single-use chains and is dropped before instruction selection, so the machine scheduler and the register
  // allocator see plain multiply-adds (the volatile asm of variant 3 orders thousands of statements and blew the four-lane
  // tail kernels up to 512 VGPRs + spills)
  __builtin_assume(acc != INT64_MIN);
#endif
#endif
  return acc;
}
// the same with a wave-uniform second factor (a modulus limb) in a scalar register
CSH_HD inline int64_t mad_pinned_s(int32_t x, int32_t y, int64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__) && CSH_PIN_MADS == 2
  uint64_t sd;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sd) : "v"(x), "s"(y));
  return acc;
#else
  return mad_pinned(x, y, acc);
#endif
}

template <class LP, class F32>
struct FpS {
  static constexpr int NL = LP::NL;
  static constexpr int B = LP::B;
  int32_t l[NL];

  CSH_HD static FpS zero() {
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = 0;
    return r;
  }
  CSH_HD static FpS one() {
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = (int32_t)LP::ONE[i];
    return r;
  }
  CSH_HD static FpS add(const FpS& a, const FpS& b) {
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
  }
  CSH_HD static FpS sub(const FpS& a, const FpS& b) {
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] - b.l[i];
    return r;
  }
  CSH_HD static FpS neg(const FpS& a) {
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = -a.l[i];
    return r;
  }
  // -a for an unpack() output (limbs 0..NL-2 in [0, 2^B), small non-negative top limb) by limb-wise complement:
  // sum (MASK - l_i) 2^(B i) + 1 - (l_top + 1) 2^(B (NL-1)) = -a. One instruction per limb, no carry step: limbs 0..NL-2 stay in
  // [0, 2^B], the top limb turns negative (it carries the sign, as after a reduction). Replaces neg().normalized() where a
  // stored base point is negated (4 instructions per limb).
  CSH_HD FpS neg_unpacked() const {
    FpS r;
    r.l[0] = (int32_t)(LP::MASK + 1u) - l[0];
#pragma unroll
    for (int i = 1; i < NL - 1; ++i) r.l[i] = (int32_t)LP::MASK - l[i];
    r.l[NL - 1] = -1 - l[NL - 1];
    return r;
  }
  // neg01 ? neg_unpacked() : *this, branch-free (x ^ MASK = MASK - x on B-bit limbs, x ^ -1 = -1 - x on the top limb)
  CSH_HD FpS cneg_unpacked(uint32_t neg01) const {
    const uint32_t all = 0u - neg01, low = all & LP::MASK;
    FpS r;
    r.l[0] = (int32_t)(((uint32_t)l[0] ^ low) + neg01);
#pragma unroll
    for (int i = 1; i < NL - 1; ++i) r.l[i] = (int32_t)((uint32_t)l[i] ^ low);
    r.l[NL - 1] = (int32_t)((uint32_t)l[NL - 1] ^ all);
    return r;
  }
  CSH_HD FpS normalized() const {
    FpS r;
    r.l[0] = (int32_t)((uint32_t)l[0] & LP::MASK);
#pragma unroll
    for (int i = 1; i < NL - 1; ++i) r.l[i] = (int32_t)((uint32_t)l[i] & LP::MASK) + (l[i - 1] >> B);
    r.l[NL - 1] = l[NL - 1] + (l[NL - 2] >> B);
    return r;
  }

  // ---- double-width (unreduced) products: 2*NL signed 64-bit columns ------------------------------------
  struct Wide {
    int64_t t[2 * NL];
  };
  static constexpr int64_t LIM1 = (int64_t(1) << B) + 8;        // normalised operand
  static constexpr int64_t LIM2 = (int64_t(1) << (B + 1)) + 16;  // one two-term sum
  CSH_HD static Wide mul_wide(const FpS& a, const FpS& b) {
    CSH_LIMB_BOUND(a, LIM2, "mul_wide(a)");
    CSH_LIMB_BOUND(b, LIM1, "mul_wide(b)");
    Wide w;
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) w.t[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int j = 0; j < NL; ++j) w.t[i + j] = (int64_t)a.l[i] * (int64_t)b.l[j] + w.t[i + j];
    }
    return w;
  }
  // a^2 with NL(NL+1)/2 products: squares + doubled cross terms (|2 a_j| <= 2^(B+1))
  CSH_HD static Wide sqr_wide(const FpS& a) {
    CSH_LIMB_BOUND(a, LIM1, "sqr_wide");
    Wide w;
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) w.t[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      w.t[2 * i] = (int64_t)a.l[i] * (int64_t)a.l[i] + w.t[2 * i];
      const int32_t a2 = a.l[i] + a.l[i];
#pragma unroll
      for (int j = i + 1; j < NL; ++j) w.t[i + j] = (int64_t)a2 * (int64_t)a.l[j] + w.t[i + j];
    }
    return w;
  }
  // w = a*b - c*d accumulated column-wise before ONE reduction (2 * NL^2 products, saves a reduction)
  CSH_HD static Wide mul_sub_wide(const FpS& a, const FpS& b, const FpS& c, const FpS& d) {
    CSH_LIMB_BOUND(a, LIM1, "mul_sub_wide(a)");
    CSH_LIMB_BOUND(c, LIM1, "mul_sub_wide(c)");
    CSH_LIMB_BOUND(d, LIM1, "mul_sub_wide(d)");
    Wide w = mul_wide(a, b);
    FpS nc = neg(c);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int j = 0; j < NL; ++j) w.t[i + j] = (int64_t)nc.l[i] * (int64_t)d.l[j] + w.t[i + j];
    }
    return w;
  }
  // w = a*b + c*d
  CSH_HD static Wide mul_add_wide(const FpS& a, const FpS& b, const FpS& c, const FpS& d) {
    CSH_LIMB_BOUND(a, LIM1, "mul_add_wide(a)");
    CSH_LIMB_BOUND(c, LIM1, "mul_add_wide(c)");
    CSH_LIMB_BOUND(d, LIM1, "mul_add_wide(d)");
    Wide w = mul_wide(a, b);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int j = 0; j < NL; ++j) w.t[i + j] = (int64_t)c.l[i] * (int64_t)d.l[j] + w.t[i + j];
    }
    return w;
  }
  // w = a^2 - b^2
  CSH_HD static Wide sqr_sub_wide(const FpS& a, const FpS& b) {
    CSH_LIMB_BOUND(b, LIM1, "sqr_sub_wide(b)");
    Wide w = sqr_wide(a);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int32_t nb = -b.l[i];
      w.t[2 * i] = (int64_t)nb * (int64_t)b.l[i] + w.t[2 * i];
      const int32_t nb2 = nb + nb;
#pragma unroll
      for (int j = i + 1; j < NL; ++j) w.t[i + j] = (int64_t)nb2 * (int64_t)b.l[j] + w.t[i + j];
    }
    return w;
  }
  // accumulate a*b (sign = +1) or -a*b (sign = -1) into w
  CSH_HD static void mac_wide(Wide& w, const FpS& a, const FpS& b, bool negate) {
    CSH_LIMB_BOUND(a, LIM1, "mac_wide(a)");
    CSH_LIMB_BOUND(b, LIM1, "mac_wide(b)");
    FpS aa = negate ? neg(a) : a;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int j = 0; j < NL; ++j) w.t[i + j] = (int64_t)aa.l[i] * (int64_t)b.l[j] + w.t[i + j];
    }
  }
  // One signed carry sweep over a double-width value: columns 0..2NL-2 come back into [0, 2^B), the top column takes the rest.
  // Lets a second pair of NL-term product sums share the accumulator where four would overflow a 63-bit column (9 x 29 bits):
  // ~4 simple instructions per column instead of a whole extra Montgomery reduction (NL^2 mads + carries).
  CSH_HD static void compress_wide(Wide& w) {
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
      w.t[k + 1] += w.t[k] >> B;
      w.t[k] = (int64_t)((uint64_t)w.t[k] & (uint64_t)LP::MASK);
    }
  }
  CSH_HD static void add_wide(Wide& a, const Wide& b) {
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) a.t[k] += b.t[k];
  }
  // can four NL-term product sums (+ the reduction's NL terms) share one 63-bit column?
  static constexpr bool FOUR_PRODUCTS_FIT = (5.0 * NL) * (double)(1ull << (2 * B - 40)) < (double)(1ull << 23);

  // Montgomery reduction of a double-width value: (w + m p) / R', result in (-p/16, p + p/16) for |w| < 2^6 p R'/64
  CSH_HD static FpS reduce(Wide w) {
#if CSH_REDUCE_SCAN
    return reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return acc + w.t[k]; });
#endif
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int32_t m = (int32_t)(((uint32_t)w.t[i] * LP::INV) & LP::MASK);
#pragma unroll
      for (int j = 0; j < NL; ++j) w.t[i + j] = (int64_t)m * (int64_t)(int32_t)LP::MOD[j] + w.t[i + j];
      w.t[i + 1] += w.t[i] >> B;  // exact: the low B bits of t[i] are zero now
    }
    FpS r;
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) {
      r.l[k] = (int32_t)((uint32_t)w.t[NL + k] & LP::MASK);
      w.t[NL + k + 1] += w.t[NL + k] >> B;
    }
    r.l[NL - 1] = (int32_t)w.t[2 * NL - 1];
    return r;
  }
  // reduce(w) - s for a subtrahend with |limb| < 2^31 (a sum of a few reduced values): s enters the upper columns before the
  // output carry chain, one multiply-add per limb, so the difference comes out with limbs 0..NL-2 in [0, 2^B) -- instead of a
  // limb-wise subtraction per term plus a carry step afterwards (x3 = r^2 - ppp - 2q in every point addition).
  CSH_HD static FpS reduce_sub(Wide w, const FpS& s) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t m1;  // -1 the compiler cannot see through: keeps "s * -1 + t" ONE v_mad_i64_i32 (it would otherwise be rewritten
    asm("s_mov_b32 %0, -1" : "=s"(m1));  // as sign extension + a two-instruction 64-bit subtraction)
#else
    const int32_t m1 = -1;
#endif
#pragma unroll
    for (int k = 0; k < NL; ++k) w.t[NL + k] = (int64_t)s.l[k] * (int64_t)m1 + w.t[NL + k];
    return reduce(w);
  }
  // ---- product-scanning Montgomery multiplication -----------------------------------------------------------------------
  // The same sums as mul_wide() + reduce(), taken column by column: column k starts from the carry of column k - 1 (the
  // addend of its first multiply-add), collects its operand products and the m_i p_(k-i) terms, and hands acc >> B on. A carry
  // is then ONE shift instead of shift + 64-bit addition (16 instructions less per reduction), and only one 64-bit accumulator
  // is live instead of 2 NL columns. `col(k, acc)` adds the operand products of column k. Bit-identical to the row-wise form.
  // (column K as a template parameter: a #pragma unroll over the 2 NL columns exceeds the compiler's pragma-unroll size
  // threshold at NL = 14 and silently stays a loop, with the column index -- and every limb index derived from it -- dynamic)
  template <int K, class ColFn>
  CSH_HD static void scan_column(ColFn& col, int32_t* m, FpS& r, int64_t& acc) {
    acc = col(K, acc);
    if constexpr (K < NL) {
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i < K) acc = mad_pinned_s(m[i], (int32_t)LP::MOD[K - i], acc);
      m[K] = (int32_t)(((uint32_t)acc * LP::INV) & LP::MASK);
      acc = mad_pinned_s(m[K], (int32_t)LP::MOD[0], acc);  // the low B bits are zero now
      acc >>= B;
    } else if constexpr (K < 2 * NL - 1) {
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i > K - NL) acc = mad_pinned_s(m[i], (int32_t)LP::MOD[K - i], acc);
      r.l[K - NL] = (int32_t)((uint32_t)acc & LP::MASK);
      acc >>= B;
    } else {
      r.l[NL - 1] = (int32_t)acc;
    }
    if constexpr (K + 1 < 2 * NL) scan_column<K + 1>(col, m, r, acc);
  }
  template <class ColFn>
  CSH_HD static FpS reduce_scan(ColFn&& col) {
    int32_t m[NL];
    FpS r;
    int64_t acc = 0;
    scan_column<0>(col, m, r, acc);
    return r;
  }
  // ---- two multiplications in lockstep ---------------------------------------------------------------------------------------
  // A lane's multiply-adds of ONE product-scanning multiplication form a single dependent chain; in a register-only probe a dependent
  // v_mad_i64_i32 chain issues at 21.9 T mad/s (two waves per SIMD) against 26.3 for two chains per lane and 33.4 at the pipe's peak
  // (tools/gpu_probe_chain.py, profiles/archive/r03_o_probe_chain.log). reduce_scan2 runs two INDEPENDENT multiplications column by column with
  // their terms alternating A, B, A, B in program order (which the pin of mad_pinned preserves). Measured (profiles/archive/r03_p_*, A/B on one
  // box): the radix-4 NTT pass, whose folds hold two independent products, gains ~1.5 %; the bucket accumulation gains nothing (its
  // mixed addition paired as (u2, s2) (ppp, q) (x3, zzz3) (y3, zz3): BN254 G1 +-0, BN254 G2 +2 %, BLS12-381 G2 +10 % from spills) --
  // the other VALU work between a real multiplication's multiply-adds already fills the chain's latency -- so only the NTT uses it.
  // `ta(k, i, acc)` / `tb(k, i, acc)` add term slot i (0 <= i < NTA / NTB) of column k of their multiplication, if that slot exists.
  template <int K, int NTA, int NTB, class TA, class TB>
  CSH_HD static void scan_column2(TA& ta, TB& tb, int32_t* ma, int32_t* mb, FpS& ra, FpS& rb, int64_t& acca, int64_t& accb) {
#pragma unroll
    for (int i = 0; i < (NTA > NTB ? NTA : NTB); ++i) {
      if (i < NTA) acca = ta(K, i, acca);
      if (i < NTB) accb = tb(K, i, accb);
    }
    if constexpr (K < NL) {
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i < K) {
          acca = mad_pinned_s(ma[i], (int32_t)LP::MOD[K - i], acca);
          accb = mad_pinned_s(mb[i], (int32_t)LP::MOD[K - i], accb);
        }
      ma[K] = (int32_t)(((uint32_t)acca * LP::INV) & LP::MASK);
      mb[K] = (int32_t)(((uint32_t)accb * LP::INV) & LP::MASK);
      acca = mad_pinned_s(ma[K], (int32_t)LP::MOD[0], acca);
      accb = mad_pinned_s(mb[K], (int32_t)LP::MOD[0], accb);
      acca >>= B;
      accb >>= B;
    } else if constexpr (K < 2 * NL - 1) {
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i > K - NL) {
          acca = mad_pinned_s(ma[i], (int32_t)LP::MOD[K - i], acca);
          accb = mad_pinned_s(mb[i], (int32_t)LP::MOD[K - i], accb);
        }
      ra.l[K - NL] = (int32_t)((uint32_t)acca & LP::MASK);
      rb.l[K - NL] = (int32_t)((uint32_t)accb & LP::MASK);
      acca >>= B;
      accb >>= B;
    } else {
      ra.l[NL - 1] = (int32_t)acca;
      rb.l[NL - 1] = (int32_t)accb;
    }
    if constexpr (K + 1 < 2 * NL) scan_column2<K + 1, NTA, NTB>(ta, tb, ma, mb, ra, rb, acca, accb);
  }
  template <int NTA, int NTB, class TA, class TB>
  CSH_HD static void reduce_scan2(TA&& ta, TB&& tb, FpS& ra, FpS& rb) {
#if CSH_DUAL_CHAIN
    int32_t ma[NL], mb[NL];
    int64_t acca = 0, accb = 0;
    scan_column2<0, NTA, NTB>(ta, tb, ma, mb, ra, rb, acca, accb);
#else  // A/B builds (tools/experiments): the same two multiplications one after the other
    ra = reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE {
#pragma unroll
      for (int i = 0; i < NTA; ++i) acc = ta(k, i, acc);
      return acc;
    });
    rb = reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE {
#pragma unroll
      for (int i = 0; i < NTB; ++i) acc = tb(k, i, acc);
      return acc;
    });
#endif
  }
  // term slot i of column k: product a b (slots 0..NL-1)
  CSH_HD static int64_t term_mul(const FpS& a, const FpS& b, int k, int i, int64_t acc) {
    const int j = k - i;
    if (i >= 0 && i < NL && j >= 0 && j < NL) acc = mad_pinned(a.l[i], b.l[j], acc);
    return acc;
  }
  // (r1, r2) = (a b, c d)
  CSH_HD static void mul2(const FpS& a, const FpS& b, const FpS& c, const FpS& d, FpS& r1, FpS& r2) {
    CSH_LIMB_BOUND(a, LIM2, "mul2(a)");
    CSH_LIMB_BOUND(b, LIM1, "mul2(b)");
    CSH_LIMB_BOUND(c, LIM2, "mul2(c)");
    CSH_LIMB_BOUND(d, LIM1, "mul2(d)");
    reduce_scan2<NL, NL>([&](int k, int i, int64_t acc) CSH_LAMBDA_INLINE { return term_mul(a, b, k, i, acc); },
                         [&](int k, int i, int64_t acc) CSH_LAMBDA_INLINE { return term_mul(c, d, k, i, acc); }, r1, r2);
  }

  // k * a limb-wise for a small constant k (the Fp2 non-residue 5 of BLS12-377): a normalised, |k| LIM1 < 2^31
  CSH_HD static FpS scaled(const FpS& a, int32_t k) {
    CSH_LIMB_BOUND(a, LIM1, "scaled");
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = k * a.l[i];
    return r;
  }
  static constexpr int64_t LIM_SCALED = 5 * LIM1;  // scaled() output
  // column products with a scaled() first operand (its bound is checked as such; the caller accounts for the column sum)
  CSH_HD static int64_t col_mul_scaled(const FpS& a, const FpS& b, int k, int64_t acc) {
    CSH_LIMB_BOUND(a, LIM_SCALED, "col_mul_scaled(a)");
    CSH_LIMB_BOUND(b, LIM1, "col_mul_scaled(b)");
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (k - i >= 0 && k - i < NL) acc = mad_pinned(a.l[i], b.l[k - i], acc);
    return acc;
  }
  // - s b^2 for bs = s b (scaled), nb = -b, nb2 = -2 b
  CSH_HD static int64_t col_nsqr_scaled(const FpS& bs, const FpS& nb, const FpS& nb2, int k, int64_t acc) {
    CSH_LIMB_BOUND(bs, LIM_SCALED, "col_nsqr_scaled(bs)");
    CSH_LIMB_BOUND(nb, LIM1, "col_nsqr_scaled(nb)");
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j == i) acc = mad_pinned(nb.l[i], bs.l[i], acc);
      else if (j > i && j < NL) acc = mad_pinned(nb2.l[i], bs.l[j], acc);
    }
    return acc;
  }
  CSH_HD static void mac_wide_scaled(Wide& w, const FpS& a, const FpS& b) {
    CSH_LIMB_BOUND(a, LIM_SCALED, "mac_wide_scaled(a)");
    CSH_LIMB_BOUND(b, LIM1, "mac_wide_scaled(b)");
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int j = 0; j < NL; ++j) w.t[i + j] = (int64_t)a.l[i] * (int64_t)b.l[j] + w.t[i + j];
    }
  }
  // operand products of column k
  CSH_HD static int64_t col_mul(const FpS& a, const FpS& b, int k, int64_t acc) {
    CSH_LIMB_BOUND(a, LIM2, "col_mul(a)");
    CSH_LIMB_BOUND(b, LIM1, "col_mul(b)");
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (k - i >= 0 && k - i < NL) acc = mad_pinned(a.l[i], b.l[k - i], acc);
    return acc;
  }
  // a^2: squares + doubled cross terms (a2 = 2 a)
  CSH_HD static int64_t col_sqr(const FpS& a, const FpS& a2, int k, int64_t acc) {
    CSH_LIMB_BOUND(a, LIM1, "col_sqr");
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j == i) acc = mad_pinned(a.l[i], a.l[i], acc);
      else if (j > i && j < NL) acc = mad_pinned(a2.l[i], a.l[j], acc);
    }
    return acc;
  }
  // - b^2 (nb = -b, nb2 = -2 b)
  CSH_HD static int64_t col_nsqr(const FpS& b, const FpS& nb, const FpS& nb2, int k, int64_t acc) {
    CSH_LIMB_BOUND(b, LIM1, "col_nsqr");
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j == i) acc = mad_pinned(nb.l[i], b.l[i], acc);
      else if (j > i && j < NL) acc = mad_pinned(nb2.l[i], b.l[j], acc);
    }
    return acc;
  }
  // - s_(k - NL) in the upper columns (reduce_sub's subtrahend), one multiply-add by a -1 the compiler cannot fold
  CSH_HD static int64_t col_sub_hi(const FpS& s, int32_t m1, int k, int64_t acc) {
    if (k >= NL) acc = mad_pinned_s(s.l[k - NL], m1, acc);
    return acc;
  }
  CSH_HD static int32_t opaque_minus_one() {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t m1;
    asm("s_mov_b32 %0, -1" : "=s"(m1));
    return m1;
#else
    return -1;
#endif
  }
#if CSH_REDUCE_SCAN
  CSH_HD static FpS mul(const FpS& a, const FpS& b) {
    CSH_LIMB_BOUND(a, LIM2, "mul(a)");
    CSH_LIMB_BOUND(b, LIM1, "mul(b)");
    return reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return col_mul(a, b, k, acc); });
  }
  CSH_HD static FpS sqr(const FpS& a) {
    CSH_LIMB_BOUND(a, LIM1, "sqr");
    const FpS a2 = add(a, a);
    return reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return col_sqr(a, a2, k, acc); });
  }
  CSH_HD static FpS sqr_sub(const FpS& a, const FpS& s) {  // a^2 - s
    CSH_LIMB_BOUND(a, LIM1, "sqr_sub");
    const FpS a2 = add(a, a);
    const int32_t m1 = opaque_minus_one();
    return reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return col_sub_hi(s, m1, k, col_sqr(a, a2, k, acc)); });
  }
  CSH_HD static FpS mul_sub(const FpS& a, const FpS& b, const FpS& c, const FpS& d) {  // a b - c d
    CSH_LIMB_BOUND(a, LIM1, "mul_sub(a)");
    CSH_LIMB_BOUND(b, LIM1, "mul_sub(b)");
    CSH_LIMB_BOUND(c, LIM1, "mul_sub(c)");
    CSH_LIMB_BOUND(d, LIM1, "mul_sub(d)");
    const FpS nc = neg(c);
    return reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return col_mul(nc, d, k, col_mul(a, b, k, acc)); });
  }
#else
  CSH_HD static FpS mul(const FpS& a, const FpS& b) { return reduce(mul_wide(a, b)); }
  CSH_HD static FpS sqr(const FpS& a) { return reduce(sqr_wide(a)); }
  CSH_HD static FpS sqr_sub(const FpS& a, const FpS& s) { return reduce_sub(sqr_wide(a), s); }  // a^2 - s
  CSH_HD static FpS mul_sub(const FpS& a, const FpS& b, const FpS& c, const FpS& d) { return reduce(mul_sub_wide(a, b, c, d)); }
#endif

  // exact value in [0, p) with limbs in [0, 2^B); input value must lie within (-2p, 4p)
  CSH_HD FpS canonical() const {
    FpS r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      const int32_t v = l[i] + c;
      r.l[i] = (int32_t)((uint32_t)v & LP::MASK);
      c = v >> B;
    }
    r.l[NL - 1] = l[NL - 1] + c;
    for (int rep = 0; rep < 2 && r.l[NL - 1] < 0; ++rep) {  // negative: add p
      c = 0;
#pragma unroll
      for (int i = 0; i < NL - 1; ++i) {
        const int32_t v = r.l[i] + (int32_t)LP::MOD[i] + c;
        r.l[i] = (int32_t)((uint32_t)v & LP::MASK);
        c = v >> B;
      }
      r.l[NL - 1] += (int32_t)LP::MOD[NL - 1] + c;
    }
    for (int rep = 0; rep < 4; ++rep) {  // >= p: subtract p
      int32_t d[NL];
      c = 0;
#pragma unroll
      for (int i = 0; i < NL - 1; ++i) {
        const int32_t v = r.l[i] - (int32_t)LP::MOD[i] + c;
        d[i] = (int32_t)((uint32_t)v & LP::MASK);
        c = v >> B;
      }
      d[NL - 1] = r.l[NL - 1] - (int32_t)LP::MOD[NL - 1] + c;
      if (d[NL - 1] < 0) break;
#pragma unroll
      for (int i = 0; i < NL; ++i) r.l[i] = d[i];
    }
    return r;
  }

  // 32 * x with carries propagated (limbs 0..NL-2 back in [0, 2^B), the top limb takes the growth). R' = 2^(NL*B) is
  // 2^5 * 2^256 for the 9 x 29-bit scalar fields, so  mul(a, b.times32()) = a b 2^5 / R' = a b / 2^256  is the arkworks
  // Montgomery product of two arkworks-encoded operands with no table or domain change (share-vector kernels).
  CSH_HD FpS times32() const {
    static_assert(NL * B - 32 * F32::N == 5, "R' / R must be 2^5");
    FpS r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      const int64_t t = ((int64_t)l[i] << 5) + c;
      r.l[i] = (int32_t)((uint32_t)t & LP::MASK);
      c = t >> B;
    }
    r.l[NL - 1] = (int32_t)(((int64_t)l[NL - 1] << 5) + c);
    return r;
  }

  // Partial reduction of a sum: full signed carry propagation, then subtract k*p with k estimated from the top limb
  // (float reciprocal of T + 1, T = top limb of p; exact to +-1 for |k| <= 64). Result: value in (-p - eps, 2p + eps), limbs
  // 0..NL-2 in [0, 2^B), small signed top limb -- a valid product operand. ~45 simple instructions, no multiplication by
  // a field constant. Used where sums feed sums (the s = u + v output of a decimation-in-frequency butterfly doubles
  // per stage and would leave the product routine's value range after ~6 stages).
  CSH_HD FpS fold_top() const {
    FpS r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      const int32_t v = l[i] + c;
      r.l[i] = (int32_t)((uint32_t)v & LP::MASK);
      c = v >> B;
    }
    r.l[NL - 1] = l[NL - 1] + c;
    const float inv_t = 1.0f / (float)(LP::MOD[NL - 1] + 1u);
    const int32_t k = (int32_t)floorf((float)r.l[NL - 1] * inv_t);
    int64_t cc = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      const int64_t t = (int64_t)r.l[i] - (int64_t)k * (int64_t)(int32_t)LP::MOD[i] + cc;
      r.l[i] = (int32_t)((uint32_t)t & LP::MASK);
      cc = t >> B;
    }
    r.l[NL - 1] = (int32_t)((int64_t)r.l[NL - 1] - (int64_t)k * (int64_t)(int32_t)LP::MOD[NL - 1] + cc);
    return r;
  }

  // canonical() for values that drifted further from [0, p): |value| < 32p (sums over up to ~11 butterfly stages).
  // A quotient estimate from the top limb (float reciprocal: off by at most one for |k| <= 32) brings the value into
  // (-p - eps, 2p + eps), inside canonical()'s range. Dividing by T + 1 instead of T keeps the remainder's sign.
  CSH_HD FpS canonical_wide() const { return fold_top().canonical_narrow(); }

  // Exact value in [0, p) for an input in [-p, 2p) whose limbs 0..NL-2 lie in [0, 2^B) -- what fold_top() returns, with a wide margin:
  // its quotient estimate floor(top / (T + 1)) differs from floor(value / p) only when value / p lies within ~|k| / T (1e-5 for the
  // 9 x 29-bit scalar fields) of an integer, and then by one, so its result is in (-p 1e-5, p (1 + 1e-5)). Branch-free: add p when
  // negative, then subtract p unless that borrows -- ~75 straight-line instructions where canonical()'s data-dependent correction
  // loops (a wave executes the union of its lanes' paths) cost 180-250; the canonicalise-and-store tail of an NTT pass, the kernel's
  // per-sweep fixed cost, spends most of its ~370 instructions per element there (profiles/archive/r04_e_ntt_per_pass.log).
  CSH_HD FpS canonical_narrow() const {
    FpS r;
    const int32_t m = l[NL - 1] >> 31;  // all ones when the value is negative (the top limb carries the sign)
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      const int32_t v = l[i] + ((int32_t)LP::MOD[i] & m) + c;
      r.l[i] = (int32_t)((uint32_t)v & LP::MASK);
      c = v >> B;
    }
    r.l[NL - 1] = l[NL - 1] + ((int32_t)LP::MOD[NL - 1] & m) + c;
    int32_t d[NL];
    c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      const int32_t v = r.l[i] - (int32_t)LP::MOD[i] + c;
      d[i] = (int32_t)((uint32_t)v & LP::MASK);
      c = v >> B;
    }
    d[NL - 1] = r.l[NL - 1] - (int32_t)LP::MOD[NL - 1] + c;
    const bool ge = d[NL - 1] >= 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = ge ? d[i] : r.l[i];
    return r;
  }

  // cheap necessary condition for value == 0 (mod p), valid for |value| < 8p: value = k p with |k| < 8
  CSH_HD bool maybe_zero() const {
    const uint32_t k = ((uint32_t)l[0] * LP::PINV) & LP::MASK;
    return ((k + 8u) & LP::MASK) < 16u;
  }
  CSH_HD bool is_zero_slow() const {
    FpS o = one();
    FpS y = mul(this->normalized(), o).canonical();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) acc |= (uint32_t)y.l[i];
    return acc == 0;
  }
  CSH_HD bool is_zero() const { return maybe_zero() && is_zero_slow(); }

  // ---- packed storage: canonical x*R' mod p in F32::N 32-bit words (same size as the arkworks encoding)
  CSH_HD static FpS unpack(const F32& f) {
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int bit = i * B;
      const int w = bit >> 5, off = bit & 31;
      uint64_t two = w < F32::N ? f.l[w] : 0;
      if (w + 1 < F32::N) two |= (uint64_t)f.l[w + 1] << 32;
      r.l[i] = (int32_t)((uint32_t)(two >> off) & LP::MASK);
    }
    return r;
  }
  CSH_HD F32 pack() const {  // requires canonical()
    F32 f = F32::zero();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int bit = i * B;
      const int w = bit >> 5, off = bit & 31;
      const uint64_t sh = (uint64_t)(uint32_t)l[i] << off;
      if (w < F32::N) f.l[w] |= (uint32_t)sh;
      if (w + 1 < F32::N) f.l[w + 1] |= (uint32_t)(sh >> 32);
    }
    return f;
  }
  // arkworks Montgomery (x * 2^(32N)) <-> lazy Montgomery (x * R')
  CSH_HD static FpS from_fp(const F32& f) {
    FpS c;
#pragma unroll
    for (int i = 0; i < NL; ++i) c.l[i] = (int32_t)LP::TO_LAZY[i];
    return mul(unpack(f), c);
  }
  CSH_HD F32 to_fp() const {
    FpS c;
#pragma unroll
    for (int i = 0; i < NL; ++i) c.l[i] = (int32_t)LP::FROM_LAZY[i];
    return mul(this->normalized(), c).canonical().pack();
  }
  // x*2^(32N) (arkworks) -> packed canonical x*R' (what Bases keeps on the device for lazy curves)
  CSH_HD static F32 repack_for_storage(const F32& f) { return from_fp(f).canonical().pack(); }
};

using Fq29s = FpS<Bn254Fq29Params, Bn254Fq>;
using Fq28s = FpS<Bls381Fq28Params, Bls381Fq>;
using Fq28s377 = FpS<Bls377Fq28Params, Bls377Fq>;
using Fr29s = FpS<Bn254Fr29Params, Bn254Fr>;  // Grumpkin base field; BN254 NTT butterflies
using Bls381Fr29s = FpS<Bls381Fr29Params, Bls381Fr>;
using Bls377Fr29s = FpS<Bls377Fr29Params, Bls377Fr>;

// scalar field -> its lazy representation (NTT butterflies, share-vector kernels)
template <class F>
struct LazyOf;
template <>
struct LazyOf<Bn254Fr> { using type = Fr29s; };
template <>
struct LazyOf<Bls381Fr> { using type = Bls381Fr29s; };
template <>
struct LazyOf<Bls377Fr> { using type = Bls377Fr29s; };

// ---- Fp2 = Fp[i]/(i^2 + NR) over the signed lazy field: schoolbook products accumulated double-width with ONE
// reduction per output component (2 NL^2 + NL^2 mads per component, cheaper than Karatsuba's three full
// multiplications and free of the 2^(B+1)-limb operand sums Karatsuba needs).
// NR = 1 (BN254, BLS12-381) or 5 (BLS12-377, F32x2::NONRESIDUE_NEG): the factor rides on ONE operand of the a1 b1 products
// (limb-wise -5 a1: |limb| < 5 (2^28 + 8) < 2^31). Column bound with NR = 5, 14 x 28-bit limbs, every operand normalised:
// (1 + 5) 14 2^56 products + 14 2^56 of the reduction = 98 2^56 < 2^63; four-product sums take the carry sweep between the pairs.
template <class LF, class F32x2>
struct Fp2S {
  static constexpr int NR = F32x2::NONRESIDUE_NEG;
  static_assert(NR == 1 || (NR == 5 && (6.0 * LF::NL + LF::NL) * (double)(1ull << (2 * LF::B - 40)) < (double)(1ull << 23)), "Fp2S: non-residue / limb layout");
  LF c0, c1;
  CSH_HD static Fp2S zero() { return {LF::zero(), LF::zero()}; }
  CSH_HD static Fp2S one() { return {LF::one(), LF::zero()}; }
  CSH_HD static Fp2S add(const Fp2S& a, const Fp2S& b) { return {LF::add(a.c0, b.c0), LF::add(a.c1, b.c1)}; }
  CSH_HD static Fp2S sub(const Fp2S& a, const Fp2S& b) { return {LF::sub(a.c0, b.c0), LF::sub(a.c1, b.c1)}; }
  CSH_HD static Fp2S neg(const Fp2S& a) { return {LF::neg(a.c0), LF::neg(a.c1)}; }
  CSH_HD Fp2S normalized() const { return {c0.normalized(), c1.normalized()}; }
  CSH_HD Fp2S neg_unpacked() const { return {c0.neg_unpacked(), c1.neg_unpacked()}; }
  CSH_HD Fp2S cneg_unpacked(uint32_t neg01) const { return {c0.cneg_unpacked(neg01), c1.cneg_unpacked(neg01)}; }
  // ---- NR != 1 (BLS12-377): the same sums with the factor on one operand; no four-product columns ----------------------
  CSH_HD static Fp2S mul_nr(const Fp2S& a, const Fp2S& b) {
    const LF na1 = LF::scaled(a.c1, -NR);
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_mul_scaled(na1, b.c1, k, LF::col_mul(a.c0, b.c0, k, acc)); }),
            LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_mul(a.c1, b.c0, k, LF::col_mul(a.c0, b.c1, k, acc)); })};
  }
  CSH_HD static Fp2S sqr_nr(const Fp2S& a) {
    const LF a2 = LF::add(a.c0, a.c0), bs = LF::scaled(a.c1, NR), nb = LF::neg(a.c1), nb2 = LF::add(nb, nb);
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_nsqr_scaled(bs, nb, nb2, k, LF::col_sqr(a.c0, a2, k, acc)); }),
            LF::mul(a2, a.c1)};
  }
  CSH_HD static Fp2S sqr_sub_nr(const Fp2S& a, const Fp2S& s) {
    const LF a2 = LF::add(a.c0, a.c0), bs = LF::scaled(a.c1, NR), nb = LF::neg(a.c1), nb2 = LF::add(nb, nb);
    const int32_t m1 = LF::opaque_minus_one();
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_sub_hi(s.c0, m1, k, LF::col_nsqr_scaled(bs, nb, nb2, k, LF::col_sqr(a.c0, a2, k, acc))); }),
            LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_sub_hi(s.c1, m1, k, LF::col_mul(a2, a.c1, k, acc)); })};
  }
  CSH_HD static Fp2S mul_sub_nr(const Fp2S& a, const Fp2S& b, const Fp2S& c, const Fp2S& d) {
    typename LF::Wide w0 = LF::mul_wide(a.c0, b.c0);                   //   a0 b0 - NR a1 b1
    LF::mac_wide_scaled(w0, LF::scaled(a.c1, -NR), b.c1);
    LF::compress_wide(w0);
    typename LF::Wide v0 = LF::mul_wide(LF::neg(c.c0), d.c0);          // - c0 d0 + NR c1 d1
    LF::mac_wide_scaled(v0, LF::scaled(c.c1, NR), d.c1);
    LF::add_wide(w0, v0);
    typename LF::Wide w1 = LF::mul_add_wide(a.c0, b.c1, a.c1, b.c0);   //   a0 b1 + a1 b0
    LF::compress_wide(w1);
    typename LF::Wide v1 = LF::mul_wide(LF::neg(c.c0), d.c1);          // - c0 d1 - c1 d0
    LF::mac_wide(v1, c.c1, d.c0, true);
    LF::add_wide(w1, v1);
    return {LF::reduce(w0), LF::reduce(w1)};
  }
#if CSH_REDUCE_SCAN
  CSH_HD static Fp2S mul(const Fp2S& a, const Fp2S& b) {
    if constexpr (NR != 1) return mul_nr(a, b);
    const LF na1 = LF::neg(a.c1);
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_mul(na1, b.c1, k, LF::col_mul(a.c0, b.c0, k, acc)); }),
            LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_mul(a.c1, b.c0, k, LF::col_mul(a.c0, b.c1, k, acc)); })};
  }
  CSH_HD static Fp2S sqr(const Fp2S& a) {
    if constexpr (NR != 1) return sqr_nr(a);
    const LF a2 = LF::add(a.c0, a.c0), nb = LF::neg(a.c1), nb2 = LF::add(nb, nb);
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_nsqr(a.c1, nb, nb2, k, LF::col_sqr(a.c0, a2, k, acc)); }),
            LF::mul(a2, a.c1)};
  }
  // a^2 - s (s: |limb| < 2^31)
  CSH_HD static Fp2S sqr_sub(const Fp2S& a, const Fp2S& s) {
    if constexpr (NR != 1) return sqr_sub_nr(a, s);
    const LF a2 = LF::add(a.c0, a.c0), nb = LF::neg(a.c1), nb2 = LF::add(nb, nb);
    const int32_t m1 = LF::opaque_minus_one();
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_sub_hi(s.c0, m1, k, LF::col_nsqr(a.c1, nb, nb2, k, LF::col_sqr(a.c0, a2, k, acc))); }),
            LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_sub_hi(s.c1, m1, k, LF::col_mul(a2, a.c1, k, acc)); })};
  }
#else
  static_assert(NR == 1, "the row-wise build has no BLS12-377 Fp2");
  CSH_HD static Fp2S mul(const Fp2S& a, const Fp2S& b) {
    return {LF::reduce(LF::mul_sub_wide(a.c0, b.c0, a.c1, b.c1)), LF::reduce(LF::mul_add_wide(a.c0, b.c1, a.c1, b.c0))};
  }
  CSH_HD static Fp2S sqr(const Fp2S& a) {
    return {LF::reduce(LF::sqr_sub_wide(a.c0, a.c1)), LF::mul(LF::add(a.c0, a.c0), a.c1)};
  }
  // a^2 - s (s: |limb| < 2^31)
  CSH_HD static Fp2S sqr_sub(const Fp2S& a, const Fp2S& s) {
    return {LF::reduce_sub(LF::sqr_sub_wide(a.c0, a.c1), s.c0), LF::reduce_sub(LF::mul_wide(LF::add(a.c0, a.c0), a.c1), s.c1)};
  }
#endif
  // a*b - c*d
  CSH_HD static Fp2S mul_sub(const Fp2S& a, const Fp2S& b, const Fp2S& c, const Fp2S& d) {
    if constexpr (NR != 1) return mul_sub_nr(a, b, c, d);
    if constexpr (LF::FOUR_PRODUCTS_FIT) {
#if CSH_REDUCE_SCAN
      const LF na1 = LF::neg(a.c1), nc0 = LF::neg(c.c0), nc1 = LF::neg(c.c1);
      return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE {
                return LF::col_mul(c.c1, d.c1, k, LF::col_mul(nc0, d.c0, k, LF::col_mul(na1, b.c1, k, LF::col_mul(a.c0, b.c0, k, acc))));
              }),
              LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE {
                return LF::col_mul(nc1, d.c0, k, LF::col_mul(nc0, d.c1, k, LF::col_mul(a.c1, b.c0, k, LF::col_mul(a.c0, b.c1, k, acc))));
              })};
#else
      typename LF::Wide w0 = LF::mul_sub_wide(a.c0, b.c0, a.c1, b.c1);
      LF::mac_wide(w0, c.c0, d.c0, true);
      LF::mac_wide(w0, c.c1, d.c1, false);
      typename LF::Wide w1 = LF::mul_add_wide(a.c0, b.c1, a.c1, b.c0);
      LF::mac_wide(w1, c.c0, d.c1, true);
      LF::mac_wide(w1, c.c1, d.c0, true);
      return {LF::reduce(w0), LF::reduce(w1)};
#endif
    } else {
      // two products per accumulator, a carry sweep between the pairs, ONE reduction per component (instead of two full Fp2
      // products = four reductions, a limb-wise subtraction and a carry step)
      typename LF::Wide w0 = LF::mul_sub_wide(a.c0, b.c0, a.c1, b.c1);   //   a0 b0 - a1 b1
      LF::compress_wide(w0);
      LF::add_wide(w0, LF::mul_sub_wide(c.c1, d.c1, c.c0, d.c0));        // + c1 d1 - c0 d0
      typename LF::Wide w1 = LF::mul_add_wide(a.c0, b.c1, a.c1, b.c0);   //   a0 b1 + a1 b0
      LF::compress_wide(w1);
      typename LF::Wide v1 = LF::mul_wide(LF::neg(c.c0), d.c1);          // - c0 d1 - c1 d0
      LF::mac_wide(v1, c.c1, d.c0, true);
      LF::add_wide(w1, v1);
      return {LF::reduce(w0), LF::reduce(w1)};
    }
  }
  CSH_HD bool maybe_zero() const { return c0.maybe_zero() && c1.maybe_zero(); }
  CSH_HD bool is_zero_slow() const { return c0.is_zero_slow() && c1.is_zero_slow(); }
  CSH_HD bool is_zero() const { return maybe_zero() && is_zero_slow(); }
  CSH_HD static Fp2S unpack(const F32x2& f) { return {LF::unpack(f.c0), LF::unpack(f.c1)}; }
  CSH_HD F32x2 to_fp() const { return {c0.to_fp(), c1.to_fp()}; }
  CSH_HD static F32x2 repack_for_storage(const F32x2& f) { return {LF::repack_for_storage(f.c0), LF::repack_for_storage(f.c1)}; }
};
using Fq29s2 = Fp2S<Fq29s, Bn254Fq2>;
using Fq28s2 = Fp2S<Fq28s, Bls381Fq2>;
using Fq28s377x2 = Fp2S<Fq28s377, Bls377Fq2>;

}  // namespace csh
