// Fp2 elements distributed over a lane pair, for the G2 bucket kernels.
//
// Why: with whole Fp2 points in one lane the G2 accumulate kernels need 298 (BN254) / 474 (BLS12-381) VGPRs -> one wave per
// SIMD, nothing to hide a gather or a dependent-issue stall behind (SQ counters: 0.68 / 0.81 of the cycles busy), and the
// lane-serial window reduction spills (437 / 512 VGPRs, 1.2-1.8 KB of scratch). An Fp2 product splits into two halves of equal
// cost that share their operands:  c0 = a0 b0 - a1 b1,  c1 = a0 b1 + a1 b0.  Here lane 2k holds component 0 and lane 2k+1
// component 1 of every Fp2 value of one point; a product moves three operands across the pair with v_mov_b32_dpp quad_perm
// (one VALU instruction per limb, no LDS) and then runs the same two accumulating 9x9 / 14x14 limb products + ONE Montgomery
// reduction on both lanes. Register footprint per lane ~ that of the G1 kernel over the same base field, two waves per SIMD.
//
// Fp2Pair<LF> offers the interface curve_lazy.hpp is written against (add / sub / neg / normalized / mul / sqr / mul_sub /
// zero tests), so lazy_madd, lazy_add_inl, lazy_dbl_inl and lazy_mul_small are reused unchanged with two lanes per point.
// Every predicate is combined over the pair before it reaches control flow: the two lanes of a pair never diverge.
// Operand bounds: exactly those of Fp2S (field29.hpp) -- the same products with the same operands, on two lanes.
// Memory keeps the XYZZLazy<Fp2S<LF>> / Affine<Fq2> layouts: the lane with role r touches component r of each member.
#pragma once
#include <stddef.h>

#include "curve_lazy.hpp"
#include "curve_quad.hpp"

namespace csh {

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

using PairSwap = QuadCtrl<1, 0, 3, 2>;  // the partner lane's value
using PairLo = QuadCtrl<0, 0, 2, 2>;    // component 0 on both lanes
using PairHi = QuadCtrl<1, 1, 3, 3>;    // component 1 on both lanes

__device__ __forceinline__ int pair_role() { return (int)(threadIdx.x & 1u); }
__device__ __forceinline__ bool pair_all(bool f) { return f && (quad_mov<PairSwap::value>((int32_t)f) != 0); }

// NR: Fp2 = Fp[i]/(i^2 + NR) -- 1 on BN254 / BLS12-381, 5 on BLS12-377 (the factor rides on the operand hi_signed() moves across the
// pair; four-product columns then no longer fit: mul_sub takes the carry sweep between its two pairs, as the 9 x 29-bit field does)
template <class LF, int NR = 1>
struct Fp2Pair {
  using Base = LF;
  static constexpr bool FOUR_FIT = LF::FOUR_PRODUCTS_FIT && NR == 1;
  static constexpr int NL = LF::NL;
  static constexpr int B = LF::B;
  LF v;  // component pair_role() of the element

  __device__ __forceinline__ static Fp2Pair zero() { return {LF::zero()}; }
  __device__ __forceinline__ static Fp2Pair one() {
    Fp2Pair r{LF::one()};
    if (pair_role()) r.v = LF::zero();
    return r;
  }
  __device__ __forceinline__ static Fp2Pair add(const Fp2Pair& a, const Fp2Pair& b) { return {LF::add(a.v, b.v)}; }
  __device__ __forceinline__ static Fp2Pair sub(const Fp2Pair& a, const Fp2Pair& b) { return {LF::sub(a.v, b.v)}; }
  __device__ __forceinline__ static Fp2Pair neg(const Fp2Pair& a) { return {LF::neg(a.v)}; }
  __device__ __forceinline__ Fp2Pair normalized() const { return {v.normalized()}; }
  __device__ __forceinline__ Fp2Pair neg_unpacked() const { return {v.neg_unpacked()}; }
  __device__ __forceinline__ Fp2Pair cneg_unpacked(uint32_t neg01) const { return {v.cneg_unpacked(neg01)}; }

  // component 1 of `a` on both lanes, times -NR on the lane that computes c0 = a0 b0 - NR a1 b1
  __device__ __forceinline__ static LF hi_signed(const LF& a) {
    const int32_t sgn = pair_role() ? 1 : -NR;
    LF r = quad_perm<PairHi::value>(a);
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] *= sgn;
    return r;
  }
  // role 0: a0 b0 - a1 b1, role 1: a0 b1 + a1 b0 (unreduced)
  __device__ __forceinline__ static typename LF::Wide mul_wide(const Fp2Pair& a, const Fp2Pair& b) {
    typename LF::Wide w = LF::mul_wide(quad_perm<PairLo::value>(a.v), b.v);
    LF::mac_wide(w, hi_signed(a.v), quad_perm<PairSwap::value>(b.v), false);
    return w;
  }
#if CSH_REDUCE_SCAN
  __device__ __forceinline__ static Fp2Pair mul(const Fp2Pair& a, const Fp2Pair& b) {
    const LF alo = quad_perm<PairLo::value>(a.v), ahi = hi_signed(a.v), bsw = quad_perm<PairSwap::value>(b.v);
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_mul(ahi, bsw, k, LF::col_mul(alo, b.v, k, acc)); })};
  }
  __device__ __forceinline__ static Fp2Pair sqr(const Fp2Pair& a) { return mul(a, a); }
  __device__ __forceinline__ static Fp2Pair sqr_sub(const Fp2Pair& a, const Fp2Pair& s) {
    const LF alo = quad_perm<PairLo::value>(a.v), ahi = hi_signed(a.v), bsw = quad_perm<PairSwap::value>(a.v);
    const int32_t m1 = LF::opaque_minus_one();
    return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE { return LF::col_sub_hi(s.v, m1, k, LF::col_mul(ahi, bsw, k, LF::col_mul(alo, a.v, k, acc))); })};
  }
#else
  __device__ __forceinline__ static Fp2Pair mul(const Fp2Pair& a, const Fp2Pair& b) { return {LF::reduce(mul_wide(a, b))}; }
  __device__ __forceinline__ static Fp2Pair sqr(const Fp2Pair& a) { return mul(a, a); }
  __device__ __forceinline__ static Fp2Pair sqr_sub(const Fp2Pair& a, const Fp2Pair& s) { return {LF::reduce_sub(mul_wide(a, a), s.v)}; }
#endif
  // a*b - c*d
  __device__ __forceinline__ static Fp2Pair mul_sub(const Fp2Pair& a, const Fp2Pair& b, const Fp2Pair& c, const Fp2Pair& d) {
    if constexpr (FOUR_FIT) {
#if CSH_REDUCE_SCAN
      const LF alo = quad_perm<PairLo::value>(a.v), ahi = hi_signed(a.v), bsw = quad_perm<PairSwap::value>(b.v);
      const LF nclo = LF::neg(quad_perm<PairLo::value>(c.v)), nchi = LF::neg(hi_signed(c.v)), dsw = quad_perm<PairSwap::value>(d.v);
      return {LF::reduce_scan([&](int k, int64_t acc) CSH_LAMBDA_INLINE {
        return LF::col_mul(nchi, dsw, k, LF::col_mul(nclo, d.v, k, LF::col_mul(ahi, bsw, k, LF::col_mul(alo, b.v, k, acc))));
      })};
#endif
      typename LF::Wide w = mul_wide(a, b);
      LF::mac_wide(w, quad_perm<PairLo::value>(c.v), d.v, true);
      LF::mac_wide(w, hi_signed(c.v), quad_perm<PairSwap::value>(d.v), true);
      return {LF::reduce(w)};
    } else {
      typename LF::Wide w = mul_wide(a, b);  // two products, then a carry sweep so that the next two fit the same columns
      LF::compress_wide(w);
      typename LF::Wide v = LF::mul_wide(LF::neg(quad_perm<PairLo::value>(c.v)), d.v);
      LF::mac_wide(v, hi_signed(c.v), quad_perm<PairSwap::value>(d.v), true);
      LF::add_wide(w, v);
      return {LF::reduce(w)};
    }
  }
  __device__ __forceinline__ bool maybe_zero() const { return pair_all(v.maybe_zero()); }
  __device__ __forceinline__ bool is_zero_slow() const { return pair_all(v.is_zero_slow()); }
  __device__ __forceinline__ bool is_zero() const { return maybe_zero() && is_zero_slow(); }
};

// ---- memory: the lane's half of points stored in the whole-element layouts -----------------------------------------------
template <class LF, class F2>
__device__ __forceinline__ XYZZLazy<Fp2Pair<LF, F2::NONRESIDUE_NEG>> pair_load(const XYZZLazy<Fp2S<LF, F2>>* p, int role) {
  using Whole = XYZZLazy<Fp2S<LF, F2>>;
  static_assert(sizeof(Fp2S<LF, F2>) == 2 * sizeof(LF) && offsetof(Whole, zzz) == 6 * sizeof(LF), "XYZZLazy<Fp2S> must be 8 consecutive base elements");
  const LF* f = reinterpret_cast<const LF*>(p) + role;
  XYZZLazy<Fp2Pair<LF, F2::NONRESIDUE_NEG>> r;
  r.empty = p->empty;
  r.x.v = f[0];
  r.y.v = f[2];
  r.zz.v = f[4];
  r.zzz.v = f[6];
  return r;
}
template <class LF, class F2>
__device__ __forceinline__ void pair_store(XYZZLazy<Fp2S<LF, F2>>* p, int role, const XYZZLazy<Fp2Pair<LF, F2::NONRESIDUE_NEG>>& q) {
  LF* f = reinterpret_cast<LF*>(p) + role;
  f[0] = q.x.v;
  f[2] = q.y.v;
  f[4] = q.zz.v;
  f[6] = q.zzz.v;
  if (role == 0) p->empty = q.empty;
}

// Affine<Fq2> in the stored encoding (packed canonical x R'): component `role` of x and y, unpacked
template <class L, class AffT>
__device__ __forceinline__ void pair_unpack_affine(const AffT* src, L* x, L* y) {
  using LF = typename L::Base;
  using F32 = decltype(src->x.c0);
  static_assert(sizeof(AffT) == 4 * sizeof(F32), "Affine<Fq2> must be four base-field elements");
  const F32* f = reinterpret_cast<const F32*>(src) + pair_role();
  x->v = LF::unpack(f[0]);
  y->v = LF::unpack(f[2]);
}

#endif  // device

}  // namespace csh
