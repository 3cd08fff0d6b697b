// Bucket accumulation in the signed lazy field (field29.hpp FpS): XYZZ += affine, 8M + 2S with limb-wise
// subtractions and no modular corrections. Same formulas as curve.hpp (EFD madd-2008-s / mdbl-2008-s-1); the
// explicit `empty` flag replaces the ZZ == 0 test, and the P == 0 (doubling / cancellation) case is detected
// with a 3-instruction filter on the low limb before the exact check.
#pragma once
#include <type_traits>

#include "curve.hpp"
#include "field29.hpp"

namespace csh {

template <class L>
struct XYZZLazy {
  L x, y, zz, zzz;
  bool empty;
  CSH_HD static XYZZLazy inf() { return {L::zero(), L::zero(), L::zero(), L::zero(), true}; }
};

// 2 * (x, y) for an affine point with y != 0 (rare path)
// (big aggregates cross the call boundary through memory, by pointer: passing ~450-byte structs by value inside the
// already register-saturated wide-field kernels proved fragile)
template <class L>
CSH_HD XYZZLazy<L> lazy_mdbl_inl(const L& x, const L& y) {
  L u = L::add(y, y).normalized();
  L v = L::sqr(u);
  L w = L::mul(u, v);
  L s = L::mul(x, v);
  L xx = L::sqr(x);
  L m = L::add(L::add(xx, xx), xx).normalized();
  XYZZLazy<L> r;
  r.x = L::sqr_sub(m, L::add(s, s));
  r.y = L::mul_sub(m, L::sub(s, r.x), w, y);
  r.zz = v;
  r.zzz = w;
  r.empty = false;
  return r;
}
template <class L>
CSH_HD_NOINLINE void lazy_mdbl(const L* x, const L* y, XYZZLazy<L>* out) {
  *out = lazy_mdbl_inl<L>(*x, *y);
}
template <class L>
CSH_HD_NOINLINE XYZZLazy<L> lazy_mdbl_v(L x, L y) {  // by value: fine (and faster around the call) for the 9-limb field
  return lazy_mdbl_inl<L>(x, y);
}

// Fp2 values split over a lane pair (curve_pair.hpp): the lane's half of an affine point comes from memory by role
template <class LF, int NR>
struct Fp2Pair;
template <class L>
struct IsPair { static constexpr bool value = false; };
template <class LF, int NR>
struct IsPair<Fp2Pair<LF, NR>> { static constexpr bool value = true; };
template <class L, class AffT>
__device__ void pair_unpack_affine(const AffT* src, L* x, L* y);

// Rare-path helper of the accumulate kernel for the wide fields: 2 * (the affine point at `src`, negated if asked), re-read
// from memory INSIDE the out-of-line routine. (The former version took private stack copies of x2 / y2 in the caller's rare
// branch; the compiler hoisted those stores to the top of the loop body, so every mixed addition wrote 2 field elements of
// scratch -- 2.5 GB per 2^20 BN254 G2 MSM, 4.2 GB on BLS12-381 G2 by WRITE_SIZE, profiles/archive/r02_a_msm_*_pmc_hbm_bytes.csv.)
template <class L, class AffT>
CSH_HD_NOINLINE void lazy_mdbl_mem(const AffT* src, uint32_t negate, XYZZLazy<L>* out) {
  L x, y;
  if constexpr (IsPair<L>::value) {
    pair_unpack_affine<L, AffT>(src, &x, &y);
  } else {
    const AffT pt = *src;
    x = L::unpack(pt.x);
    y = L::unpack(pt.y);
  }
  if (negate) y = y.neg_unpacked();
  *out = lazy_mdbl_inl<L>(x, y);
}

// acc += (x2, y2); the caller has already excluded the point at infinity. Contract: x2, y2 have limbs 0..NL-2 in
// [-2, 2^B + 2] and a small signed top limb (unpack() output, or .neg_unpacked() of it for a negated point) -- acc.x / acc.y inherit that bound
// and are subtracted limb-wise from fresh products below. src (nullable) / negate: where (x2, y2) came from, for the rare
// doubling path of the wide fields.
template <class L, class AffT = void>
CSH_HD void lazy_madd(XYZZLazy<L>& acc, const L& x2, const L& y2, const AffT* src = nullptr, uint32_t negate = 0) {
  if (acc.empty) {
    acc.x = x2;
    acc.y = y2;
    acc.zz = L::one();
    acc.zzz = L::one();
    acc.empty = false;
    return;
  }
  const L u2 = L::mul(x2, acc.zz);
  const L s2 = L::mul(y2, acc.zzz);
  const L p = L::sub(u2, acc.x);
  const L r = L::sub(s2, acc.y);
  if (p.maybe_zero()) {
    if (p.is_zero_slow()) {
      if (r.is_zero()) {
        if (y2.is_zero()) acc.empty = true;  // 2-torsion cannot occur on these curves; kept for completeness
        else if constexpr (sizeof(L) <= 40) {
          acc = lazy_mdbl_v<L>(x2, y2);
        } else if constexpr (!std::is_void<AffT>::value) {
          XYZZLazy<L> d;
          lazy_mdbl_mem<L, AffT>(src, negate, &d);
          acc = d;
        } else {
          const L tx = x2, ty = y2;  // private copies: the call takes addresses
          XYZZLazy<L> d;
          lazy_mdbl<L>(&tx, &ty, &d);
          acc = d;
        }
      } else {
        acc.empty = true;  // P + (-P)
      }
      return;
    }
  }
  const L pp = L::sqr(p);
  const L ppp = L::mul(p, pp);
  const L q = L::mul(acc.x, pp);
  const L x3 = L::sqr_sub(r, L::add(ppp, L::add(q, q)));   // r^2 - ppp - 2q, normalised by the reduction's own carry chain
  const L y3 = L::mul_sub(r, L::sub(q, x3), acc.y, ppp);   // r*(q - x3) - y1*ppp, one reduction
  acc.x = x3;
  acc.y = y3;
  acc.zz = L::mul(acc.zz, pp);
  acc.zzz = L::mul(acc.zzz, ppp);
}

// ---- general XYZZ arithmetic in the lazy field (bucket merge / window reduction kernels) -------------------------
// Stored points are XYZZLazy values exactly as the accumulate kernel leaves them: x normalised, y / zz / zzz straight
// out of a Montgomery reduction (|limb| <= 2^B + 1), so they are valid product operands again with no canonical form
// in between.

// 2 * P (EFD dbl-2008-s-1, a = 0): 6M + 3S... here 5 products + 3 squares with the fused M*(S - X3) - W*Y1
template <class L>
CSH_HD XYZZLazy<L> lazy_dbl_inl(const XYZZLazy<L>& p) {
  if (p.empty) return p;
  if (p.y.is_zero()) return XYZZLazy<L>::inf();  // 2-torsion: not on these curves, kept for completeness
  const L u = L::add(p.y, p.y).normalized();
  const L v = L::sqr(u);
  const L w = L::mul(u, v);
  const L s = L::mul(p.x, v);
  const L xx = L::sqr(p.x);
  const L m = L::add(L::add(xx, xx), xx).normalized();
  XYZZLazy<L> r;
  r.x = L::sqr_sub(m, L::add(s, s));
  r.y = L::mul_sub(m, L::sub(s, r.x), w, p.y);
  r.zz = L::mul(v, p.zz);
  r.zzz = L::mul(w, p.zzz);
  r.empty = false;
  return r;
}
template <class L>
CSH_HD_NOINLINE void lazy_dbl_p(const XYZZLazy<L>* p, XYZZLazy<L>* out) {
  *out = lazy_dbl_inl<L>(*p);
}

// acc += p (EFD add-2008-s): 12 products + 2 squares, 13 reductions
template <class L>
CSH_HD void lazy_add_inl(XYZZLazy<L>& acc, const XYZZLazy<L>& p) {
  if (p.empty) return;
  if (acc.empty) {
    acc = p;
    return;
  }
  const L u1 = L::mul(acc.x, p.zz);
  const L u2 = L::mul(p.x, acc.zz);
  const L s1 = L::mul(acc.y, p.zzz);
  const L s2 = L::mul(p.y, acc.zzz);
  const L pd = L::sub(u2, u1);
  const L r = L::sub(s2, s1);
  if (pd.maybe_zero()) {
    if (pd.is_zero_slow()) {
      if (r.is_zero()) {
        const XYZZLazy<L> t = acc;
        lazy_dbl_p<L>(&t, &acc);
      } else {
        acc.empty = true;  // P + (-P)
      }
      return;
    }
  }
  const L pp = L::sqr(pd);
  const L ppp = L::mul(pd, pp);
  const L q = L::mul(u1, pp);
  const L x3 = L::sqr_sub(r, L::add(ppp, L::add(q, q)));
  const L y3 = L::mul_sub(r, L::sub(q, x3), s1, ppp);
  acc.x = x3;
  acc.y = y3;
  acc.zz = L::mul(L::mul(acc.zz, p.zz), pp);
  acc.zzz = L::mul(L::mul(acc.zzz, p.zzz), ppp);
}
// out of line, operands through memory (see the note on lazy_mdbl)
template <class L>
CSH_HD_NOINLINE void lazy_add_p(XYZZLazy<L>* acc, const XYZZLazy<L>* p) {
  XYZZLazy<L> a = *acc;
  lazy_add_inl<L>(a, *p);
  *acc = a;
}

// k * P for a small unsigned k (double-and-add). INL: both steps inlined into the loop body (9-limb field; the
// window reduction spends half its time here), otherwise out-of-line pieces.
template <class L, bool INL = false>
CSH_HD XYZZLazy<L> lazy_mul_small(const XYZZLazy<L>& p, uint32_t k) {
  XYZZLazy<L> r = XYZZLazy<L>::inf();
  if (k == 0 || p.empty) return r;
  int top = 31;
  while (!((k >> top) & 1)) --top;
  for (int b = top; b >= 0; --b) {
    if constexpr (INL) {
      r = lazy_dbl_inl<L>(r);
      if ((k >> b) & 1) lazy_add_inl<L>(r, p);
    } else {
      if (!r.empty) {
        const XYZZLazy<L> t = r;
        lazy_dbl_p<L>(&t, &r);
      }
      if ((k >> b) & 1) lazy_add_p<L>(&r, &p);
    }
  }
  return r;
}

template <class L, class F32>
CSH_HD XYZZ<F32> lazy_to_xyzz(const XYZZLazy<L>& a) {
  if (a.empty) return XYZZ<F32>::inf();
  return {a.x.to_fp(), a.y.to_fp(), a.zz.to_fp(), a.zzz.to_fp()};
}

}  // namespace csh
