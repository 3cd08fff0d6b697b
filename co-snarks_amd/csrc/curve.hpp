// Short-Weierstrass group law (a = 0: BN254 G1/G2, BLS12-381 G1/G2), generic over the base field
// (Fp or Fp2). Shared by the gfx950 MSM kernels and the host glue.
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// mixed addition of an affine base costs 8M+2S with no inversion, the cheapest complete-enough form
// for Pippenger buckets. Formulas: the published EFD "madd-2008-s", "add-2008-s", "dbl-2008-s-1",
// "mdbl-2008-s-1". Wire formats: affine = x || y with infinity encoded as all-zero (zkey convention);
// results leave as arkworks `Projective` = Jacobian (X, Y, Z), infinity = (1, 1, 0).
#pragma once
#include "field.hpp"

namespace csh {

template <class F>
struct Affine {
  F x, y;
  CSH_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  CSH_HD static Affine inf() { return {F::zero(), F::zero()}; }
};

template <class F>
struct Jac {
  F x, y, z;
  CSH_HD bool is_inf() const { return z.is_zero(); }
  CSH_HD static Jac inf() { return {F::one(), F::one(), F::zero()}; }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  CSH_HD bool is_inf() const { return zz.is_zero(); }
  CSH_HD static XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  CSH_HD static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one(), F::one()};
  }
};

// 2*P for affine P (mdbl-2008-s-1), P != inf
template <class F>
CSH_HD_NOINLINE XYZZ<F> xyzz_mdbl(Affine<F> p) {
  if (p.y.is_zero()) return XYZZ<F>::inf();
  F u = F::mul2(p.y);
  F v = F::sqr(u);
  F w = F::mul(u, v);
  F s = F::mul(p.x, v);
  F m = F::mul3(F::sqr(p.x));
  XYZZ<F> r;
  r.x = F::sub(F::sqr(m), F::mul2(s));
  r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// acc += P (affine, P != inf handled by caller or here)
template <class F>
CSH_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& p) {
  if (p.is_inf()) return;
  if (acc.is_inf()) {
    acc = {p.x, p.y, F::one(), F::one()};
    return;
  }
  F u2 = F::mul(p.x, acc.zz);
  F s2 = F::mul(p.y, acc.zzz);
  F pp_ = F::sub(u2, acc.x);
  F r = F::sub(s2, acc.y);
  if (pp_.is_zero()) {
    if (r.is_zero()) {
      acc = xyzz_mdbl(p);
    } else {
      acc = XYZZ<F>::inf();
    }
    return;
  }
  F pp = F::sqr(pp_);
  F ppp = F::mul(pp_, pp);
  F q = F::mul(acc.x, pp);
  F x3 = F::sub(F::sub(F::sqr(r), ppp), F::mul2(q));
  F y3 = F::sub(F::mul(r, F::sub(q, x3)), F::mul(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = F::mul(acc.zz, pp);
  acc.zzz = F::mul(acc.zzz, ppp);
}

template <class F>
CSH_HD XYZZ<F> xyzz_dbl_inl(const XYZZ<F>& p) {
  if (p.is_inf() || p.y.is_zero()) return XYZZ<F>::inf();
  F u = F::mul2(p.y);
  F v = F::sqr(u);
  F w = F::mul(u, v);
  F s = F::mul(p.x, v);
  F m = F::mul3(F::sqr(p.x));
  XYZZ<F> r;
  r.x = F::sub(F::sqr(m), F::mul2(s));
  r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
  r.zz = F::mul(v, p.zz);
  r.zzz = F::mul(w, p.zzz);
  return r;
}
template <class F>
CSH_HD_NOINLINE XYZZ<F> xyzz_dbl(XYZZ<F> p) {
  return xyzz_dbl_inl(p);
}

template <class F>
CSH_HD XYZZ<F> xyzz_add_inl(XYZZ<F> acc, const XYZZ<F>& p) {
  if (p.is_inf()) return acc;
  if (acc.is_inf()) return p;
  F u1 = F::mul(acc.x, p.zz);
  F u2 = F::mul(p.x, acc.zz);
  F s1 = F::mul(acc.y, p.zzz);
  F s2 = F::mul(p.y, acc.zzz);
  F pp_ = F::sub(u2, u1);
  F r = F::sub(s2, s1);
  if (pp_.is_zero()) {
    if (r.is_zero()) return xyzz_dbl(acc);
    return XYZZ<F>::inf();
  }
  F pp = F::sqr(pp_);
  F ppp = F::mul(pp_, pp);
  F q = F::mul(u1, pp);
  F x3 = F::sub(F::sub(F::sqr(r), ppp), F::mul2(q));
  F y3 = F::sub(F::mul(r, F::sub(q, x3)), F::mul(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = F::mul(F::mul(acc.zz, p.zz), pp);
  acc.zzz = F::mul(F::mul(acc.zzz, p.zzz), ppp);
  return acc;
}
template <class F>
CSH_HD_NOINLINE XYZZ<F> xyzz_add_v(XYZZ<F> acc, XYZZ<F> p) {
  return xyzz_add_inl(acc, p);
}
template <class F>
CSH_HD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& p) {
  acc = xyzz_add_v(acc, p);
}

template <class F>
CSH_HD XYZZ<F> xyzz_neg(const XYZZ<F>& p) {
  return {p.x, F::neg(p.y), p.zz, p.zzz};
}

// k * P for a small unsigned k (double-and-add), used when folding bucket segments
template <class F>
CSH_HD_NOINLINE XYZZ<F> xyzz_mul_small(XYZZ<F> p, uint32_t k) {
  XYZZ<F> r = XYZZ<F>::inf();
  if (k == 0 || p.is_inf()) return r;
  int top = 31;
  while (!((k >> top) & 1)) --top;
  for (int b = top; b >= 0; --b) {
    r = xyzz_dbl(r);
    if ((k >> b) & 1) xyzz_add(r, p);
  }
  return r;
}

// ---- conversions (host side: need one inversion) -------------------------------------------------
template <class F>
CSH_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
  if (p.is_inf()) return Affine<F>::inf();
  F izzz = F::inv(p.zzz);           // 1/z^3
  F iz = F::mul(p.zz, izzz);        // z^2/z^3 = 1/z
  F izz = F::sqr(iz);               // 1/z^2
  return {F::mul(p.x, izz), F::mul(p.y, izzz)};
}

template <class F>
CSH_HD Jac<F> affine_to_jac(const Affine<F>& p) {
  if (p.is_inf()) return Jac<F>::inf();
  return {p.x, p.y, F::one()};
}

template <class F>
CSH_HD Affine<F> jac_to_affine(const Jac<F>& p) {
  if (p.is_inf()) return Affine<F>::inf();
  F iz = F::inv(p.z);
  F iz2 = F::sqr(iz);
  return {F::mul(p.x, iz2), F::mul(p.y, F::mul(iz2, iz))};
}

template <class F>
CSH_HD XYZZ<F> jac_to_xyzz(const Jac<F>& p) {
  if (p.is_inf()) return XYZZ<F>::inf();
  F zz = F::sqr(p.z);
  return {p.x, p.y, zz, F::mul(zz, p.z)};
}

template <class F>
CSH_HD Affine<F> affine_neg(const Affine<F>& p) {
  return {p.x, F::neg(p.y)};
}

// y^2 == x^3 + b
template <class F>
CSH_HD bool affine_on_curve(const Affine<F>& p, const F& b) {
  if (p.is_inf()) return true;
  return F::sqr(p.y) == F::add(F::mul(F::sqr(p.x), p.x), b);
}

}  // namespace csh
