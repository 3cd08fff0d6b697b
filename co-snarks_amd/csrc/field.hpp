// Montgomery prime-field arithmetic on 32-bit limbs, shared by gfx950 kernels and the host glue.
//
// Memory encoding = arkworks Fp<MontBackend<_, N64>, N64> (ark-ff 0.6.0, /root/reference Cargo.toml:46):
// N64 little-endian u64 limbs of x*R mod p, R = 2^(64*N64). A u64 LE limb is two u32 LE limbs, so the
// same bytes are N = 2*N64 u32 limbs of x*R mod p with R = 2^(32*N): no repacking at the boundary.
//
// gfx950 notes: the workhorse is v_mad_u64_u32 (32x32+64 -> 64). All four moduli leave >= 1 spare bit in
// the top limb, so CIOS needs no extra carry word and a single conditional subtraction suffices.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CSH_HD __host__ __device__ __forceinline__
// Out-of-line device functions (by value in, by value out): keeps the wide-field / rare-path code from
// being replicated into every kernel (a fully inlined BLS12-381 G2 bucket add is ~10^5 instructions).
#define CSH_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define CSH_HD inline
#define CSH_HD_NOINLINE inline
#endif

namespace csh {

#include "field_constants.inc"

// ---- raw multi-limb helpers -----------------------------------------------------------------
template <int N>
CSH_HD uint32_t limbs_add(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  unsigned c = 0;  // __builtin_addc lowers to v_add_co_u32 / v_addc_co_u32 chains on gfx950
#pragma unroll
  for (int i = 0; i < N; ++i) {
    unsigned co;
    r[i] = __builtin_addc(a[i], b[i], c, &co);
    c = co;
  }
  return c;
}

template <int N>
CSH_HD uint32_t limbs_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    unsigned co;
    r[i] = __builtin_subc(a[i], b[i], c, &co);
    c = co;
  }
  return c;  // borrow
}

template <int N>
CSH_HD bool limbs_geq(const uint32_t* a, const uint32_t* b) {
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    if (a[i] > b[i]) return true;
    if (a[i] < b[i]) return false;
  }
  return true;
}

template <class P>
struct Fp {
  using Params = P;
  static constexpr int N = P::N;
  uint32_t l[N];

  CSH_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = 0;
    return r;
  }
  CSH_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = P::R1[i];
    return r;
  }
  CSH_HD static Fp modulus() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = P::MOD[i];
    return r;
  }
  CSH_HD static Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = P::R2[i];
    return r;
  }
  CSH_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) o |= l[i];
    return o == 0;
  }
  CSH_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) o |= l[i] ^ b.l[i];
    return o == 0;
  }
  CSH_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // r = a - p if a >= p (inputs < 2p)
  CSH_HD void reduce_once() {
    uint32_t t[N];
    uint32_t m[N];
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = P::MOD[i];
    uint32_t borrow = limbs_sub<N>(t, l, m);
    if (!borrow) {
#pragma unroll
      for (int i = 0; i < N; ++i) l[i] = t[i];
    }
  }

  CSH_HD static Fp add(const Fp& a, const Fp& b) {
    Fp r;
    limbs_add<N>(r.l, a.l, b.l);  // < 2p < 2^(32N): no carry out (spare bit)
    r.reduce_once();
    return r;
  }
  CSH_HD static Fp dbl(const Fp& a) { return add(a, a); }
  CSH_HD static Fp sub(const Fp& a, const Fp& b) {
    Fp r;
    uint32_t borrow = limbs_sub<N>(r.l, a.l, b.l);
    if (borrow) {
      uint32_t m[N];
#pragma unroll
      for (int i = 0; i < N; ++i) m[i] = P::MOD[i];
      limbs_add<N>(r.l, r.l, m);
    }
    return r;
  }
  CSH_HD static Fp neg(const Fp& a) {
    if (a.is_zero()) return a;
    Fp r;
    uint32_t m[N];
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = P::MOD[i];
    limbs_sub<N>(r.l, m, a.l);
    return r;
  }

  // CIOS Montgomery product a*b*R^-1 mod p. 8-limb fields inline it; wider fields call it.
  CSH_HD static Fp mul(const Fp& a, const Fp& b) {
    if constexpr (N > 8) return mul_call(a, b);
    else return mul_impl(a, b);
  }
  CSH_HD_NOINLINE static Fp mul_call(Fp a, Fp b) { return mul_impl(a, b); }
  CSH_HD static Fp mul_impl(const Fp& a, const Fp& b) {
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t c = 0;
      const uint32_t bi = b.l[i];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        c = (uint64_t)a.l[j] * bi + t[j] + c;
        t[j] = (uint32_t)c;
        c >>= 32;
      }
      uint32_t tn = t[N] + (uint32_t)c;  // spare bit: no overflow
      const uint32_t m = t[0] * P::INV;
      c = (uint64_t)m * P::MOD[0] + t[0];
      c >>= 32;
#pragma unroll
      for (int j = 1; j < N; ++j) {
        c = (uint64_t)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (uint32_t)c;
        c >>= 32;
      }
      c += tn;
      t[N - 1] = (uint32_t)c;
      t[N] = (uint32_t)(c >> 32);
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = t[i];
    r.reduce_once();
    return r;
  }
  CSH_HD static Fp sqr(const Fp& a) { return mul(a, a); }

  // Montgomery -> canonical limbs (x*R * 1 * R^-1) and back.
  CSH_HD Fp from_mont() const {
    Fp o = zero();
    o.l[0] = 1;
    return mul(*this, o);
  }
  CSH_HD Fp to_mont() const { return mul(*this, r2()); }

  CSH_HD static Fp from_u64(uint64_t v) {
    Fp o = zero();
    o.l[0] = (uint32_t)v;
    o.l[1] = (uint32_t)(v >> 32);
    return o.to_mont();
  }

  // a^e, e given as little-endian 32-bit limbs (canonical integer)
  CSH_HD static Fp pow_limbs(const Fp& a, const uint32_t* e, int nlimbs) {
    Fp r = one();
    bool started = false;
    for (int i = nlimbs - 1; i >= 0; --i) {
      for (int b = 31; b >= 0; --b) {
        if (started) r = sqr(r);
        if ((e[i] >> b) & 1) {
          r = started ? mul(r, a) : a;
          started = true;
        }
      }
    }
    return started ? r : one();
  }
  CSH_HD static Fp pow_u64(const Fp& a, uint64_t e) {
    uint32_t l2[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
    return pow_limbs(a, l2, 2);
  }
  CSH_HD static Fp inv(const Fp& a) {
    uint32_t e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = P::PM2[i];
    return pow_limbs(a, e, N);
  }
  // small-integer multiples used by curve formulas
  CSH_HD static Fp mul2(const Fp& a) { return add(a, a); }
  CSH_HD static Fp mul3(const Fp& a) { return add(add(a, a), a); }
  CSH_HD static Fp mul4(const Fp& a) { return mul2(mul2(a)); }
  CSH_HD static Fp mul8(const Fp& a) { return mul2(mul4(a)); }
};

using Bn254Fq = Fp<Bn254FqParams>;
using Bn254Fr = Fp<Bn254FrParams>;
using Bls377Fr = Fp<Bls377FrParams>;
using Bls377Fq = Fp<Bls377FqParams>;
using Bls381Fq = Fp<Bls381FqParams>;
using Bls381Fr = Fp<Bls381FrParams>;

// ---- Fp2 = Fp[i]/(i^2 + NR): NR = 1 on BN254 / BLS12-381, NR = 5 on BLS12-377 (ark-bls12-377 Fq2Config::NONRESIDUE = -5) ----
template <class F, int NR = 1>
struct Fp2T {
  static constexpr int NONRESIDUE_NEG = NR;  // i^2 = -NR
  F c0, c1;
  CSH_HD static Fp2T zero() { return {F::zero(), F::zero()}; }
  CSH_HD static Fp2T one() { return {F::one(), F::zero()}; }
  CSH_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  CSH_HD bool operator==(const Fp2T& b) const { return c0 == b.c0 && c1 == b.c1; }
  CSH_HD bool operator!=(const Fp2T& b) const { return !(*this == b); }
  CSH_HD static Fp2T add(const Fp2T& a, const Fp2T& b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
  CSH_HD static Fp2T sub(const Fp2T& a, const Fp2T& b) { return {F::sub(a.c0, b.c0), F::sub(a.c1, b.c1)}; }
  CSH_HD static Fp2T neg(const Fp2T& a) { return {F::neg(a.c0), F::neg(a.c1)}; }
  CSH_HD static Fp2T dbl(const Fp2T& a) { return add(a, a); }
  // NR * v by additions (NR is 1 or 5)
  CSH_HD static F times_nr(const F& v) {
    if constexpr (NR == 1) return v;
    else {
      static_assert(NR == 5, "Fp2T: nonresidue -1 or -5");
      const F v2 = F::add(v, v);
      return F::add(F::add(v2, v2), v);
    }
  }
  // Karatsuba: 3 base multiplications (out of line: see CSH_HD_NOINLINE)
  CSH_HD static Fp2T mul(const Fp2T& a, const Fp2T& b) { return mul_call(a, b); }
  CSH_HD static Fp2T sqr(const Fp2T& a) { return sqr_call(a); }
  CSH_HD_NOINLINE static Fp2T mul_call(Fp2T a, Fp2T b) {
    F v0 = F::mul(a.c0, b.c0);
    F v1 = F::mul(a.c1, b.c1);
    F s = F::mul(F::add(a.c0, a.c1), F::add(b.c0, b.c1));
    return {F::sub(v0, times_nr(v1)), F::sub(F::sub(s, v0), v1)};
  }
  // (a+bi)^2 = (a+b)(a-b) + 2ab i for i^2 = -1; a^2 - NR b^2 + 2ab i in general
  CSH_HD_NOINLINE static Fp2T sqr_call(Fp2T a) {
    F ab = F::mul(a.c0, a.c1);
    if constexpr (NR == 1) {
      F t = F::mul(F::add(a.c0, a.c1), F::sub(a.c0, a.c1));
      return {t, F::add(ab, ab)};
    } else {
      return {F::sub(F::sqr(a.c0), times_nr(F::sqr(a.c1))), F::add(ab, ab)};
    }
  }
  CSH_HD static Fp2T inv(const Fp2T& a) {
    F n = F::inv(F::add(F::sqr(a.c0), times_nr(F::sqr(a.c1))));
    return {F::mul(a.c0, n), F::neg(F::mul(a.c1, n))};
  }
  CSH_HD static Fp2T mul2(const Fp2T& a) { return add(a, a); }
  CSH_HD static Fp2T mul3(const Fp2T& a) { return add(add(a, a), a); }
  CSH_HD static Fp2T mul4(const Fp2T& a) { return mul2(mul2(a)); }
  CSH_HD static Fp2T mul8(const Fp2T& a) { return mul2(mul4(a)); }
};

using Bn254Fq2 = Fp2T<Bn254Fq>;
using Bls381Fq2 = Fp2T<Bls381Fq>;
using Bls377Fq2 = Fp2T<Bls377Fq, 5>;

}  // namespace csh
