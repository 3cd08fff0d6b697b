// Host-only Montgomery field on 64-bit limbs (unsigned __int128 CIOS) for the sequential tail of an MSM: the Horner
// fold of the W window sums (W*c doublings) and the final inversion. Same modulus, same R = 2^(32*N) = 2^(64*N/2) and
// the same little-endian bytes as Fp<P> (field.hpp), so device results are reinterpreted in place; a 64-bit limb product
// replaces four 32-bit ones on the CPU (the fold drops from ~270 us to ~60 us, 10% of a 2^20 MSM).
#pragma once
#include <stdint.h>
#include <string.h>

#include "field.hpp"

namespace csh {

template <class P>
struct Fp64 {
  static_assert(P::N % 2 == 0, "limb count must be even");
  static constexpr int N = P::N / 2;
  using Params = P;
  uint64_t l[N];

  static uint64_t word(const uint32_t* w, int i) { return (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32); }
  static uint64_t inv64() {  // -p^-1 mod 2^64 by Newton iteration on the low modulus word
    static const uint64_t v = [] {
      const uint64_t p0 = word(P::MOD, 0);
      uint64_t y = p0;  // p0 * p0 = 1 mod 8
      for (int i = 0; i < 6; ++i) y *= 2 - p0 * y;
      return (uint64_t)0 - y;
    }();
    return v;
  }
  static Fp64 zero() {
    Fp64 r;
    for (int i = 0; i < N; ++i) r.l[i] = 0;
    return r;
  }
  static Fp64 one() {
    Fp64 r;
    for (int i = 0; i < N; ++i) r.l[i] = word(P::R1, i);
    return r;
  }
  bool is_zero() const {
    uint64_t a = 0;
    for (int i = 0; i < N; ++i) a |= l[i];
    return a == 0;
  }
  bool operator==(const Fp64& b) const { return memcmp(l, b.l, sizeof l) == 0; }
  bool operator!=(const Fp64& b) const { return !(*this == b); }

  static bool geq_mod(const uint64_t* a) {
    for (int i = N - 1; i >= 0; --i) {
      const uint64_t m = word(P::MOD, i);
      if (a[i] > m) return true;
      if (a[i] < m) return false;
    }
    return true;
  }
  static void sub_mod(uint64_t* a) {
    unsigned __int128 br = 0;
    for (int i = 0; i < N; ++i) {
      const unsigned __int128 d = (unsigned __int128)a[i] - word(P::MOD, i) - (uint64_t)br;
      a[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
  static Fp64 add(const Fp64& a, const Fp64& b) {  // all four moduli leave a spare top bit: no carry out
    Fp64 r;
    unsigned __int128 c = 0;
    for (int i = 0; i < N; ++i) {
      c += (unsigned __int128)a.l[i] + b.l[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  static Fp64 sub(const Fp64& a, const Fp64& b) {
    Fp64 r;
    uint64_t br = 0;
    for (int i = 0; i < N; ++i) {
      const unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - br;
      r.l[i] = (uint64_t)d;
      br = (uint64_t)(d >> 64) & 1;
    }
    if (br) {
      unsigned __int128 c = 0;
      for (int i = 0; i < N; ++i) {
        c += (unsigned __int128)r.l[i] + word(P::MOD, i);
        r.l[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  static Fp64 neg(const Fp64& a) { return a.is_zero() ? a : sub(zero(), a); }
  static Fp64 mul(const Fp64& a, const Fp64& b) {  // CIOS
    uint64_t t[N + 2];
    for (int i = 0; i < N + 2; ++i) t[i] = 0;
    const uint64_t inv = inv64();
    for (int i = 0; i < N; ++i) {
      unsigned __int128 c = 0;
      for (int j = 0; j < N; ++j) {
        c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[N];
      t[N] = (uint64_t)c;
      t[N + 1] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * inv;
      c = (unsigned __int128)m * word(P::MOD, 0) + t[0];
      c >>= 64;
      for (int j = 1; j < N; ++j) {
        c += (unsigned __int128)m * word(P::MOD, j) + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[N];
      t[N - 1] = (uint64_t)c;
      t[N] = t[N + 1] + (uint64_t)(c >> 64);
    }
    Fp64 r;
    for (int i = 0; i < N; ++i) r.l[i] = t[i];
    if (t[N] || geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  static Fp64 sqr(const Fp64& a) { return mul(a, a); }
  static Fp64 mul2(const Fp64& a) { return add(a, a); }
  static Fp64 mul3(const Fp64& a) { return add(add(a, a), a); }
  static Fp64 mul4(const Fp64& a) { return mul2(mul2(a)); }
  static Fp64 mul8(const Fp64& a) { return mul2(mul4(a)); }
  static Fp64 inv(const Fp64& a) {  // a^(p-2)
    Fp64 r = one();
    for (int i = P::N * 32 - 1; i >= 0; --i) {
      r = sqr(r);
      if ((P::PM2[i >> 5] >> (i & 31)) & 1) r = mul(r, a);
    }
    return r;
  }
};

// Fp<P> -> Fp64<P>, Fp2T<Fp<P>> -> Fp2T<Fp64<P>> (identical bytes)
template <class F>
struct Host64;
template <class P>
struct Host64<Fp<P>> {
  using type = Fp64<P>;
};
template <class P, int NR>
struct Host64<Fp2T<Fp<P>, NR>> {
  using type = Fp2T<Fp64<P>, NR>;
};

}  // namespace csh
