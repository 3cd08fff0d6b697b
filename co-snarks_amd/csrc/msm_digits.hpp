// Signed-digit window recoding shared by the MSM kernels and the host self-test.
#pragma once
#include "field.hpp"

namespace csh {

// windows needed for scalars < 2^bits with signed c-bit digits (one spare bit absorbs the final carry)
CSH_HD int windows_for(int bits, int c) { return (bits + 1 + c - 1) / c; }

// signed-digit recoding: calls fn(w, bucket in 1..NB, negative) for every non-zero digit
template <int N, class Fn>
CSH_HD void for_each_digit(const uint32_t* s, int c, int W, Fn fn) {
  uint32_t carry = 0;
  const uint32_t mask = (1u << c) - 1;
  const uint32_t half = 1u << (c - 1);
  for (int w = 0; w < W; ++w) {
    const int bit = w * c;
    const int limb = bit >> 5, off = bit & 31;
    uint64_t two = 0;
    if (limb < N) two = s[limb];
    if (limb + 1 < N) two |= (uint64_t)s[limb + 1] << 32;
    uint32_t v = ((uint32_t)(two >> off) & mask) + carry;
    if (v > half) {
      carry = 1;
      uint32_t mag = (1u << c) - v;  // 0 .. half-1
      if (mag) fn(w, mag, 1u);
    } else {
      carry = 0;
      if (v) fn(w, v, 0u);
    }
  }
}


}  // namespace csh
