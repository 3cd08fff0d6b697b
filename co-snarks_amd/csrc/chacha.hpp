// ChaCha12 block function as rand_chacha::ChaCha12Rng lays it out (djb variant: 64-bit block counter in words
// 12-13, 64-bit stream id 0 in words 14-15, little-endian output). Shared by the on-device Rep3 mask generator and
// the host self-test. mpc-core: RngType = ChaCha12Rng (mpc-core/src/lib.rs:13); masks = rep3/rngs.rs:137-156.
#pragma once
#include "field.hpp"

namespace csh {

CSH_HD uint32_t chacha_rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define CSH_CHACHA_QR(a, b, c, d)                     \
  x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 16);  \
  x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 12);  \
  x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 8);   \
  x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 7);

// out[16] = keystream words of block `counter`
CSH_HD void chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
  uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                     (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = st[i];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    CSH_CHACHA_QR(0, 4, 8, 12) CSH_CHACHA_QR(1, 5, 9, 13) CSH_CHACHA_QR(2, 6, 10, 14) CSH_CHACHA_QR(3, 7, 11, 15)
    CSH_CHACHA_QR(0, 5, 10, 15) CSH_CHACHA_QR(1, 6, 11, 12) CSH_CHACHA_QR(2, 7, 8, 13) CSH_CHACHA_QR(3, 4, 9, 14)
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[i] = x[i] + st[i];
}

// F::from_be_bytes_mod_order over the 32-byte half `half` (0/1) of a keystream block: the bytes are the
// little-endian serialisation of words[8*half .. 8*half+8); read as a big-endian integer, limb i (LE) is the
// byte-swapped word 7-i. One Montgomery multiplication by R^2 reduces mod r and enters Montgomery form.
template <class Fr>
CSH_HD Fr chacha_half_to_field(const uint32_t words[16], int half) {
  Fr v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t w = words[8 * half + 7 - i];
    v.l[i] = (w >> 24) | ((w >> 8) & 0xff00u) | ((w << 8) & 0xff0000u) | (w << 24);
  }
  return v.to_mont();
}

// Mask element number `e` of the stream pair: from_be(a_e) - from_be(b_e), a/b = 32-byte chunks e of the two streams
template <class Fr>
CSH_HD Fr rep3_mask_element(const uint32_t key1[8], const uint32_t key2[8], uint64_t e1, uint64_t e2) {
  uint32_t w1[16], w2[16];
  chacha12_block(key1, e1 >> 1, w1);
  chacha12_block(key2, e2 >> 1, w2);
  return Fr::sub(chacha_half_to_field<Fr>(w1, (int)(e1 & 1)), chacha_half_to_field<Fr>(w2, (int)(e2 & 1)));
}

}  // namespace csh
