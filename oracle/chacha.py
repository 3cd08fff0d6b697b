"""ChaCha keystream as rand_chacha::ChaCha12Rng produces it (oracle; test infrastructure only).

mpc-core's `RngType` is `rand_chacha::ChaCha12Rng` (mpc-core/src/lib.rs:13); `Rep3Rand` draws its masks with
`fill_bytes` on two such generators seeded with 32-byte keys (rep3/rngs.rs:83-95, 137-156). rand_chacha (external
crate) uses the original djb layout: constants, 8 key words, a 64-bit block counter in words 12-13, a 64-bit stream id
(0) in words 14-15, little-endian word serialisation, blocks consumed in order. The core is pinned in
tests/test_oracle.py on the RFC 7539 ChaCha20 block vector (same layout when the nonce words are zero)."""
from __future__ import annotations

import struct

MASK = 0xFFFFFFFF


def _rotl(v, c):
    return ((v << c) & MASK) | (v >> (32 - c))


def _qr(x, a, b, c, d):
    x[a] = (x[a] + x[b]) & MASK; x[d] = _rotl(x[d] ^ x[a], 16)
    x[c] = (x[c] + x[d]) & MASK; x[b] = _rotl(x[b] ^ x[c], 12)
    x[a] = (x[a] + x[b]) & MASK; x[d] = _rotl(x[d] ^ x[a], 8)
    x[c] = (x[c] + x[d]) & MASK; x[b] = _rotl(x[b] ^ x[c], 7)


def block(key: bytes, counter: int, rounds: int = 12, nonce_words=(0, 0)) -> bytes:
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", key))
    st += [counter & MASK, (counter >> 32) & MASK, nonce_words[0], nonce_words[1]]
    x = list(st)
    for _ in range(rounds // 2):
        _qr(x, 0, 4, 8, 12); _qr(x, 1, 5, 9, 13); _qr(x, 2, 6, 10, 14); _qr(x, 3, 7, 11, 15)
        _qr(x, 0, 5, 10, 15); _qr(x, 1, 6, 11, 12); _qr(x, 2, 7, 8, 13); _qr(x, 3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & MASK for a, b in zip(x, st)])


def keystream(key: bytes, nbytes: int, start_byte: int = 0, rounds: int = 12) -> bytes:
    out = bytearray()
    blk = start_byte // 64
    skip = start_byte % 64
    while len(out) < nbytes + skip:
        out += block(key, blk, rounds)
        blk += 1
    return bytes(out[skip:skip + nbytes])
