"""Wire formats either side of the hot path pinned on byte strings derived BY HAND from the documented encodings:
(1) `.shared` witness-share files (tests/golden/make_handderived_share_bytes.py: bincode 1.3 fixed-width little-endian over the
reference's `ark_se` byte strings) through BOTH restatements, host/sharefile.hpp and oracle/arkfmt.py; (2) the payloads of
Rep3NetworkExt::{send_many, recv_many} (mpc-core/src/protocols/rep3/network.rs:103-109, 152-156) -- what a GPU party must put on the wire
to face two reference CPU parties. Host-only code paths: run without a GPU."""
import json
import os
import struct

import numpy as np
import pytest

from cosnarks_amd import groth16 as g
from cosnarks_amd.bindings import CoSnarksHipError
from oracle import arkfmt
from oracle import curves as cv
from oracle import fields as fl
from tests import helpers as H

HD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "share_files_handderived")
INDEX = json.load(open(os.path.join(HD, "index.json")))
R = fl.BN254_FR.p


def _seed(t):
    return ("seed", bytes(t["seed"]), t["len"]) if t["kind"] == "seed" else ("shares", t["values"])


@pytest.mark.parametrize("name", sorted(INDEX))
def test_hand_derived_share_files_parse_and_reserialize_in_both_restatements(name):
    meta = INDEX[name]
    data = open(os.path.join(HD, name + ".shared"), "rb").read()
    assert len(data) == meta["size"]
    w = meta["witness"]
    # oracle/arkfmt.py
    if meta["protocol"] == "shamir":
        pub, shares = arkfmt.parse_shamir_share_file(data)
        assert pub == meta["public_inputs"] and shares == [int(x) for x in w["values"]]
        assert arkfmt.ser_shamir_share_file(pub, shares) == data
        n_wit = len(shares)
    else:
        pub, kind, shares = arkfmt.parse_rep3_share_file(data)
        assert pub == meta["public_inputs"] and kind == meta["variant"]
        if w["kind"] == "replicated":
            assert shares == [tuple(int(v) for v in x) for x in w["values"]]
            n_wit = len(shares)
        elif w["kind"] == "additive":
            assert shares == [int(x) for x in w["values"]]
            n_wit = len(shares)
        elif w["kind"] == "seeded_replicated":
            assert (shares[0][0], [int(x) for x in shares[0][1]]) == ("shares", w["a"]["values"]) and shares[1] == _seed(w["b"])
            n_wit = len(w["a"]["values"])
        else:
            assert tuple(shares) == _seed(w) if w["kind"] == "seed" else (shares[0], [int(x) for x in shares[1]]) == ("shares", [int(x) for x in w["values"]])
            n_wit = w["len"] if w["kind"] == "seed" else len(w["values"])
        assert arkfmt.ser_rep3_share_file(pub, kind, shares) == data
    # host/sharefile.hpp (the product's host mirror): parse, report, write the same bytes back
    back, variant, n_pub, nw = g.share_file_roundtrip(0, meta["protocol"], data)
    assert back == data and n_pub == len(meta["public_inputs"]) and nw == n_wit
    if meta["protocol"] == "rep3":
        assert variant == meta["variant"]


def test_hand_derived_files_reject_damage():
    data = open(os.path.join(HD, "rep3_additive.shared"), "rb").read()
    for bad in (data[:-1], data + b"\0", data[:8] + b"\xff" + data[9:], data[:80] + struct.pack("<I", 9) + data[84:]):
        with pytest.raises(CoSnarksHipError):
            g.share_file_roundtrip(0, "rep3", bad)
    # a non-canonical field element (>= r) inside the share vector: ark's Fp::deserialize fails, so must the mirror
    i = data.rindex((3).to_bytes(32, "little"))
    with pytest.raises(CoSnarksHipError, match="canonical"):
        g.share_file_roundtrip(0, "rep3", data[:i] + R.to_bytes(32, "little") + data[i + 32:])


def test_send_many_recv_many_payloads_byte_for_byte():
    F = fl.BN254_FR
    vals = [1, 2, R - 1]
    msg = g.rep3_send_many(0, H.pack(F, vals))
    # [T]::serialize_uncompressed: the count as u64 LE, then 32 canonical little-endian bytes per element
    assert msg == struct.pack("<Q", 3) + b"".join(v.to_bytes(32, "little") for v in vals)
    assert H.unpack(F, g.rep3_recv_many(0, msg)) == vals
    assert g.rep3_send_many(0, np.zeros(0, dtype=np.uint64)) == struct.pack("<Q", 0)          # send_many(&[]) is a valid message
    assert len(g.rep3_recv_many(0, struct.pack("<Q", 0))) == 0
    one = g.rep3_send_many(0, H.pack(F, [7]))                                                   # send_to / send_next: a one-item slice (:96-99)
    assert one == struct.pack("<Q", 1) + (7).to_bytes(32, "little")
    with pytest.raises(CoSnarksHipError, match="canonical"):                                     # Fp::from_bigint fails for values >= r
        g.rep3_recv_many(0, struct.pack("<Q", 1) + R.to_bytes(32, "little"))
    with pytest.raises(CoSnarksHipError):                                                         # count exceeds the message
        g.rep3_recv_many(0, struct.pack("<Q", 2) + (7).to_bytes(32, "little"))


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_send_many_points_carry_the_swflags(curve):
    """Curve points travel as uncompressed affine coordinates with SWFlags in the two top bits of y's last byte (bit 7: y is the larger
    of {y, -y}, bit 6: infinity with all-zero coordinates) -- open_point / broadcast of half shares (rep3/pointshare.rs:152-155)."""
    G = cv.CURVES[curve][0]
    q = G.F.p
    nb = 32 if curve == "bn254" else 48
    gx, gy = G.gen
    neg = (gx, q - gy)
    msg = g.rep3_send_many(H.CURVE_IDS[curve], cv.pack_points(G, [G.gen, neg, None]), points=True)
    assert msg[:8] == struct.pack("<Q", 3) and len(msg) == 8 + 3 * 2 * nb
    body = msg[8:]
    enc = lambda x, y, flag: x.to_bytes(nb, "little") + (y | (flag << (8 * nb - 8))).to_bytes(nb, "little")
    larger = lambda y: 0x80 if y > q - y else 0
    assert body[:2 * nb] == enc(gx, gy, larger(gy))
    assert body[2 * nb:4 * nb] == enc(gx, q - gy, larger(q - gy))
    assert body[4 * nb:] == bytes(2 * nb - 1) + b"\x40"
    assert larger(gy) != larger(q - gy)
    back = cv.unpack_points(G, g.rep3_recv_many(H.CURVE_IDS[curve], msg, points=True))
    assert back == [G.gen, neg, None]
    # deserialize_uncompressed_unchecked does not consult the sign bit of an uncompressed point: a flipped bit is accepted
    flipped = bytearray(msg)
    flipped[8 + 2 * nb - 1] ^= 0x80
    assert cv.unpack_points(G, g.rep3_recv_many(H.CURVE_IDS[curve], bytes(flipped), points=True))[0] == G.gen


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_mirror_rep3_randomness_parallel_draw_and_fresh_seeds(curve):
    """The mirror's Rep3Rand (host/mpc.hpp) on its own, no device: (1) masking_field_elements_vec -- drawn in parallel over the host's
    cores from the two counter-mode keystreams -- equals the serial restatement of rngs.rs:137-156 over the oracle's ChaCha12 keystream,
    for a first vector, and for a second one drawn after a random_seeds() call (the generators continue where a serial draw leaves
    them: 32 keystream WORDS per seed, rand 0.8's `gen::<[u8; 32]>()`); (2) the three parties' masks sum to zero element by element;
    (3) random_seeds (rngs.rs:233): the seed party p draws from its own generator is the one party p + 1 draws from its second -- what
    keeps the opt-in seeded masks of the Rust shim correlated."""
    import ctypes as C

    import numpy as np
    from cosnarks_amd import groth16 as g
    from oracle import chacha, mpc
    F = H.FR[curve]
    n, n2 = 20000, 777                                 # spans several parallel chunks, odd tail
    keys = bytes((37 * i + 11) & 0xFF for i in range(96))
    masks = np.zeros(3 * (n + n2) * 4, dtype=np.uint64)
    seeds = (C.c_uint8 * 192)()
    rc = g.glib().cog16_rep3_rand_selftest(H.CURVE_IDS[curve], keys, C.c_size_t(n), C.c_size_t(n2), masks.ctypes.data_as(C.c_void_p), seeds)
    assert rc == 0, g.glib().cog16_last_error()
    got = [H.unpack(F, masks[4 * (n + n2) * p:4 * (n + n2) * (p + 1)]) for p in range(3)]
    seeds = bytes(seeds)
    for p in range(3):
        k1, k2 = keys[32 * p:32 * p + 32], keys[32 * ((p + 2) % 3):32 * ((p + 2) % 3) + 32]
        s1 = chacha.keystream(k1, 32 * (n + n2) + 128)
        s2 = chacha.keystream(k2, 32 * (n + n2) + 128)
        assert got[p][:n] == mpc.masks_from_streams(F, s1, s2, n), p
        # random_seeds: byte i of a seed = the low byte of the i-th next keystream word
        want1 = bytes(s1[32 * n + 4 * i] for i in range(32))
        want2 = bytes(s2[32 * n + 4 * i] for i in range(32))
        assert seeds[64 * p:64 * p + 32] == want1 and seeds[64 * p + 32:64 * p + 64] == want2, p
        assert got[p][n:] == mpc.masks_from_streams(F, s1[32 * n + 128:], s2[32 * n + 128:], n2), p
    assert all((a + b + c) % F.p == 0 for a, b, c in zip(*got))
    for p in range(3):
        assert seeds[64 * p:64 * p + 32] == seeds[64 * ((p + 1) % 3) + 32:64 * ((p + 1) % 3) + 64]

