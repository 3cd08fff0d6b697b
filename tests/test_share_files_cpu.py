"""Witness-share files either side of the prover: what `co-circom split-witness` writes and `generate-proof`
reads (co-circom.rs:700-735, 1014-1035) -- bincode 1.3 over the reference's `ark_se` byte strings. The reference commits
no `.shared` fixture, so the byte layout is pinned by two independent restatements (host/sharefile.hpp and
oracle/arkfmt.py) agreeing byte for byte, by the shares reconstructing the reference's committed witness.wtns, and (GPU
suite) by proofs made from the files equalling the plain proof. Host-only code path: runs without a GPU."""
import os
import struct

import pytest

from cosnarks_amd import groth16 as g
from cosnarks_amd.bindings import CoSnarksHipError
from oracle import arkfmt, mpc, zkey as oz
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CIRCUITS = [("bn254", "multiplier2"), ("bn254", "poseidon"), ("bls12_381", "multiplier2"), ("bls12_381", "poseidon")]


def _load(curve, circ):
    d = os.path.join(GOLD, "Groth16", curve, circ)
    rd = lambda f: open(os.path.join(d, f), "rb").read()
    zk, wt = rd("circuit.zkey"), rd("witness.wtns")
    return zk, wt, oz.parse_zkey(zk).n_public + 1, oz.parse_wtns(wt)


@pytest.mark.parametrize("curve,circ", CIRCUITS)
@pytest.mark.parametrize("compression", [0, 1])
def test_rep3_split_witness_files(curve, circ, compression):
    _, wt, npub, w = _load(curve, circ)
    F = H.FR[curve]
    files = g.split_witness(H.CURVE_IDS[curve], "rep3", wt, npub, seed=7, compression=compression)
    assert len(files) == 3
    parsed = [arkfmt.parse_rep3_share_file(f) for f in files]
    for f, (pub, kind, shares) in zip(files, parsed):
        assert pub == w[:npub]                                              # lib.rs:285, 320-322
        assert kind == (arkfmt.REP3_REPLICATED if compression == 0 else arkfmt.REP3_ADDITIVE)
        assert len(shares) == len(w) - npub
        assert arkfmt.ser_rep3_share_file(pub, kind, shares) == f           # the oracle's writer gives the same bytes
        back, variant, n_pub, n_wit = g.share_file_roundtrip(H.CURVE_IDS[curve], "rep3", f)
        assert back == f and variant == kind and n_pub == npub and n_wit == len(w) - npub
    if compression == 0:
        a = [[s[0] for s in p[2]] for p in parsed]
        b = [[s[1] for s in p[2]] for p in parsed]
        for i in range(3):
            assert b[i] == a[(i + 2) % 3]                                   # replicated: my b is the previous party's a
    else:
        a = [p[2] for p in parsed]
    assert [(x + y + z) % F.p for x, y, z in zip(*a)] == w[npub:]           # rep3.rs:281-292
    if len(w) - npub > 1:
        assert a[0] != w[npub:] and len(set(a[0])) > 1                      # actually masked


@pytest.mark.parametrize("curve,circ", CIRCUITS[:1] + CIRCUITS[3:])
@pytest.mark.parametrize("t,n", [(1, 3), (2, 5)])
def test_shamir_split_witness_files(curve, circ, t, n):
    _, wt, npub, w = _load(curve, circ)
    F = H.FR[curve]
    files = g.split_witness(H.CURVE_IDS[curve], "shamir", wt, npub, seed=11, threshold=t, num_parties=n)
    assert len(files) == n
    parsed = [arkfmt.parse_shamir_share_file(f) for f in files]
    for f, (pub, shares) in zip(files, parsed):
        assert pub == w[:npub] and len(shares) == len(w) - npub
        assert arkfmt.ser_shamir_share_file(pub, shares) == f
        back, _, n_pub, n_wit = g.share_file_roundtrip(H.CURVE_IDS[curve], "shamir", f)
        assert back == f and n_pub == npub and n_wit == len(w) - npub
    # any t+1 parties reconstruct (shamir.rs:442-491); t parties' points do not determine the secret
    for ids in ([0, 1, 2][: t + 1], list(range(n))[-(t + 1):]):
        lag = mpc.lagrange_from_coeff(F, [i + 1 for i in ids])
        rec = [mpc.shamir_reconstruct(F, [parsed[i][1][k] for i in ids], lag) for k in range(len(w) - npub)]
        assert rec == w[npub:]


def test_share_file_reader_rejects_malformed_input():
    _, wt, npub, w = _load("bn254", "multiplier2")
    good = g.split_witness(0, "rep3", wt, npub, seed=3)[0]
    rt = lambda d, proto="rep3": g.share_file_roundtrip(0, proto, d)
    assert rt(good)[0] == good
    for bad, why in [
        (good[:-1], "truncated"),
        (good + b"\0", "trailing"),
        (good[:8] + struct.pack("<Q", 1 << 40) + good[16:], "Vec length"),
        (struct.pack("<Q", 1 << 50) + good[8:], "byte-string length"),
    ]:
        with pytest.raises(CoSnarksHipError):
            rt(bad)
    # seeded variants (SeededReplicated = 1, SeededAdditive = 3) carry RNG seeds, not shares: refused with a clear message
    pub_len = 8 + struct.unpack_from("<Q", good, 0)[0]
    for variant in (1, 3, 9):
        with pytest.raises(CoSnarksHipError, match="variant"):
            rt(good[:pub_len] + struct.pack("<I", variant) + good[pub_len + 4:])
    # a non-canonical field element (>= r) is refused as ark-serialize does
    F = H.FR["bn254"]
    blob = bytearray(good)
    blob[16:48] = (F.p).to_bytes(32, "little")
    with pytest.raises(CoSnarksHipError, match="canonical"):
        rt(bytes(blob))
    with pytest.raises(CoSnarksHipError):
        g.split_witness(0, "rep3", wt, len(w) + 1, seed=1)
    with pytest.raises(CoSnarksHipError):
        g.split_witness(0, "rep3", wt, npub, seed=1, compression=2)       # seeded compression levels are not produced
    with pytest.raises(CoSnarksHipError):
        g.split_witness(0, "shamir", wt, npub, seed=1, threshold=2, num_parties=4)


def test_empty_private_witness_share_files():
    """A witness with nothing private (num_inputs == len) gives empty share vectors: 8-byte Vec headers only."""
    _, wt, npub, w = _load("bn254", "multiplier2")
    files = g.split_witness(0, "rep3", wt, len(w), seed=5)
    pub, kind, shares = arkfmt.parse_rep3_share_file(files[0])
    assert pub == w and shares == [] and kind == arkfmt.REP3_REPLICATED
    assert len(files[0]) == 8 + 8 + 32 * len(w) + 4 + 8 + 8
    assert g.share_file_roundtrip(0, "rep3", files[0])[0] == files[0]
