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


def _expand(F, x):
    return arkfmt.expand_seeded(x, F.p, F.p.bit_length(), 1 << 256)


@pytest.mark.parametrize("curve,circ", CIRCUITS)
@pytest.mark.parametrize("compression", [0, 1, 2, 3])
def test_rep3_split_witness_files(curve, circ, compression):
    """All four Compression levels of co-circom-types/src/lib.rs:150-161; 3 (SeededHalfShares) is what the reference's
    `split-witness` command writes (co-circom.rs:693-697)."""
    _, wt, npub, w = _load(curve, circ)
    F = H.FR[curve]
    files = g.split_witness(H.CURVE_IDS[curve], "rep3", wt, npub, seed=7, compression=compression)
    assert len(files) == 3
    parsed = [arkfmt.parse_rep3_share_file(f) for f in files]
    want_kind = [arkfmt.REP3_REPLICATED, arkfmt.REP3_ADDITIVE, arkfmt.REP3_SEEDED_REPLICATED, arkfmt.REP3_SEEDED_ADDITIVE][compression]
    nw = len(w) - npub
    for f, (pub, kind, shares) in zip(files, parsed):
        assert pub == w[:npub]                                              # lib.rs:285, 320-322
        assert kind == want_kind
        assert arkfmt.ser_rep3_share_file(pub, kind, shares) == f           # the oracle's writer gives the same bytes
        back, variant, n_pub, n_wit = g.share_file_roundtrip(H.CURVE_IDS[curve], "rep3", f)
        assert back == f and variant == kind and n_pub == npub and n_wit == nw
    if compression == 0:
        a = [[s[0] for s in p[2]] for p in parsed]
        b = [[s[1] for s in p[2]] for p in parsed]
    elif compression == 1:
        a, b = [p[2] for p in parsed], None
    elif compression == 2:                                                   # rep3.rs:455-497: {a, c}, {b, a}, {c, b}
        assert [p[2][0][0] for p in parsed] == ["shares", "seed", "seed"] and [p[2][1][0] for p in parsed] == ["seed", "shares", "seed"]
        a = [_expand(F, p[2][0]) for p in parsed]
        b = [_expand(F, p[2][1]) for p in parsed]
    else:                                                                    # rep3.rs:500-533: [Shares(a), Seed(b), Seed(c)]
        assert [p[2][0] for p in parsed] == ["shares", "seed", "seed"]
        assert len(files[1]) == len(files[2]) == 8 + 8 + 32 * npub + 4 + 4 + 32 + 8
        a, b = [_expand(F, p[2]) for p in parsed], None
    assert all(len(x) == nw for x in a)
    if b is not None:
        for i in range(3):
            assert b[i] == a[(i + 2) % 3]                                   # replicated: my b is the previous party's a
    assert [(x + y + z) % F.p for x, y, z in zip(*a)] == w[npub:]           # rep3.rs:281-292
    if nw > 1:
        assert a[0] != w[npub:] and len(set(a[0])) > 1                      # actually masked


def test_seed_expansion_is_the_restated_field_sampler():
    """F::rand over ChaCha12Rng (ark-ff 0.6.0 `Distribution<Fp> for Standard`, restated): 32 keystream bytes per draw as
    little-endian limbs, top bits masked, rejected while >= p, taken as the Montgomery representation. Checked here
    independently of the oracle helper, straight from the ChaCha12 keystream (itself pinned on RFC 7539 in test_oracle)."""
    from oracle import chacha
    _, wt, npub, w = _load("bls12_381", "multiplier2")
    F = H.FR["bls12_381"]
    files = g.split_witness(1, "rep3", wt, npub, seed=99, compression=3)
    _, _, (tag, seed, n) = arkfmt.parse_rep3_share_file(files[1])
    assert tag == "seed" and n == len(w) - npub
    ks = chacha.keystream(seed, 32 * 64)
    draws, pos = [], 0
    while len(draws) < n:
        v = int.from_bytes(ks[pos:pos + 32], "little") & ((1 << 255) - 1)
        pos += 32
        if v < F.p:
            draws.append(v * pow(1 << 256, -1, F.p) % F.p)
    assert draws == _expand(F, ("seed", seed, n))
    # the explicit party's share is witness - b - c with those draws
    _, _, (_, a) = arkfmt.parse_rep3_share_file(files[0])
    c = _expand(F, arkfmt.parse_rep3_share_file(files[2])[2])
    assert [(x + y + z) % F.p for x, y, z in zip(a, draws, c)] == w[npub:]


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
    # an unknown Rep3ShareVecType / SeededType variant index, or a seeded body cut short, is refused
    pub_len = 8 + struct.unpack_from("<Q", good, 0)[0]
    with pytest.raises(CoSnarksHipError, match="variant"):
        rt(good[:pub_len] + struct.pack("<I", 9) + good[pub_len + 4:])
    seeded = g.split_witness(0, "rep3", wt, npub, seed=3, compression=3)[1]
    assert rt(seeded)[0] == seeded
    with pytest.raises(CoSnarksHipError, match="variant"):
        rt(seeded[:pub_len + 4] + struct.pack("<I", 2) + seeded[pub_len + 8:])
    for cut in (1, 8, 20, 41):
        with pytest.raises(CoSnarksHipError):
            rt(seeded[:-cut])
    with pytest.raises(CoSnarksHipError, match="implausible"):
        rt(seeded[:-8] + struct.pack("<Q", 1 << 40))
    # a non-canonical field element (>= r) is refused as ark-serialize does
    F = H.FR["bn254"]
    blob = bytearray(good)
    blob[16:48] = (F.p).to_bytes(32, "little")
    with pytest.raises(CoSnarksHipError, match="canonical"):
        rt(bytes(blob))
    with pytest.raises(CoSnarksHipError):
        g.split_witness(0, "rep3", wt, len(w) + 1, seed=1)
    with pytest.raises(CoSnarksHipError):
        g.split_witness(0, "rep3", wt, npub, seed=1, compression=4)
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


@pytest.mark.parametrize("curve,circ", CIRCUITS)
def test_public_input_file_matches_the_reference_fixture(curve, circ):
    """generate-proof writes the share file's public inputs (constant 1 skipped) as decimal strings (co-circom.rs:1120-1140):
    equal to the reference's committed public.json for every circuit, from a Rep3 and from a Shamir share file."""
    import json
    _, wt, npub, w = _load(curve, circ)
    want = json.load(open(os.path.join(GOLD, "Groth16", curve, circ, "public.json")))
    cid = H.CURVE_IDS[curve]
    rep = g.split_witness(cid, "rep3", wt, npub, seed=1, compression=3)
    sham = g.split_witness(cid, "shamir", wt, npub, seed=1, threshold=1, num_parties=3)
    for proto, f in (("rep3", rep[0]), ("rep3", rep[2]), ("shamir", sham[1])):
        txt = g.public_inputs_json(cid, proto, f)
        assert json.loads(txt) == want
        assert " " not in txt and "\n" not in txt                      # serde_json::to_writer is compact


def test_public_input_file_decimal_edge_values():
    """Zero prints as "0" (co-circom.rs:1126-1130), limb-boundary values and r - 1 print exactly."""
    import json
    F = H.FR["bn254"]
    vals = [1, 0, 5, 10**9, 10**9 - 1, 10**18 + 7, 2**64, 2**128 - 1, F.p - 1]
    txt = g.public_inputs_json(0, "shamir", arkfmt.ser_shamir_share_file(vals, []))
    assert json.loads(txt) == [str(v) for v in vals[1:]]
    assert g.public_inputs_json(0, "rep3", arkfmt.ser_rep3_share_file([1], arkfmt.REP3_ADDITIVE, [])) == "[]"


SHARE_FIXTURES = os.path.join(GOLD, "share_files")


@pytest.mark.parametrize("family,proto,kind", [("rep3_replicated", "rep3", 0), ("rep3_seeded_replicated", "rep3", 1), ("rep3_additive", "rep3", 2),
                                               ("rep3_seeded_additive", "rep3", 3), ("shamir_t1", "shamir", 0)])
def test_committed_share_file_fixtures(family, proto, kind):
    """tests/golden/share_files (written by the oracle's restatement with fixed randomness,
    tests/golden/make_golden_share_files.py -- regression fixtures, not reference output): the host mirror parses every
    file, re-serializes it byte for byte, and the three parties' shares reconstruct the reference's committed witness."""
    _, _, npub, w = _load("bn254", "multiplier2")
    F = H.FR["bn254"]
    files = [open(os.path.join(SHARE_FIXTURES, f"{family}.{p}.shared"), "rb").read() for p in range(3)]
    for f in files:
        back, variant, n_pub, n_wit = g.share_file_roundtrip(0, proto, f)
        assert back == f and variant == kind and n_pub == npub and n_wit == len(w) - npub
    if proto == "shamir":
        parsed = [arkfmt.parse_shamir_share_file(f) for f in files]
        lag = mpc.lagrange_from_coeff(F, [1, 3])
        assert [mpc.shamir_reconstruct(F, [parsed[0][1][k], parsed[2][1][k]], lag) for k in range(len(w) - npub)] == w[npub:]
        return
    parsed = [arkfmt.parse_rep3_share_file(f) for f in files]
    assert all(p[0] == w[:npub] and p[1] == kind for p in parsed)
    if kind == 0:
        a = [[s[0] for s in p[2]] for p in parsed]
    elif kind == 1:
        a = [_expand(F, p[2][0]) for p in parsed]
    elif kind == 2:
        a = [p[2] for p in parsed]
    else:
        a = [_expand(F, p[2]) for p in parsed]
    assert [(x + y + z) % F.p for x, y, z in zip(*a)] == w[npub:]


def test_share_file_fixtures_are_reproducible(tmp_path):
    """The committed fixtures are exactly what the generating script writes today."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    before = {f: open(os.path.join(SHARE_FIXTURES, f), "rb").read() for f in sorted(os.listdir(SHARE_FIXTURES))}
    assert len(before) == 15
    src = open(os.path.join(GOLD, "make_golden_share_files.py")).read().replace('OUT = os.path.join(HERE, "share_files")', f'OUT = {str(tmp_path)!r}')
    script = tmp_path / "gen.py"
    script.write_text(src.replace("HERE = os.path.dirname(os.path.abspath(__file__))", f"HERE = {GOLD!r}").replace(
        "sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))", f"sys.path.insert(0, {root!r})"))
    subprocess.run([sys.executable, str(script)], check=True, capture_output=True)
    for f, data in before.items():
        assert open(tmp_path / f, "rb").read() == data, f
