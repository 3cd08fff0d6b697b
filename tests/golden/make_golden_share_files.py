#!/usr/bin/env python3
"""Regression fixtures for the witness-share file formats (tests/golden/share_files/).

NOT reference output: the reference commits no `.shared` file and cannot be run here (no Rust toolchain). These files
are written by the ORACLE's restatement (oracle/arkfmt.py) from the reference's committed
test_vectors/Groth16/bn254/multiplier2/witness.wtns with fixed share randomness, so that a later change to either
restatement (oracle or host/sharefile.hpp) that alters the byte layout is caught. Layout sources:
co-circom-types/src/lib.rs:150-218, mpc-core/src/protocols/rep3.rs:135-165, mpc-core/src/serde_compat.rs:7-15.

Run from the repository root:  python tests/golden/make_golden_share_files.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import arkfmt, fields, zkey as oz  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "share_files")


def det(tag: str, i: int, p: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{tag}/{i}".encode()).digest() + hashlib.sha256(f"{tag}/{i}/b".encode()).digest(), "big") % p


def main():
    F = fields.BN254_FR
    d = os.path.join(HERE, "Groth16", "bn254", "multiplier2")
    w = oz.parse_wtns(open(os.path.join(d, "witness.wtns"), "rb").read())
    npub = oz.parse_zkey(open(os.path.join(d, "circuit.zkey"), "rb").read()).n_public + 1
    pub, priv = w[:npub], w[npub:]
    os.makedirs(OUT, exist_ok=True)
    files = {}
    # replicated (Compression::None)
    a = [det("a", i, F.p) for i in range(len(priv))]
    b = [det("b", i, F.p) for i in range(len(priv))]
    c = [(v - x - y) % F.p for v, x, y in zip(priv, a, b)]
    for p, (mine, prev) in enumerate(((a, c), (b, a), (c, b))):
        files[f"rep3_replicated.{p}.shared"] = arkfmt.ser_rep3_share_file(pub, arkfmt.REP3_REPLICATED, list(zip(mine, prev)))
        files[f"rep3_additive.{p}.shared"] = arkfmt.ser_rep3_share_file(pub, arkfmt.REP3_ADDITIVE, mine)
    # seeded half shares (Compression::SeededHalfShares, what `co-circom split-witness` writes)
    seed_b, seed_c = hashlib.sha256(b"seed_b").digest(), hashlib.sha256(b"seed_c").digest()
    sb, sc = ("seed", seed_b, len(priv)), ("seed", seed_c, len(priv))
    eb, ec = (arkfmt.expand_seeded(x, F.p, 254, 1 << 256) for x in (sb, sc))
    sa = ("shares", [(v - x - y) % F.p for v, x, y in zip(priv, eb, ec)])
    for p, s in enumerate((sa, sb, sc)):
        files[f"rep3_seeded_additive.{p}.shared"] = arkfmt.ser_rep3_share_file(pub, arkfmt.REP3_SEEDED_ADDITIVE, s)
    for p, s in enumerate(((sa, sc), (sb, sa), (sc, sb))):
        files[f"rep3_seeded_replicated.{p}.shared"] = arkfmt.ser_rep3_share_file(pub, arkfmt.REP3_SEEDED_REPLICATED, s)
    # Shamir, 3 parties, threshold 1: f(X) = secret + r X at X = 1, 2, 3
    r = [det("r", i, F.p) for i in range(len(priv))]
    for p in range(3):
        files[f"shamir_t1.{p}.shared"] = arkfmt.ser_shamir_share_file(pub, [(v + (p + 1) * x) % F.p for v, x in zip(priv, r)])
    for name, data in files.items():
        open(os.path.join(OUT, name), "wb").write(data)
    print(len(files), "files,", sum(len(v) for v in files.values()), "bytes")


if __name__ == "__main__":
    main()
