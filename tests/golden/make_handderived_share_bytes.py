#!/usr/bin/env python3
"""Byte strings of `.shared` witness-share files DERIVED BY HAND from the documented encodings, with no code of this repository's
serializers (neither host/sharefile.hpp nor oracle/arkfmt.py) in the loop: only struct.pack / int.to_bytes below. They pin the two
restatements against something other than each other. The reference commits no `.shared` fixture and cannot run here.

What the reference writes (co-circom/co-circom/src/bin/co-circom.rs:711, 733, 960): `bincode::serialize_into(file, &share)` with
bincode 1.3 (Cargo.toml:52), whose legacy free functions use the fixed-width little-endian configuration:
    * a struct is its fields in declaration order, nothing else;
    * an enum is its variant index as u32 LE, then the variant's fields in order;
    * `Serializer::serialize_bytes(b)` -- what `ark_se` calls (mpc-core/src/serde_compat.rs:7-15) -- is len(b) as u64 LE, then b;
    * usize is u64 LE; a `[u8; 32]` (the `Seed` of ChaCha12Rng) is its 32 bytes with no length prefix (serde serializes arrays as
      tuples); PhantomData is nothing.
`ark_se(x)` = x.serialize_with_mode(Compress::Yes): a Vec<T> is len as u64 LE then the items (ark-serialize's impl for slices); a prime
field element is its canonical value, 32 bytes LE for BN254 Fr; `Rep3PrimeFieldShare {a, b}` derives CanonicalSerialize: a then b
(mpc-core/src/protocols/rep3/arithmetic/types.rs:21-28); `ShamirPrimeFieldShare {a}` likewise one element.

Types (co-circom/co-circom-types/src/lib.rs:163-218; mpc-core/src/protocols/rep3.rs:135-165, 225-238):
    struct CompressedRep3SharedWitness { #[ark_se] public_inputs: Vec<F>, witness: Rep3ShareVecType<F> }
    enum   Rep3ShareVecType { 0 Replicated(#[ark_se] Vec<Rep3PrimeFieldShare<F>>), 1 SeededReplicated(ReplicatedSeedType {a, b: SeededType}),
                              2 Additive(#[ark_se] Vec<F>), 3 SeededAdditive(SeededType) }
    enum   SeededType { 0 Shares(#[ark_se] T), 1 Seed(U::Seed = [u8; 32], usize, PhantomData) }
    struct SharedWitness<P, S> { #[ark_se] public_inputs: Vec<P>, #[ark_se] witness: Vec<S> }        (the Shamir file)

Run from the repository root:  python tests/golden/make_handderived_share_bytes.py"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "share_files_handderived")


def u32(v):
    return struct.pack("<I", v)


def u64(v):
    return struct.pack("<Q", v)


def fr(v):                      # canonical little-endian, 32 bytes (BN254 Fr)
    return int(v).to_bytes(32, "little")


def ark_vec(items):             # ark-serialize Vec<T>: u64 length, then the items
    return u64(len(items)) + b"".join(items)


def serde_bytes(b):             # bincode serialize_bytes: u64 length, then the bytes
    return u64(len(b)) + b


R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
PUB = [1, 33]                                   # public inputs incl. the constant 1
SEED = bytes(range(100, 132))                   # a ChaCha12 seed
CASES = {}

# 1. party 0 of `split-witness` with Compression::SeededHalfShares: the explicit additive share vector (SeededAdditive(Shares(v)))
shares = [5, R - 1, 0x0123456789abcdef << 128]
CASES["rep3_seeded_additive_shares"] = {
    "protocol": "rep3", "variant": 3, "public_inputs": PUB, "witness": {"kind": "shares", "values": shares},
    "bytes": serde_bytes(ark_vec([fr(x) for x in PUB]))     # public_inputs
             + u32(3)                                        # Rep3ShareVecType::SeededAdditive
             + u32(0)                                        # SeededType::Shares
             + serde_bytes(ark_vec([fr(x) for x in shares])),
}
# 2. parties 1 / 2 of the same command: a seed and a length (SeededAdditive(Seed(seed, 3)))
CASES["rep3_seeded_additive_seed"] = {
    "protocol": "rep3", "variant": 3, "public_inputs": PUB, "witness": {"kind": "seed", "seed": list(SEED), "len": 3},
    "bytes": serde_bytes(ark_vec([fr(x) for x in PUB])) + u32(3) + u32(1) + SEED + u64(3),
}
# 3. Compression::None: replicated shares {a, b}
rep = [(7, 11), (R - 2, 13)]
CASES["rep3_replicated"] = {
    "protocol": "rep3", "variant": 0, "public_inputs": PUB, "witness": {"kind": "replicated", "values": [list(x) for x in rep]},
    "bytes": serde_bytes(ark_vec([fr(x) for x in PUB])) + u32(0) + serde_bytes(ark_vec([fr(a) + fr(b) for a, b in rep])),
}
# 4. Compression::SeededShares: {a: Shares(v), b: Seed(seed, 2)}
CASES["rep3_seeded_replicated"] = {
    "protocol": "rep3", "variant": 1, "public_inputs": PUB, "witness": {"kind": "seeded_replicated", "a": {"kind": "shares", "values": [9, 10]},
                                                                        "b": {"kind": "seed", "seed": list(SEED), "len": 2}},
    "bytes": serde_bytes(ark_vec([fr(x) for x in PUB])) + u32(1) + u32(0) + serde_bytes(ark_vec([fr(9), fr(10)])) + u32(1) + SEED + u64(2),
}
# 5. Compression::HalfShares: additive shares
CASES["rep3_additive"] = {
    "protocol": "rep3", "variant": 2, "public_inputs": PUB, "witness": {"kind": "additive", "values": [3, 4, 5]},
    "bytes": serde_bytes(ark_vec([fr(x) for x in PUB])) + u32(2) + serde_bytes(ark_vec([fr(3), fr(4), fr(5)])),
}
# 6. the Shamir file: SharedWitness<F, ShamirPrimeFieldShare<F>>
sh = [21, R - 5]
CASES["shamir"] = {
    "protocol": "shamir", "variant": 0, "public_inputs": PUB, "witness": {"kind": "shamir", "values": sh},
    "bytes": serde_bytes(ark_vec([fr(x) for x in PUB])) + serde_bytes(ark_vec([fr(x) for x in sh])),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    index = {}
    for name, c in CASES.items():
        open(os.path.join(OUT, name + ".shared"), "wb").write(c["bytes"])
        meta = {k: v for k, v in c.items() if k != "bytes"}
        meta["witness"] = json.loads(json.dumps(meta["witness"], default=str))
        meta["size"] = len(c["bytes"])
        index[name] = meta
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1, default=str)
    print(len(CASES), "files ->", OUT)


if __name__ == "__main__":
    main()
