#!/usr/bin/env python3
"""Builds tests/golden/Groth16/bls12_377/penumbra_output/ from the reference's own LibSnarkReduction fixture
(/root/reference/test_vectors/Groth16/bls12_377/penumbra_output: a.bin, b.bin, c.bin, witness.wtns, circuit.vk -- data files
of the reference's test co-circom/co-groth16/src/lib.rs:231-300; its circuit.pk is absent, so no proof can be formed).

Committed: the data files gzip-compressed (inputs) and expected.json (outputs of the oracle restatement, itself checked here
against the QAP identity H(t) Z(t) = A(t) B(t) - C(t)): sha256 of the little-endian canonical h vector, H evaluated at a
fixed point, the first coefficients. Run in the build container only (needs /root/reference)."""
import gzip
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import arkfmt, fields as fl, groth16 as g16, ntt  # noqa: E402

SRC = "/root/reference/test_vectors/Groth16/bls12_377/penumbra_output"
DST = os.path.join(HERE, "Groth16", "bls12_377", "penumbra_output")
GENERATOR = 22  # ark_bls12_377::Fr::GENERATOR
TAU = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF


def main():
    os.makedirs(DST, exist_ok=True)
    raw = {}
    for name in ["a.bin", "b.bin", "c.bin", "witness.wtns", "circuit.vk"]:
        raw[name] = open(os.path.join(SRC, name), "rb").read()
        with gzip.GzipFile(os.path.join(DST, name + ".gz"), "wb", mtime=0) as f:
            f.write(raw[name])
    F = fl.BLS377_FR
    A, B, Cm = (arkfmt.parse_matrix(raw[n]) for n in ["a.bin", "b.bin", "c.bin"])
    prime, w = arkfmt.parse_wtns_positional(raw["witness.wtns"])
    assert prime == F.p
    ni = arkfmt.vk_num_instance_variables(raw["circuit.vk"], 96, 192)
    pub, wit = w[:ni], w[ni:]
    h = g16.witness_map_libsnark(F, GENERATOR, A, B, Cm, len(A), g16.PlainDriver(F), pub, wit)
    tau = TAU % F.p
    assert g16.libsnark_identity_holds(F, GENERATOR, A, B, Cm, len(A), pub, wit, h, tau)
    exp = {
        "source": "reference test_vectors/Groth16/bls12_377/penumbra_output (co-groth16/src/lib.rs:231-300)",
        "generator": GENERATOR, "num_constraints": len(A), "num_instance_variables": ni, "num_variables": len(w),
        "domain_size": len(h), "tau": str(tau), "H_at_tau": str(ntt.eval_poly_at(F, h, tau)),
        "h_sha256": hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in h)).hexdigest(),
        "h_first": [str(x) for x in h[:4]], "h_last": [str(x) for x in h[-2:]],
    }
    json.dump(exp, open(os.path.join(DST, "expected.json"), "w"), indent=1)
    print(json.dumps(exp, indent=1))


if __name__ == "__main__":
    main()
