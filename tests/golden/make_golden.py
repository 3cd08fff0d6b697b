#!/usr/bin/env python3
"""Generates tests/golden/: (1) copies the DATA fixtures the reference's own tests hold for this path
(test_vectors/Groth16/{bn254,bls12_381}/{multiplier2,poseidon}: zkey, wtns, verification key, snarkjs proof,
public inputs -- data files, not source); (2) golden h / A / B / C produced by the pinned oracle for fixed
r = 123456789, s = 987654321 (SURVEY.md section 8c lists the same values).  Run in the build container only:
    python tests/golden/make_golden.py
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import groth16, zkey  # noqa: E402

REF = "/root/reference/test_vectors/Groth16"
OUT = os.path.join(ROOT, "tests", "golden")
FILES = ["circuit.zkey", "witness.wtns", "verification_key.json", "circom.proof", "public.json"]
gold = {}
for curve in ["bn254", "bls12_381"]:
    for circ in ["multiplier2", "poseidon"]:
        dst = os.path.join(OUT, "Groth16", curve, circ)
        os.makedirs(dst, exist_ok=True)
        for f in FILES:
            shutil.copyfile(os.path.join(REF, curve, circ, f), os.path.join(dst, f))
            os.chmod(os.path.join(dst, f), 0o644)
        zk = zkey.parse_zkey(open(os.path.join(dst, "circuit.zkey"), "rb").read())
        w = zkey.parse_wtns(open(os.path.join(dst, "witness.wtns"), "rb").read())
        proof, h = groth16.prove_plain(zk, w, 123456789, 987654321)
        gold[f"{curve}/{circ}"] = {
            "r": "123456789", "s": "987654321", "h": [str(x) for x in h],
            "a": [str(proof["a"][0]), str(proof["a"][1])],
            "b": [[str(proof["b"][0][0]), str(proof["b"][0][1])], [str(proof["b"][1][0]), str(proof["b"][1][1])]],
            "c": [str(proof["c"][0]), str(proof["c"][1])],
        }
json.dump(gold, open(os.path.join(OUT, "groth16_golden.json"), "w"), indent=1)
print("wrote", OUT)
