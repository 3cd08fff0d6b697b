#!/usr/bin/env python3
"""Raw-boundary vectors held by the reference: the PLONK fixtures test_vectors/Plonk/{bn254,bls12_381}/multiplier2
(loaded by the reference's own tests, co-circom/co-plonk/src/lib.rs:295-312).

(1) copies the DATA files circuit.zkey + verification_key.json to tests/golden/Plonk/ (data, not source);
(2) extracts into tests/golden/plonk_golden.json, per curve:
      * for Qm, Ql, Qr, Qo, Qc (zkey sections 7-11) and S1, S2, S3 (section 12): the n coefficients and the 4n
        evaluations stored behind them -- what the prover reads as `zkey.*_poly.{coeffs, evaluations}`
        (round3.rs:325-332, round5.rs:154).  fft over the 4n domain with the snarkjs root roots[pow + 2]
        (types.rs:70-109) of the coefficients must give the evaluations: a raw `EvaluationDomain::fft` vector (row a5);
      * the Lagrange polynomials L_1.. of section 13 (coefficients = ifft of a unit vector over the n domain with
        roots[pow]; evaluations over 4n): a raw `ifft` vector;
      * the ptau points of section 14 and the commitments Qm..S3 of the verifying key:
        msm(ptau[0..n], coeffs) must equal vk.Q* / vk.S*: a raw `msm_unchecked` vector (row a1);
      * vk.w = the domain generator roots[pow] (row a6, snarkjs roots).
    Values are canonical integers as decimal strings; nothing in the JSON is computed by this repository's
    arithmetic -- the script only re-encodes what the files hold (Montgomery -> canonical).
Run in the build container only:  python tests/golden/make_golden_plonk.py
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import zkey  # noqa: E402

REF = "/root/reference/test_vectors/Plonk"
OUT = os.path.join(ROOT, "tests", "golden")
FILES = ["circuit.zkey", "verification_key.json"]


def pt(P):
    return None if P is None else [str(P[0]), str(P[1])]


def extract(zk_bytes, vk_text):
    zk = zkey.parse_plonk_zkey(zk_bytes)
    vk = zkey.parse_plonk_vk(vk_text)
    return {
        "n": zk.domain_size, "power": vk["power"], "n_public": zk.n_public, "k1": str(zk.k1), "k2": str(zk.k2), "w": str(vk["w"]),
        "polys": {nm: {"coeffs": [str(x) for x in co], "evaluations": [str(x) for x in ev]} for nm, (co, ev) in zk.polys.items()},
        "lagrange": [{"coeffs": [str(x) for x in co], "evaluations": [str(x) for x in ev]} for co, ev in zk.lagrange],
        "p_tau": [pt(P) for P in zk.p_tau],
        "vk_commitments": {nm: pt(vk[nm]) for nm in zkey.PLONK_POLY_SECTIONS},
        "zkey_commitments": {nm: pt(P) for nm, P in zk.commitments.items()},
    }


if __name__ == "__main__":
    gold = {}
    for curve in ["bn254", "bls12_381"]:
        dst = os.path.join(OUT, "Plonk", curve, "multiplier2")
        os.makedirs(dst, exist_ok=True)
        for f in FILES:
            shutil.copyfile(os.path.join(REF, curve, "multiplier2", f), os.path.join(dst, f))
            os.chmod(os.path.join(dst, f), 0o644)
        gold[curve] = extract(open(os.path.join(dst, "circuit.zkey"), "rb").read(), open(os.path.join(dst, "verification_key.json")).read())
    json.dump(gold, open(os.path.join(OUT, "plonk_golden.json"), "w"), indent=1)
    print("wrote", OUT)
