"""Loader for tests/golden/plonk_golden.json (the reference's PLONK fixtures re-encoded by tests/golden/make_golden_plonk.py)."""
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POLYS = ["Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"]


def _pt(v):
    return None if v is None else (int(v[0]), int(v[1]))


def load(curve):
    g = json.load(open(os.path.join(GOLD, "plonk_golden.json")))[curve]
    return {
        "n": g["n"], "power": g["power"], "w": int(g["w"]), "k1": int(g["k1"]), "k2": int(g["k2"]),
        "polys": {nm: ([int(x) for x in p["coeffs"]], [int(x) for x in p["evaluations"]]) for nm, p in g["polys"].items()},
        "lagrange": [([int(x) for x in p["coeffs"]], [int(x) for x in p["evaluations"]]) for p in g["lagrange"]],
        "p_tau": [_pt(p) for p in g["p_tau"]],
        "vk": {nm: _pt(p) for nm, p in g["vk_commitments"].items()},
        "zkey_commitments": {nm: _pt(p) for nm, p in g["zkey_commitments"].items()},
    }
