"""Shared test helpers: seeded inputs and conversions between oracle ints and C-ABI limb arrays."""
import functools
import random

import numpy as np

from oracle import curves as cv
from oracle import fields as fl

FIELD_IDS = {"bn254.Fq": 0, "bn254.Fr": 1, "bls12_381.Fq": 2, "bls12_381.Fr": 3, "bls12_377.Fq": 4, "bls12_377.Fr": 5}
CURVE_IDS = {"bn254": 0, "bls12_381": 1, "grumpkin": 2, "bls12_377": 3}
FR = {"bn254": fl.BN254_FR, "bls12_381": fl.BLS381_FR, "grumpkin": fl.BN254_FQ, "bls12_377": fl.BLS377_FR}   # scalar field of each curve


def rng(seed):
    return random.Random(seed)


def rand_elems(F, n, r):
    return [r.randrange(F.p) for _ in range(n)]


def edge_elems(F):
    return [0, 1, 2, F.p - 1, F.p - 2, (F.p - 1) // 2, (1 << 64) - 1, 1 << 64, F.Rmod, F.R2]


def pack(F, xs, mont=True):
    return fl.pack(F, xs, mont).reshape(-1)


def unpack(F, arr, mont=True):
    return fl.unpack(F, arr, mont)


def pack_shares(F, shares):
    """[(a, b)] -> AoS limbs {a, b} per entry."""
    flat = []
    for a, b in shares:
        flat += [a, b]
    return pack(F, flat)


def unpack_shares(F, arr):
    v = unpack(F, arr)
    return list(zip(v[0::2], v[1::2]))


def rand_points(curve: cv.Curve, n, r, with_inf=False):
    """n points = random multiples of the generator (in the prime-order subgroup)."""
    pts = []
    base = curve.mul(curve.gen, r.randrange(1, curve.order))
    step = curve.mul(curve.gen, r.randrange(1, curve.order))
    cur = base
    for i in range(n):
        pts.append(cur)
        cur = curve.add(cur, step)
    if with_inf and n >= 4:
        pts[1] = None
        pts[n // 2] = None
    return pts


def jac_to_affine(curve: cv.Curve, limbs):
    """C-ABI Jacobian output -> oracle affine point."""
    (X, Y, Z), = cv.unpack_points(curve, np.asarray(limbs), ncoords=3)
    return curve.to_affine((X, Y, Z))


def load_penumbra_fixture():
    """tests/golden/Groth16/bls12_377/penumbra_output (made by tests/golden/make_golden_penumbra.py from the reference's own
    LibSnarkReduction test data): -> (F, A, B, C, public, witness, expected dict)."""
    import gzip
    import json
    import os
    from oracle import arkfmt
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bls12_377", "penumbra_output")
    rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
    F = fl.BLS377_FR
    A, B, Cm = (arkfmt.parse_matrix(rd(n)) for n in ["a.bin", "b.bin", "c.bin"])
    prime, w = arkfmt.parse_wtns_positional(rd("witness.wtns"))
    assert prime == F.p
    exp = json.load(open(os.path.join(d, "expected.json")))
    ni = arkfmt.vk_num_instance_variables(rd("circuit.vk"), 96, 192)
    assert ni == exp["num_instance_variables"] and len(A) == exp["num_constraints"]
    return F, A, B, Cm, w[:ni], w[ni:], exp



@functools.lru_cache(maxsize=2)
def penumbra_libsnark_key(seed: int = 377):
    """A Groth16 key for the reference's Penumbra output circuit on BLS12-377 from the restated arkworks LibSnark generator
    (oracle.groth16.libsnark_setup) with seeded toxic waste -- the reference's own circuit.pk is absent upstream. Returns
    (fixture tuple, key dict, vk dict, ark-serialized ProvingKey bytes, oracle MSM through oracle/c)."""
    import random
    from oracle import arkfmt, cbridge as cb, groth16 as g16
    fx = load_penumbra_fixture()
    F, A, B, Cm, pub, wit, exp = fx
    G1, G2 = cv.BLS377_G1, cv.BLS377_G2
    rng = random.Random(seed)
    toxic = tuple(rng.randrange(1, F.p) for _ in range(5))

    def fixed_base(group, scalars):
        return cv.unpack_points((G1, G2)[group], cb.fixed_base_mul(3, group, fl.pack(F, scalars, mont=False)))

    def msm(G, pts, sc):
        if not pts:
            return None
        return cv.unpack_points(G, cb.msm_fast(3, 0 if G is G1 else 1, cv.pack_points(G, pts), fl.pack(F, sc)))[0]

    key = g16.libsnark_setup(F, exp["generator"], G1, G2, A, B, Cm, len(pub), len(wit), toxic, fixed_base)
    vk = {"alpha_g1": key["alpha_g1"], "beta_g2": key["beta_g2"], "gamma_g2": key["gamma_g2"], "delta_g2": key["delta_g2"], "ic": key["gamma_abc_g1"]}
    return fx, key, vk, arkfmt.ser_groth16_proving_key(key, G1.F.p, 48), msm
