"""CPU tests pinning the oracle's RAW MSM and NTT on vectors the reference itself holds: the PLONK fixtures
test_vectors/Plonk/{bn254,bls12_381}/multiplier2 (loaded by co-circom/co-plonk/src/lib.rs:295-312).

* zkey sections 7-12 store, per selector / permutation polynomial, n coefficients and the 4n evaluations the prover uses as
  `zkey.*_poly.evaluations` (round3.rs:325-332, round5.rs:154): fft over the extended domain whose generator is the snarkjs
  root roots[pow + 2] (co-plonk/src/types.rs:70-109) of the coefficients == the stored evaluations;
* section 13's Lagrange polynomials: ifft over the n domain (roots[pow]) of a unit vector == the stored coefficients;
* the verifying key's Qm..S3 = msm(p_tau[0..n], coefficients) over the zkey's section-14 points (row a1);
* vk.w = roots[pow].
Nothing here was produced by this repository's arithmetic: tests/golden/make_golden_plonk.py only re-encodes the files."""
import os

import numpy as np
import pytest

from oracle import cbridge, ntt, zkey
from oracle import curves as cv
from tests import helpers as H
from tests import plonk_vectors as PV

CURVES = ["bn254", "bls12_381"]


@pytest.mark.parametrize("curve", CURVES)
def test_json_is_a_faithful_re_encoding_of_the_committed_files(curve):
    """The JSON the GPU tests read equals a fresh parse of the committed data files (no drift between the two)."""
    from tests.golden import make_golden_plonk as mk
    d = os.path.join(PV.GOLD, "Plonk", curve, "multiplier2")
    fresh = mk.extract(open(os.path.join(d, "circuit.zkey"), "rb").read(), open(os.path.join(d, "verification_key.json")).read())
    import json
    assert fresh == json.load(open(os.path.join(PV.GOLD, "plonk_golden.json")))[curve]
    g = PV.load(curve)
    assert g["vk"] == g["zkey_commitments"] and g["n"] == 1 << g["power"] and (g["k1"], g["k2"]) == (2, 3)
    G = cv.CURVES[curve][0]
    assert all(P is None or G.is_on_curve(P) for P in g["p_tau"]) and g["p_tau"][0] == G.gen   # tau^0 G


@pytest.mark.parametrize("curve", CURVES)
def test_snarkjs_root_is_the_vk_generator(curve):
    g = PV.load(curve)
    F = H.FR[curve]
    roots = ntt.roots_of_unity(F)[1]
    assert roots[g["power"]] == g["w"]
    assert pow(g["w"], g["n"], F.p) == 1 and pow(g["w"], g["n"] // 2, F.p) == F.p - 1


@pytest.mark.parametrize("curve", CURVES)
def test_oracle_fft_reproduces_stored_evaluations(curve):
    g = PV.load(curve)
    F = H.FR[curve]
    n = g["n"]
    roots = ntt.roots_of_unity(F)[1]
    ext = ntt.Domain(F, 4 * n, roots[g["power"] + 2])
    for nm in PV.POLYS:
        co, ev = g["polys"][nm]
        assert len(co) == n and len(ev) == 4 * n
        assert ext.fft(co + [0] * (3 * n)) == ev, nm
        assert ext.ifft(ev) == co + [0] * (3 * n), nm
        # the C restatement on the same vector (in_to_out / out_to_in forms + bit reversal = natural-order fft)
        cid, logn = H.CURVE_IDS[curve], g["power"] + 2
        got = cbridge.ntt(cid, H.pack(F, ntt.bit_reverse(co + [0] * (3 * n))), logn, H.pack(F, [roots[logn]]), dif=False)
        assert H.unpack(F, got) == ev, nm
        back = cbridge.ntt(cid, H.pack(F, ev), logn, H.pack(F, [roots[logn]]), dif=True)      # ifft_in_to_out: natural in, bit-reversed out
        assert ntt.bit_reverse(H.unpack(F, back)) == co + [0] * (3 * n), nm
    dom = ntt.Domain(F, n, roots[g["power"]])
    for i, (co, ev) in enumerate(g["lagrange"]):
        unit = [0] * n
        unit[i] = 1
        assert dom.ifft(unit) == co, i
        assert dom.fft(co) == unit
        assert ext.fft(co + [0] * (3 * n)) == ev, i


@pytest.mark.parametrize("curve", CURVES)
def test_oracle_msm_reproduces_vk_commitments(curve):
    g = PV.load(curve)
    F = H.FR[curve]
    G = cv.CURVES[curve][0]
    n = g["n"]
    pts = g["p_tau"][:n]
    for nm in PV.POLYS:
        co, _ = g["polys"][nm]
        want = g["vk"][nm]
        assert G.eq(G.msm(pts, co), want), nm
        # both C restatements (Jacobian/unsigned and XYZZ/Booth) on the same vector
        cid = H.CURVE_IDS[curve]
        pp = cv.pack_points(G, pts)
        for mont in (True, False):
            ps = H.pack(F, co, mont=mont)
            assert G.eq(cv.unpack_points(G, cbridge.msm(cid, 0, pp, ps, montgomery=mont))[0], want), (nm, mont)
            assert G.eq(cv.unpack_points(G, cbridge.msm_fast(cid, 0, pp, ps, montgomery=mont))[0], want), (nm, mont)
