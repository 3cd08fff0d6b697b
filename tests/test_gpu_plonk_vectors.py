"""GPU parity of the RAW boundary on vectors the reference itself holds (no oracle arithmetic in the expected values):
the PLONK fixtures test_vectors/Plonk/{bn254,bls12_381}/multiplier2 (co-circom/co-plonk/src/lib.rs:295-312), re-encoded by
tests/golden/make_golden_plonk.py.  csh_fft / csh_ifft (EvaluationDomain::{fft, ifft}, rows a5 / a3 / a4 / a6) must reproduce the
zkey's stored 4n evaluations from its stored coefficients (round3.rs:325-332) with the snarkjs root roots[pow + 2]
(co-plonk/src/types.rs:70-109); csh_msm (msm_unchecked / msm_bigint, rows a1 / a2) over the zkey's ptau points must reproduce the
verifying key's commitments Qm..S3.  Bit-exact."""
import numpy as np
import pytest

from oracle import curves as cv
from tests import helpers as H
from tests import plonk_vectors as PV

pytestmark = pytest.mark.gpu
CURVES = ["bn254", "bls12_381"]


@pytest.mark.parametrize("curve", CURVES)
def test_fft_of_stored_coefficients_equals_stored_evaluations(gpu, curve):
    g = PV.load(curve)
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    n, pw = g["n"], g["power"]
    from oracle import ntt
    # the generator of the 4n domain is the fourth root of the n-domain generator vk.w that snarkjs picks: roots[pow + 2]
    gen4 = ntt.roots_of_unity(F)[1][pw + 2]
    assert pow(gen4, 4, F.p) == g["w"]                                       # ties the 4n generator to the reference's vk.w
    ext = gpu.Domain(cid, pw + 2, H.pack(F, [gen4]))
    dom = gpu.Domain(cid, pw, H.pack(F, [g["w"]]))
    for nm in PV.POLYS:
        co, ev = g["polys"][nm]
        padded = H.pack(F, co + [0] * (3 * n))
        assert H.unpack(F, ext.fft(padded)) == ev, nm
        assert H.unpack(F, ext.ifft(H.pack(F, ev))) == co + [0] * (3 * n), nm
        # the two half-transforms of reduction.rs:141-174 on the same vector: ifft_in_to_out = bit-reversed coefficients
        got = H.unpack(F, ext.ifft_in_to_out(H.pack(F, ev)))
        assert ntt.bit_reverse(got) == co + [0] * (3 * n), nm
        assert H.unpack(F, ext.fft_out_to_in(H.pack(F, got))) == ev, nm
    for i, (co, ev) in enumerate(g["lagrange"]):
        unit = [0] * n
        unit[i] = 1
        assert H.unpack(F, dom.ifft(H.pack(F, unit))) == co, i
        assert H.unpack(F, ext.fft(H.pack(F, co + [0] * (3 * n)))) == ev, i


@pytest.mark.parametrize("curve", CURVES)
def test_driver_fft_call_site_on_reference_vectors(gpu, curve):
    """CircomPlonkProver::fft through the host mirror's call site (zero-pads n coefficients to the 4n domain itself)."""
    from cosnarks_amd import groth16 as dev
    g = PV.load(curve)
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    n = g["n"]
    for nm in PV.POLYS:
        co, ev = g["polys"][nm]
        assert H.unpack(F, dev.driver_fft(cid, dev.PLAIN, H.pack(F, co), 4 * n, inverse=False, snarkjs=True)) == ev, nm
        assert H.unpack(F, dev.driver_fft(cid, dev.PLAIN, H.pack(F, ev), 4 * n, inverse=True, snarkjs=True)) == co + [0] * (3 * n), nm


@pytest.mark.parametrize("curve", CURVES)
def test_msm_over_ptau_equals_vk_commitments(gpu, curve):
    g = PV.load(curve)
    F = H.FR[curve]
    G = cv.CURVES[curve][0]
    cid = H.CURVE_IDS[curve]
    n = g["n"]
    bases = gpu.Bases(cid, gpu.G1, cv.pack_points(G, g["p_tau"][:n]))
    all_bases = gpu.Bases(cid, gpu.G1, cv.pack_points(G, g["p_tau"]))        # msm_unchecked: the shorter slice wins
    for nm in PV.POLYS:
        co, _ = g["polys"][nm]
        want = g["vk"][nm]
        for mont in (True, False):                                            # msm_unchecked / msm_bigint
            got = H.jac_to_affine(G, bases.msm(H.pack(F, co, mont=mont), montgomery=mont))
            assert got == want, (nm, mont)
        assert H.jac_to_affine(G, all_bases.msm(H.pack(F, co))) == want, nm


@pytest.mark.parametrize("curve", CURVES)
def test_driver_msm_call_site_on_reference_vectors(gpu, curve):
    """CircomPlonkProver::msm_public_points_g1 (co-plonk/src/mpc.rs:164) for the plain / Shamir drivers and HonkCurve::fast_msm."""
    from cosnarks_amd import groth16 as dev
    g = PV.load(curve)
    F = H.FR[curve]
    G = cv.CURVES[curve][0]
    cid = H.CURVE_IDS[curve]
    pp = cv.pack_points(G, g["p_tau"])
    for nm in PV.POLYS:
        co, _ = g["polys"][nm]
        for drv in (dev.PLAIN, dev.SHAMIR, dev.FAST_MSM):
            assert cv.unpack_points(G, dev.driver_msm(cid, drv, pp, H.pack(F, co)))[0] == g["vk"][nm], (nm, drv)
        sh = dev.driver_msm(cid, dev.REP3, pp, H.pack(F, co), seed=11)
        a = [cv.unpack_points(G, sh[p, 0])[0] for p in range(3)]
        assert G.add(G.add(a[0], a[1]), a[2]) == g["vk"][nm], nm
