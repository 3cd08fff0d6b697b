"""GPU parity of the PLONK / UltraHonk driver call sites mirrored in co-snarks_amd/host/plonk_honk.hpp:
CircomPlonkProver::{fft, ifft, local_mul_vec, msm_public_points_g1} (co-plonk/src/mpc.rs:56-166) and
NoirUltraHonkProver::{fft, ifft, local_mul_vec, msm_public_points} / HonkCurve::fast_msm (co-noir-common/src/mpc/mod.rs:236-379,
honk_curve.rs:35) for the plain, Rep3 and Shamir drivers, against the oracle. Bit-exact."""
import numpy as np
import pytest

from oracle import curves as cv
from oracle import ntt
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _domain(F, size, snarkjs):
    if snarkjs:
        return ntt.Domain.snarkjs(F, size)
    gen = 5 if F is H.FR["bn254"] else 7
    root = pow(ntt.arkworks_two_adic_root(F, gen), 1 << (F.two_adicity - (size.bit_length() - 1)), F.p)
    return ntt.Domain(F, size, root)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("n_in,size,snarkjs", [(5, 8, True), (64, 64, True), (1000, 4096, True), (300, 1024, False)])
def test_driver_fft_ifft(gpu, curve, n_in, size, snarkjs):
    """EvaluationDomain::{fft, ifft}: natural -> natural, zero-padded input, domains n and 4n with snarkjs roots
    (co-plonk/src/types.rs:70-109) or arkworks' default root."""
    from cosnarks_amd import groth16 as dev
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(size + n_in)
    vals = H.rand_elems(F, n_in, r)
    dom = _domain(F, size, snarkjs)
    want_f, want_i = dom.fft(vals), dom.ifft(vals)
    data = H.pack(F, vals)
    for drv in (dev.PLAIN, dev.SHAMIR):
        assert H.unpack(F, dev.driver_fft(cid, drv, data, size, inverse=False, snarkjs=snarkjs)) == want_f
        assert H.unpack(F, dev.driver_fft(cid, drv, data, size, inverse=True, snarkjs=snarkjs)) == want_i
    for inverse, want in ((False, want_f), (True, want_i)):
        sh = dev.driver_fft(cid, dev.REP3, data, size, inverse=inverse, snarkjs=snarkjs, seed=9)
        a = [H.unpack(F, sh[p, :, 0, :]) for p in range(3)]
        b = [H.unpack(F, sh[p, :, 1, :]) for p in range(3)]
        assert [(x + y + z) % F.p for x, y, z in zip(*a)] == want           # the a components are an additive sharing
        assert b[0] == a[2] and b[1] == a[0] and b[2] == a[1]                 # replication survives the linear map
        assert a[0] != want


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_driver_local_mul_vec(gpu, curve):
    from cosnarks_amd import groth16 as dev
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(31)
    n = 777
    a, b = H.rand_elems(F, n, r), H.rand_elems(F, n, r)
    want = [x * y % F.p for x, y in zip(a, b)]
    for drv in (dev.PLAIN, dev.SHAMIR):
        assert H.unpack(F, dev.driver_local_mul_vec(cid, drv, H.pack(F, a), H.pack(F, b))) == want
    sh = dev.driver_local_mul_vec(cid, dev.REP3, H.pack(F, a), H.pack(F, b), seed=4)
    parts = [H.unpack(F, sh[p]) for p in range(3)]
    assert [(x + y + z) % F.p for x, y, z in zip(*parts)] == want           # masks cancel (rngs.rs:103-106)
    assert parts[0] != want and parts[0] != parts[1]


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_driver_msm_public_points(gpu, curve):
    from cosnarks_amd import groth16 as dev
    G = cv.CURVES[curve][0]
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(77)
    n = 200
    pts = H.rand_points(G, n, r, with_inf=True)
    sc = H.rand_elems(F, n - 7, r)                                            # msm_unchecked: shorter slice wins
    want = G.msm(pts[:n - 7], sc)
    pp = cv.pack_points(G, pts)
    for drv in (dev.PLAIN, dev.SHAMIR, dev.FAST_MSM):
        got = cv.unpack_points(G, dev.driver_msm(cid, drv, pp, H.pack(F, sc)))[0]
        assert G.eq(got, want), drv
    sh = dev.driver_msm(cid, dev.REP3, pp, H.pack(F, sc), seed=3)
    a = [cv.unpack_points(G, sh[p, 0])[0] for p in range(3)]
    b = [cv.unpack_points(G, sh[p, 1])[0] for p in range(3)]
    assert G.eq(G.add(G.add(a[0], a[1]), a[2]), want)                         # Rep3PointShare: a parts sum to the MSM
    assert G.eq(b[0], a[2]) and G.eq(b[1], a[0]) and G.eq(b[2], a[1])


def test_fast_msm_grumpkin(gpu):
    """HonkCurve::fast_msm for Projective<GrumpkinConfig> (honk_curve.rs:163-177)."""
    from cosnarks_amd import groth16 as dev
    G = cv.GRUMPKIN_G1
    F = H.FR["grumpkin"]
    r = H.rng(5)
    pts = H.rand_points(G, 150, r, with_inf=True)
    sc = H.rand_elems(F, 150, r)
    got = cv.unpack_points(G, dev.driver_msm(2, dev.FAST_MSM, cv.pack_points(G, pts), H.pack(F, sc)))[0]
    assert G.eq(got, G.msm(pts, sc))
