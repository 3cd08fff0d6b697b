"""CPU tests that PIN the oracle: against the reference's committed fixtures (tests/golden copies of
test_vectors/Groth16: zkey, wtns, vk, snarkjs proof, public inputs), the BN254 Fr known-answer products,
and the C restatement against the Python one."""
import json
import os
import random

import numpy as np
import pytest

from oracle import cbridge, groth16, mpc, ntt, zkey
from oracle import curves as cv
from oracle import fields as fl
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CIRCUITS = [("bn254", "multiplier2"), ("bn254", "poseidon"), ("bls12_381", "multiplier2"), ("bls12_381", "poseidon")]


def load(curve, circ):
    d = os.path.join(GOLD, "Groth16", curve, circ)
    zk = zkey.parse_zkey(open(os.path.join(d, "circuit.zkey"), "rb").read())
    w = zkey.parse_wtns(open(os.path.join(d, "witness.wtns"), "rb").read())
    vk = zkey.parse_vk(open(os.path.join(d, "verification_key.json")).read())
    pr = zkey.parse_proof(open(os.path.join(d, "circom.proof")).read())
    pub = zkey.parse_public(open(os.path.join(d, "public.json")).read())
    return zk, w, vk, pr, pub


@pytest.mark.parametrize("curve,circ", CIRCUITS)
def test_committed_snarkjs_proof_verifies(curve, circ):
    """co-circom/co-groth16/src/lib.rs:72-91, 123-161: the verifier accepts the snarkjs proof."""
    zk, w, vk, pr, pub = load(curve, circ)
    assert groth16.verify(curve, zk.G1, vk, pr, pub)
    bad = dict(pr)
    bad["c"] = zk.G1.add(pr["c"], zk.G1.gen)
    assert not groth16.verify(curve, zk.G1, vk, bad, pub)


@pytest.mark.parametrize("curve,circ", [c for c in CIRCUITS if c != ("bls12_381", "poseidon")])
def test_restated_prover_verifies_and_matches_golden(curve, circ):
    """lib.rs:41-70, 93-121, 163-229: a proof from (zkey, wtns) verifies; plus the golden h / A / B / C."""
    zk, w, vk, pr, pub = load(curve, circ)
    proof, h = groth16.prove_plain(zk, w, 123456789, 987654321)
    assert groth16.verify(curve, zk.G1, vk, proof, pub)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    assert [str(x) for x in h] == gold["h"]
    assert [str(proof["a"][0]), str(proof["a"][1])] == gold["a"]
    assert [str(proof["c"][0]), str(proof["c"][1])] == gold["c"]
    assert [[str(proof["b"][0][0]), str(proof["b"][0][1])], [str(proof["b"][1][0]), str(proof["b"][1][1])]] == gold["b"]


def test_rep3_three_party_prove_equals_plain():
    """tests/tests/circom/e2e_tests/rep3.rs:38-86: parties agree on the proof; sum of h half-shares = plain h."""
    zk, w, vk, pr, pub = load("bn254", "multiplier2")
    rnd = random.Random(3)
    p1, h = groth16.prove_plain(zk, w, 11, 22)
    p3, hs = groth16.prove_rep3(zk, w, 11, 22, lambda: rnd.randrange(zk.Fr.p))
    assert p1 == p3
    assert [(a + b + c) % zk.Fr.p for a, b, c in zip(*hs)] == h
    assert groth16.verify("bn254", zk.G1, vk, p3, pub)


def test_bn254_fr_known_answer_products():
    F = fl.BN254_FR
    x = 13839525561076761625780930844889299788193703994911163378019280196128582690055
    y = 2688648969035332064113669477511029957484512453056743431884706385750388613065
    assert x * y % F.p == 14012338922664984944451142760937475581748095944353358534203030914664561190462
    got = cbridge.vec_mul(0, H.pack(F, [x]), H.pack(F, [y]))
    assert H.unpack(F, got) == [x * y % F.p]


def test_chacha_core_rfc7539_vector():
    """Pins oracle/chacha.py: RFC 7539 section 2.3.2 ChaCha20 block (same state layout as rand_chacha when the stream id
    words are zero); the Rep3 masks use the 12-round variant of the same core."""
    from oracle import chacha
    key = bytes(range(32))
    blk = chacha.block(key, 1 | (0x09000000 << 32), rounds=20, nonce_words=(0x4A000000, 0))
    assert blk[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4"
    assert chacha.keystream(key, 100, start_byte=30) == (chacha.block(key, 0) + chacha.block(key, 1) + chacha.block(key, 2))[30:130]


def test_snarkjs_roots_match_survey_values():
    q, roots = ntt.roots_of_unity(fl.BN254_FR)
    assert q == 5 and ntt.roots_of_unity(fl.BLS381_FR)[0] == 5
    assert roots[2] == 21888242871839275217838484774961031246007050428528088939761107053157389710902
    assert roots[3] == 19540430494807482326159819597004422086093766032135589407132600596362845576832


def test_shamir_share_reconstruct():
    """mpc-core/src/protocols/shamir.rs:611-742."""
    F = fl.BN254_FR
    r = random.Random(4)
    for (n, t) in [(3, 1), (10, 6)]:
        s = r.randrange(F.p)
        sh = mpc.shamir_share(F, s, n, t, lambda: r.randrange(F.p))
        idx = r.sample(range(1, n + 1), t + 1)
        assert mpc.shamir_reconstruct(F, [sh[i - 1] for i in idx], mpc.lagrange_from_coeff(F, idx)) == s


def test_rep3_to_shamir_translation_is_a_valid_sharing():
    """tests/tests/mpc/bridges.rs: translated shares reconstruct to the same secret (degree-1 Shamir)."""
    F = fl.BN254_FR
    r = random.Random(5)
    vals = [r.randrange(F.p) for _ in range(10)]
    sh = mpc.rep3_share_vec(F, vals, lambda: r.randrange(F.p))
    tr = [mpc.rep3_to_shamir_vec(F, sh[p], p) for p in range(3)]
    lag = mpc.lagrange_from_coeff(F, [1, 2])
    assert [mpc.shamir_reconstruct(F, [tr[0][i], tr[1][i]], lag) for i in range(10)] == vals
    lag = mpc.lagrange_from_coeff(F, [2, 3])
    assert [mpc.shamir_reconstruct(F, [tr[1][i], tr[2][i]], lag) for i in range(10)] == vals


# ---- C restatement vs Python restatement -----------------------------------------------------------------
@pytest.mark.parametrize("curve,group", [("bn254", 0), ("bn254", 1), ("bls12_381", 0), ("bls12_381", 1)])
def test_c_msm_matches_python(curve, group):
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    r = H.rng(group + 10)
    for n in [0, 1, 5, 40, 200]:
        pts = H.rand_points(G, n, r, with_inf=True)
        sc = H.rand_elems(F, n, r)
        if n >= 5:
            sc[0], sc[1], sc[2] = 0, 1, F.p - 1
        want = G.msm(pts, sc)
        for mont in (True, False):
            got = cbridge.msm(H.CURVE_IDS[curve], group, cv.pack_points(G, pts), H.pack(F, sc, mont=mont), montgomery=mont)
            assert G.eq(cv.unpack_points(G, got)[0], want)
            # the tuned restatement (Booth digits, XYZZ buckets; cpu_baseline "port" + full-size checker), several widths/threads
            for c, th in ((0, 0), (2, 1), (5, 3), (11, 2), (16, 1)):
                got = cbridge.msm_fast(H.CURVE_IDS[curve], group, cv.pack_points(G, pts), H.pack(F, sc, mont=mont), montgomery=mont, threads=th, c=c)
                assert G.eq(cv.unpack_points(G, got)[0], want), (n, mont, c, th)
    # doubling / cancellation inside one bucket, all-equal scalars, all-infinity bases
    pts = H.rand_points(G, 12, r)
    pts[1], pts[2], pts[5] = pts[0], pts[0], G.neg(pts[4])
    for sc in ([7] * 12, [F.p - 1] * 12, [0] * 12, [3, 3, 3, 9, 5, 5] * 2):
        got = cbridge.msm_fast(H.CURVE_IDS[curve], group, cv.pack_points(G, pts), H.pack(F, sc), c=4)
        assert G.eq(cv.unpack_points(G, got)[0], G.msm(pts, sc)), sc
    got = cbridge.msm_fast(H.CURVE_IDS[curve], group, cv.pack_points(G, [None] * 5), H.pack(F, [5] * 5))
    assert cv.unpack_points(G, got)[0] is None
    # naive double-and-add cross-check of the Python Pippenger itself
    pts = H.rand_points(G, 6, r)
    sc = H.rand_elems(F, 6, r)
    assert G.eq(G.msm(pts, sc), G.msm_naive(pts, sc))


@pytest.mark.parametrize("curve,group", [("bn254", 0), ("bn254", 1), ("bls12_381", 1)])
def test_c_progression_bases_match_python(curve, group):
    """oracle/c's cheap full-range point family (the G2 full-size GPU compares run on it): block b holds (s_b + j d_b) G"""
    G = cv.CURVES[curve][group]
    seed, n = 77, 600
    pts = cv.unpack_points(G, cbridge.generate_bases_progression(H.CURVE_IDS[curve], group, seed, n))

    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return x ^ (x >> 31)

    for i in (0, 1, 255, 256, 257, 599):
        b, j = divmod(i, 256)
        k = [sm(seed + 8 * b + l) for l in range(8)]
        k[3] >>= 3
        k[7] >>= 3
        k[4] |= 1
        s_b = sum(k[l] << (64 * l) for l in range(4))
        d_b = sum(k[4 + l] << (64 * l) for l in range(4))
        assert G.eq(pts[i], G.mul(G.gen, s_b + j * d_b)), i


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_c_ntt_and_vec_ops_match_python(curve):
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(17)
    for logn in [0, 1, 3, 8]:
        n = 1 << logn
        gen = ntt.roots_of_unity(F)[1][logn]
        do = ntt.Domain(F, n, gen)
        v = H.rand_elems(F, n, r)
        pg = H.pack(F, [gen])
        assert H.unpack(F, cbridge.ntt(cid, H.pack(F, v), logn, pg, dif=True)) == do.ifft_in_to_out(v)
        assert H.unpack(F, cbridge.ntt(cid, H.pack(F, v), logn, pg, dif=False)) == do.fft_out_to_in(v)
        assert H.unpack(F, cbridge.bit_reverse(H.pack(F, v), logn)) == ntt.bit_reverse(v)
        assert H.unpack(F, cbridge.coset_table(cid, H.pack(F, [7]), logn)) == ntt.bit_reversed_coset_table(F, 7, n)
    n = 100
    a = [(r.randrange(F.p), r.randrange(F.p)) for _ in range(n)]
    b = [(r.randrange(F.p), r.randrange(F.p)) for _ in range(n)]
    m = H.rand_elems(F, n, r)
    assert H.unpack(F, cbridge.rep3_local_mul_vec(cid, H.pack_shares(F, a), H.pack_shares(F, b), H.pack(F, m))) == mpc.rep3_local_mul_vec(F, a, b, m)
    x, y = mpc.rep3_to_shamir_points(F, 1)
    assert H.unpack(F, cbridge.rep3_to_shamir_vec(cid, H.pack_shares(F, a), H.pack(F, [x]), H.pack(F, [y]))) == mpc.rep3_to_shamir_vec(F, a, 1)


def test_c_generated_bases_closed_form():
    from tests.check_closed_form import closed_form_point
    G = cv.BN254_G1
    n = 1 << 10
    pts = cbridge.generate_bases(0, 0, 77, n)
    rs = np.random.RandomState(3)
    sc = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(3)
    got = cv.unpack_points(G, cbridge.msm(0, 0, pts, sc, True))[0]
    assert G.eq(got, closed_form_point("bn254", 0, 77, n, sc, True))
    assert (cbridge.msm_fast(0, 0, pts, sc, True) == cbridge.msm(0, 0, pts, sc, True)).all()


@pytest.mark.parametrize("curve,generator", [("bn254", 5), ("bls12_381", 7)])
def test_libsnark_reduction_restatement_satisfies_qap_identity(curve, generator):
    """LibSnarkReduction (reduction.rs:241-342) has no fixture in the reference tree (parity unpinned): the restatement is
    checked against the defining identity H(t) Z(t) = A(t) B(t) - C(t) at random points, with A, B, C interpolated by a
    direct Lagrange sum (no NTT code shared), and the three Rep3 parties' half shares must add up to the plain h."""
    import random
    from oracle import groth16 as g16
    from oracle import mpc
    F = H.FR[curve]
    rng = random.Random(2024)
    for n_public, n_constraints in [(1, 1), (2, 5), (3, 29)]:
        A, B, Cm, w = g16.random_r1cs(F, rng, n_public, n_constraints)
        pub, wit = w[:n_public], w[n_public:]
        h = g16.witness_map_libsnark(F, generator, A, B, Cm, n_constraints, g16.PlainDriver(F), pub, wit)
        for _ in range(2):
            assert g16.libsnark_identity_holds(F, generator, A, B, Cm, n_constraints, pub, wit, h, rng.randrange(F.p))
        bad = list(h)
        bad[0] = (bad[0] + 1) % F.p
        assert not g16.libsnark_identity_holds(F, generator, A, B, Cm, n_constraints, pub, wit, bad, rng.randrange(F.p))
        shares = mpc.rep3_share_vec(F, wit, lambda: rng.randrange(F.p))
        hs = [g16.witness_map_libsnark(F, generator, A, B, Cm, n_constraints, g16.Rep3Driver(F, pid), pub, shares[pid]) for pid in range(3)]
        assert [(x + y + z) % F.p for x, y, z in zip(*hs)] == h


def test_libsnark_reduction_on_the_reference_penumbra_fixture():
    """The reference's LibSnarkReduction tests (co-groth16/src/lib.rs:231-300) run on Penumbra BLS12-377 circuits whose
    matrices / witness / vk are in its tree (the proving key is not). On that data: every constraint is satisfied, the
    oracle's h satisfies the QAP identity, and equals the committed expected.json (tests/golden/make_golden_penumbra.py)."""
    import hashlib
    from oracle import groth16 as g16
    F, A, B, Cm, pub, wit, exp = H.load_penumbra_fixture()
    drv = g16.PlainDriver(F)
    for ra, rb, rc in list(zip(A, B, Cm))[::97]:
        assert drv.eval_row(ra, pub, wit) * drv.eval_row(rb, pub, wit) % F.p == drv.eval_row(rc, pub, wit)
    h = g16.witness_map_libsnark(F, exp["generator"], A, B, Cm, len(A), drv, pub, wit)
    assert len(h) == exp["domain_size"]
    assert hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in h)).hexdigest() == exp["h_sha256"]
    assert [str(x) for x in h[:4]] == exp["h_first"]
    assert g16.libsnark_identity_holds(F, exp["generator"], A, B, Cm, len(A), pub, wit, h, 0xDEADBEEF12345)


def test_ark_point_wire_format_on_the_reference_vk_fixture():
    """ark-serialize uncompressed G1 points (SWFlags in the top bits of y) as they appear in the reference's own
    circuit.vk (BLS12-377, tests/golden copy): every G1 point parses with consistent flags and lies on y^2 = x^3 + 1."""
    import gzip
    import os
    from oracle import arkfmt
    q = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bls12_377", "penumbra_output")
    vk = gzip.open(os.path.join(d, "circuit.vk.gz"), "rb").read()
    pts = []
    pt, off = arkfmt.parse_g1(vk, 0, q, 48)            # alpha_g1
    pts.append(pt)
    # beta_g2, gamma_g2, delta_g2: on the twist y^2 = x^3 + 1/u over Fq[u]/(u^2 + 5) and in the r-torsion -- the reference-held pin of
    # the oracle's BLS12-377 G2 (non-residue, twist constant, group order)
    from oracle import curves as cv
    G2 = cv.BLS377_G2
    for _ in range(3):
        Q, off = arkfmt.parse_g2(vk, off, q, 48)
        assert Q is not None and G2.is_on_curve(Q) and G2.mul(Q, G2.order) is None
    (k,) = __import__("struct").unpack_from("<Q", vk, off)
    off += 8
    for _ in range(k):                                 # gamma_abc_g1
        pt, off = arkfmt.parse_g1(vk, off, q, 48)
        pts.append(pt)
    assert off == len(vk) and k == 3
    for x, y in pts:
        assert (y * y - x * x * x - 1) % q == 0
        assert cv.BLS377_G1.mul((x, y), cv.BLS377_G1.order) is None
    # the serializer is the exact inverse
    assert b"".join(arkfmt.ser_g1(pt, q, 48) for pt in pts[1:]) == vk[-3 * 96:]



def test_libsnark_proof_on_the_reference_penumbra_circuit_verifies():
    """The whole of proof_libsnark_penumbra_output_bls12_377 (co-circom/co-groth16/src/lib.rs:231-290, 297-299) restated: a key from the
    arkworks LibSnark generator (the reference's circuit.pk is absent upstream, so the toxic waste is seeded here), the reference's a / b /
    c.bin + witness.wtns, plain_prove::<LibSnarkReduction>, Groth16::verify by the BLS12-377 pairing. Pins the reduction's root / coset
    choice (a wrong domain ordering or coset breaks H Z = A B - C at tau, hence the pairing equation) and the oracle's BLS12-377 G1 / G2 /
    Fq12 together; the serialized key starts with bytes laid out exactly as the reference's committed circuit.vk."""
    import gzip
    import os
    from oracle import arkfmt, curves as cv, groth16 as g16
    (F, A, B, Cm, pub, wit, exp), key, vk, pk_bytes, msm = H.penumbra_libsnark_key()
    G1, G2 = cv.BLS377_G1, cv.BLS377_G2
    r, s = 0x1234567890ABCDEF % F.p, (F.p - 5)
    proof, h = g16.prove_libsnark_plain(F, exp["generator"], G1, G2, key, A, B, Cm, pub, wit, r, s, msm=msm)
    assert g16.verify("bls12_377", G1, vk, proof, pub[1:])
    bad = dict(proof)
    bad["c"] = G1.add(proof["c"], G1.gen)
    assert not g16.verify("bls12_377", G1, vk, bad, pub[1:])
    assert not g16.verify("bls12_377", G1, vk, proof, [pub[1], (pub[2] + 1) % F.p])
    # the key's leading VerifyingKey has the shape of the reference's own circuit.vk: same length, same Vec count, parses the same way
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bls12_377", "penumbra_output")
    ref_vk = gzip.open(os.path.join(d, "circuit.vk.gz"), "rb").read()
    ours = pk_bytes[:len(ref_vk)]
    assert ours[96 + 3 * 192:96 + 3 * 192 + 8] == ref_vk[96 + 3 * 192:96 + 3 * 192 + 8]
    assert arkfmt.vk_num_instance_variables(ours, 96, 192) == exp["num_instance_variables"]


@pytest.mark.parametrize("curve,generator", [("bn254", 5), ("bls12_381", 7)])
def test_libsnark_setup_and_prover_restatements_verify_by_pairing(curve, generator):
    """The restated arkworks LibSnark generator + prover on the two north-star curves: a random satisfied R1CS, seeded toxic waste, the
    proof verifies under the key's vk (the same pairing code that accepts the reference's committed snarkjs proofs); a wrong h is rejected."""
    import random
    from oracle import cbridge as cb, curves as cv, fields as fl, groth16 as g16
    from tests import helpers as H
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    G1, G2 = cv.CURVES[curve]
    rng = random.Random(7 + cid)
    A, B, Cm, w = g16.random_r1cs(F, rng, 3, 40)
    pub, wit = w[:3], w[3:]
    toxic = tuple(rng.randrange(1, F.p) for _ in range(5))
    fixed_base = lambda group, sc: cv.unpack_points((G1, G2)[group], cb.fixed_base_mul(cid, group, fl.pack(F, sc, mont=False)))
    key = g16.libsnark_setup(F, generator, G1, G2, A, B, Cm, 3, len(wit), toxic, fixed_base)
    vk = {"alpha_g1": key["alpha_g1"], "beta_g2": key["beta_g2"], "gamma_g2": key["gamma_g2"], "delta_g2": key["delta_g2"], "ic": key["gamma_abc_g1"]}
    r, s = rng.randrange(F.p), rng.randrange(F.p)
    proof, h = g16.prove_libsnark_plain(F, generator, G1, G2, key, A, B, Cm, pub, wit, r, s)
    assert g16.verify(curve, G1, vk, proof, pub[1:])
    bad_h = list(h)
    bad_h[1] = (bad_h[1] + 1) % F.p
    bad, _ = g16.prove_libsnark_plain(F, generator, G1, G2, key, A, B, Cm, pub, wit, r, s, h=bad_h)
    assert not g16.verify(curve, G1, vk, bad, pub[1:])
