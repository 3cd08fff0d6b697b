"""GPU parity for the whole path behind the reference's prover interface: the C++ host mirror
(PlainGroth16Driver / Rep3Groth16Driver + CircomReduction + CoGroth16::prove) running its MSMs, NTTs and share
arithmetic through the C ABI. Mirrors the reference's own tests: a proof from (zkey, wtns) verifies under
verification_key.json (co-circom/co-groth16/src/lib.rs:41-70, 93-121, 163-229); three Rep3 parties agree and the
proof verifies (tests/tests/circom/e2e_tests/rep3.rs:38-86). Plus bit-exact A/B/C/h vs the pinned oracle for fixed r, s."""
import json
import os

import numpy as np
import pytest

from oracle import groth16 as og
from oracle import ntt
from oracle import zkey as oz
from tests import helpers as H

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CIRCUITS = [("bn254", "multiplier2"), ("bn254", "poseidon"), ("bls12_381", "multiplier2"), ("bls12_381", "poseidon")]
R, S = 123456789, 987654321


def _load(curve, circ):
    d = os.path.join(GOLD, "Groth16", curve, circ)
    rd = lambda f, m="rb": open(os.path.join(d, f), m).read()
    return rd("circuit.zkey"), rd("witness.wtns"), oz.parse_vk(rd("verification_key.json", "r")), oz.parse_public(rd("public.json", "r"))


def _as_points(j):
    return oz.parse_proof(json.dumps(j))


@pytest.mark.parametrize("curve,circ", CIRCUITS)
def test_plain_prove_matches_golden_and_verifies(gpu, curve, circ):
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    proof, h = g.prove_plain(H.CURVE_IDS[curve], zk, wt, R, S, want_h=True, h_elems=zko.domain_size)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    assert proof["pi_a"][:2] == gold["a"]
    assert proof["pi_b"][:2] == gold["b"]
    assert proof["pi_c"][:2] == gold["c"]
    assert [str(x) for x in H.unpack(zko.Fr, h)] == gold["h"]
    assert og.verify(curve, zko.G1, vk, _as_points(proof), pub)          # snarkjs verification equation
    assert proof["protocol"] == "groth16" and proof["pi_a"][2] == "1"    # circom.proof schema


@pytest.mark.parametrize("curve,circ", [("bn254", "poseidon"), ("bls12_381", "poseidon")])
def test_h_pipeline_unfused_sequence_matches_golden(gpu, curve, circ):
    """The default h pipeline folds the coset table into the last pass of the inverse transforms and a b - c into one kernel; the
    step-by-step sequence of reduction.rs:135-192 (tune h_unfused) must give the same h, plain and with three Rep3 parties."""
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    with gpu.tuned(h_unfused=1):
        proof, h = g.prove_plain(H.CURVE_IDS[curve], zk, wt, R, S, want_h=True, h_elems=zko.domain_size)
        proof3, hs = g.prove_rep3(H.CURVE_IDS[curve], zk, wt, seed=42, r=R, s=S, want_h=True, h_elems=zko.domain_size)
    assert [str(x) for x in H.unpack(zko.Fr, h)] == gold["h"]
    n = zko.domain_size
    parts = [H.unpack(zko.Fr, hs[4 * n * p:4 * n * (p + 1)]) for p in range(3)]
    assert [str((a + b + c) % zko.Fr.p) for a, b, c in zip(*parts)] == gold["h"]
    for p in (proof, proof3):
        assert p["pi_a"][:2] == gold["a"] and p["pi_b"][:2] == gold["b"] and p["pi_c"][:2] == gold["c"]


@pytest.mark.parametrize("curve,circ", [("bn254", "multiplier2"), ("bn254", "poseidon")])
def test_plain_prove_with_fresh_randomness_verifies(gpu, curve, circ):
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    p1, _ = g.prove_plain(H.CURVE_IDS[curve], zk, wt)
    p2, _ = g.prove_plain(H.CURVE_IDS[curve], zk, wt)
    assert p1 != p2                                                     # r, s are fresh (mpc/plain.rs:23-26)
    assert og.verify(curve, zko.G1, vk, _as_points(p1), pub)
    assert og.verify(curve, zko.G1, vk, _as_points(p2), pub)


@pytest.mark.parametrize("curve,circ", [("bn254", "multiplier2"), ("bn254", "poseidon"), ("bls12_381", "multiplier2"), ("bls12_381", "poseidon")])
def test_rep3_three_parties_agree_and_match_plain(gpu, curve, circ):
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    F = zko.Fr
    proof, hs = g.prove_rep3(H.CURVE_IDS[curve], zk, wt, seed=42, r=R, s=S, want_h=True, h_elems=zko.domain_size)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]
    n = zko.domain_size
    parts = [H.unpack(F, hs[4 * n * p:4 * n * (p + 1)]) for p in range(3)]
    assert parts[0] != [int(x) for x in gold["h"]]                      # masked: no party holds h itself
    assert [str((a + b + c) % F.p) for a, b, c in zip(*parts)] == gold["h"]   # masks cancel (rngs.rs:103-106)
    assert og.verify(curve, zko.G1, vk, _as_points(proof), pub)
    fresh, _ = g.prove_rep3(H.CURVE_IDS[curve], zk, wt, seed=7)          # r, s from the correlated randomness
    assert og.verify(curve, zko.G1, vk, _as_points(fresh), pub)


@pytest.mark.parametrize("curve,circ,n,t,bridge", [("bn254", "multiplier2", 3, 1, False), ("bn254", "poseidon", 3, 1, False),
                                                   ("bn254", "multiplier2", 5, 2, False), ("bls12_381", "multiplier2", 3, 1, False),
                                                   ("bn254", "poseidon", 3, 1, True)])
def test_shamir_parties_agree_and_match_plain(gpu, curve, circ, n, t, bridge):
    """tests/tests/circom/e2e_tests/shamir.rs:34-...: n Shamir parties (threshold t) agree on a verifying proof;
    with the dealt randomness sharing r, s it is the plain proof. bridge=True: prove_with_shamir_bridge
    (e2e_tests/rep3.rs:88-137), Rep3 shares translated on the device."""
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    proof = g.prove_shamir(H.CURVE_IDS[curve], zk, wt, n, t, seed=11, r=R, s=S, bridge=bridge)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]
    fresh = g.prove_shamir(H.CURVE_IDS[curve], zk, wt, n, t, seed=12, bridge=bridge)
    assert og.verify(curve, zko.G1, vk, _as_points(fresh), pub)
    with pytest.raises(gpu.CoSnarksHipError, match="at least 2 \\* threshold \\+ 1"):
        g.prove_shamir(H.CURVE_IDS[curve], zk, wt, 2 * t, t, seed=1)


@pytest.mark.parametrize("curve,circ,protocol,compression,n,t", [
    ("bn254", "multiplier2", "rep3", 0, 3, 1), ("bn254", "poseidon", "rep3", 1, 3, 1), ("bls12_381", "poseidon", "rep3", 0, 3, 1),
    ("bls12_381", "multiplier2", "rep3", 1, 3, 1), ("bn254", "poseidon", "rep3", 3, 3, 1), ("bls12_381", "poseidon", "rep3", 3, 3, 1),
    ("bn254", "multiplier2", "rep3", 2, 3, 1), ("bn254", "poseidon", "shamir", 0, 3, 1), ("bls12_381", "multiplier2", "shamir", 0, 5, 2)])
def test_prove_from_witness_share_files(gpu, curve, circ, protocol, compression, n, t):
    """`co-circom split-witness` then `generate-proof` (co-circom.rs:660-740, 1008-1050): every party reads its bincode
    `.shared` file (compression 0 replicated, 1 additive half shares completed with one reshare_vec round,
    2 / 3 their seeded forms -- 3 is what the reference's CLI writes) and the parties' proof is the plain proof for the
    same r, s."""
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    cid = H.CURVE_IDS[curve]
    files = g.split_witness(cid, protocol, wt, zko.n_public + 1, seed=21, compression=compression, threshold=t, num_parties=n)
    proof = g.prove_from_shares(cid, protocol, zk, files, threshold=t, seed=5, r=R, s=S)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]
    fresh = g.prove_from_shares(cid, protocol, zk, files, threshold=t, seed=6)
    assert og.verify(curve, zko.G1, vk, _as_points(fresh), pub)
    # files of another circuit / a wrong party count are refused before any device work
    other = g.split_witness(cid, protocol, wt, zko.n_public + 2, seed=1, compression=compression, threshold=t, num_parties=n)
    with pytest.raises(gpu.CoSnarksHipError, match="public input count"):
        g.prove_from_shares(cid, protocol, zk, other, threshold=t)
    if protocol == "rep3":
        with pytest.raises(gpu.CoSnarksHipError, match="three parties"):
            g.prove_from_shares(cid, protocol, zk, files[:2], threshold=t)
        mixed = [files[0]] + g.split_witness(cid, "rep3", wt, zko.n_public + 1, seed=21, compression=(compression + 1) % 4)[1:]
        with pytest.raises(gpu.CoSnarksHipError, match="different compression"):
            g.prove_from_shares(cid, protocol, zk, mixed)


@pytest.mark.parametrize("curve,circ,compression", [("bn254", "poseidon", 0), ("bn254", "multiplier2", 1), ("bls12_381", "multiplier2", 0),
                                                    ("bls12_381", "poseidon", 3), ("bn254", "multiplier2", 2)])
def test_translate_witness_share_files(gpu, curve, circ, compression):
    """`co-circom translate-witness` (lib.rs:93-135): Rep3 `.shared` files -> Shamir `.shared` files on the device
    (translate_primefield_repshare_vec); the translated shares reconstruct the witness (any two parties, degree 1), match
    the oracle's translation share for share, and the Shamir parties' proof from them is the plain proof."""
    from cosnarks_amd import groth16 as g
    from oracle import arkfmt, mpc
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    F = zko.Fr
    cid = H.CURVE_IDS[curve]
    npub = zko.n_public + 1
    w = oz.parse_wtns(wt)
    rep = g.split_witness(cid, "rep3", wt, npub, seed=33, compression=compression)
    sham = g.translate_witness(cid, rep)
    parsed = [arkfmt.parse_shamir_share_file(f) for f in sham]
    for p in range(3):
        assert parsed[p][0] == w[:npub]
    for ids in ([0, 1], [1, 2], [0, 2]):
        lag = mpc.lagrange_from_coeff(F, [i + 1 for i in ids])
        assert [mpc.shamir_reconstruct(F, [parsed[i][1][k] for i in ids], lag) for k in range(len(w) - npub)] == w[npub:]
    if compression == 0:
        for p in range(3):
            _, _, shares = arkfmt.parse_rep3_share_file(rep[p])
            assert parsed[p][1] == mpc.rep3_to_shamir_vec(F, shares, p)         # bridges/rep3_to_shamir.rs:43-62
    proof = g.prove_from_shares(cid, "shamir", zk, sham, threshold=1, seed=5, r=R, s=S)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]


@pytest.mark.parametrize("family,proto", [("rep3_replicated", "rep3"), ("rep3_seeded_replicated", "rep3"), ("rep3_additive", "rep3"),
                                          ("rep3_seeded_additive", "rep3"), ("shamir_t1", "shamir")])
def test_prove_from_committed_share_files(gpu, family, proto):
    """The committed `.shared` fixtures (tests/golden/share_files, written by the oracle's restatement) prove to the golden
    plain proof of the reference's multiplier2 circuit."""
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load("bn254", "multiplier2")
    d = os.path.join(GOLD, "share_files")
    files = [open(os.path.join(d, f"{family}.{p}.shared"), "rb").read() for p in range(3)]
    proof = g.prove_from_shares(0, proto, zk, files, threshold=1, seed=9, r=R, s=S)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))["bn254/multiplier2"]
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]
    assert json.loads(g.public_inputs_json(0, proto, files[1])) == json.load(open(os.path.join(GOLD, "Groth16", "bn254", "multiplier2", "public.json")))


def test_prove_rejects_wrong_witness_length(gpu):
    from cosnarks_amd import groth16 as g
    zk, wt, _, _ = _load("bn254", "multiplier2")
    _, wt_big, _, _ = _load("bn254", "poseidon")
    with pytest.raises(gpu.CoSnarksHipError, match="amount of private witness variables does not match"):
        g.prove_plain(0, zk, wt_big, R, S)
    with pytest.raises(gpu.CoSnarksHipError, match="does not match the selected curve"):
        g.prove_plain(1, zk, wt, R, S)


@pytest.mark.parametrize("curve,logd", [("bn254", 12), ("bn254", 16), ("bls12_381", 12)])
def test_synthetic_circuit_prove_closed_form(gpu, curve, logd):
    """SURVEY 8d config 1 (synthetic large circuit, known-dlog key): A, B, C equal their closed-form discrete logs."""
    from cosnarks_amd import groth16 as g
    res = g.bench_synthetic(H.CURVE_IDS[curve], logd, iters=1, with_rep3=(logd == 12))
    if logd == 12:
        assert res["rep3_proofs_equal_plain"]          # three Rep3 parties reproduce the plain proof of the synthetic circuit
    assert res["closed_form_check"], res


# ---- one prover's five query MSMs placed on several GPUs (groth16.rs:227-294: the closures of rayon_join5 are independent) ----
# The device list may name a GPU more than once: every slot then clones its queries (csh_bases_clone), receives the scalars by
# csh_memcpy_peer and runs its MSM group from its own host thread -- the whole multi-GPU code path on the one GPU of this box.
@pytest.mark.parametrize("devices,mode", [([0, 0], 0), ([0, 0, 0], 0), ([0, 0, 0], 1), ([0, 0, 0, 0, 0, 0], 1), ([0, 0], 2), ([0] * 8, 2)])
@pytest.mark.parametrize("curve,circ", [("bn254", "poseidon"), ("bls12_381", "multiplier2")])
def test_plain_prove_with_placed_queries_matches_golden(gpu, curve, circ, devices, mode):
    """mode 1: whole queries per slot; mode 2: the k-th range of every query per slot (multiplier2 has fewer entries than slots:
    some ranges are empty); mode 0: automatic"""
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    g.set_prover_devices(devices, mode)
    try:
        proof, h = g.prove_plain(H.CURVE_IDS[curve], zk, wt, R, S, want_h=True, h_elems=zko.domain_size)
    finally:
        g.set_prover_devices(None)
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]
    assert [str(x) for x in H.unpack(zko.Fr, h)] == gold["h"]
    assert og.verify(curve, zko.G1, vk, _as_points(proof), pub)


@pytest.mark.parametrize("devices,mode", [([0, 0], 1), ([0, 0, 0, 0, 0], 1), ([0, 0, 0], 2)])
def test_rep3_prove_with_placed_queries_matches_golden(gpu, devices, mode):
    from cosnarks_amd import groth16 as g
    curve, circ = "bn254", "poseidon"
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    gold = json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]
    g.set_prover_devices(devices, mode)
    try:
        proof, _ = g.prove_rep3(H.CURVE_IDS[curve], zk, wt, seed=42, r=R, s=S)
        shamir = g.prove_shamir(H.CURVE_IDS[curve], zk, wt, 3, 1, seed=11, r=R, s=S)
    finally:
        g.set_prover_devices(None)
    for p in (proof, shamir):
        assert p["pi_a"][:2] == gold["a"] and p["pi_b"][:2] == gold["b"] and p["pi_c"][:2] == gold["c"]
    assert og.verify(curve, zko.G1, vk, _as_points(proof), pub)


@pytest.mark.parametrize("curve,logd,devices,mode", [("bn254", 14, [0, 0, 0], 1), ("bn254", 14, [0, 0, 0], 2), ("bn254", 16, [0, 0], 0),
                                                     ("bn254", 16, [0] * 8, 2), ("bls12_381", 14, [0, 0, 0, 0, 0], 1), ("bls12_381", 14, [0, 0, 0], 2)])
def test_synthetic_prove_with_placed_queries_closed_form(gpu, curve, logd, devices, mode):
    """keys large enough for fixed-base tables: the clones carry the tables; plain + three Rep3 parties, closed form"""
    from cosnarks_amd import groth16 as g
    g.set_prover_devices(devices, mode)
    try:
        res = g.bench_synthetic(H.CURVE_IDS[curve], logd, iters=2, with_rep3=(logd == 14))
    finally:
        g.set_prover_devices(None)
    assert res["closed_form_check"], res
    if logd == 14:
        assert res["rep3_proofs_equal_plain"]
    ph = res["prove_phases_ms"]
    assert ph["msm_groups"] > 0 and ph["witness_upload_and_map"] > 0 and ph["msm_groups"] + ph["witness_upload_and_map"] <= res["prove_ms"] + 1e-6


def test_set_prover_devices_rejects_unknown_device(gpu):
    from cosnarks_amd import groth16 as g
    with pytest.raises(gpu.CoSnarksHipError, match="device out of range"):
        g.set_prover_devices([0, 97])
    g.set_prover_devices(None)


def test_bases_clone_and_peer_copy(gpu):
    """csh_bases_clone: an MSM on the clone (tables included) equals the MSM on the original; csh_memcpy_peer moves the scalars"""
    import ctypes as C

    import numpy as np

    from cosnarks_amd import bindings as B
    L = gpu.lib()
    n = 5000
    buf = gpu.DeviceBuffer(n * 64)
    B._check(L.csh_util_generate_bases_dev(0, 0, C.c_uint64(7), C.c_size_t(n), buf.ptr, None))
    B.sync()
    h, h2 = C.c_void_p(), C.c_void_p()
    B._check(L.csh_bases_upload_dev(0, 0, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
    B._check(L.csh_bases_precompute_grouped(h, 12, 4))
    B._check(L.csh_bases_clone(h, 0, C.byref(h2)))
    rs = np.random.RandomState(3)
    sc = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    d1 = gpu.DeviceBuffer.from_host(sc)
    d2 = gpu.DeviceBuffer(n * 32)
    B._check(L.csh_memcpy_peer(d2.ptr, 0, d1.ptr, 0, C.c_size_t(n * 32), None))
    o1, o2 = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
    B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), d1.ptr, 1, o1.ctypes.data_as(C.c_void_p), None))
    B._check(L.csh_msm_dev(h2, C.c_size_t(0), C.c_size_t(n), d2.ptr, 1, o2.ctypes.data_as(C.c_void_p), None))
    assert (o1 == o2).all() and o1.any()
    B.tune_set("msm_no_table", 1)
    try:
        o3 = np.zeros(12, dtype=np.uint64)
        B._check(L.csh_msm_dev(h2, C.c_size_t(0), C.c_size_t(n), d2.ptr, 1, o3.ctypes.data_as(C.c_void_p), None))
    finally:
        B.tune_set("msm_no_table", 0)
    assert (o3 == o1).all()
    B._check(L.csh_bases_drop_tables(h2))
    B._check(L.csh_msm_dev(h2, C.c_size_t(0), C.c_size_t(n), d2.ptr, 1, o3.ctypes.data_as(C.c_void_p), None))
    assert (o3 == o1).all()
    c, rows = C.c_int(0), C.c_int(0)
    B._check(L.csh_bases_table_policy(C.c_size_t(1 << 20), C.byref(c), C.byref(rows)))
    assert (c.value, rows.value) == (17, 16)      # round 6: one bucket set, one row per window, 17-bit windows up to 3 * 2^20 points
    B._check(L.csh_bases_table_policy(C.c_size_t(1 << 16), C.byref(c), C.byref(rows)))
    assert (c.value, rows.value) == (17, 16)
    B._check(L.csh_bases_table_policy(C.c_size_t(1 << 24), C.byref(c), C.byref(rows)))
    assert (c.value, rows.value) == (20, 13)
    B._check(L.csh_bases_table_policy(C.c_size_t(1 << 14), C.byref(c), C.byref(rows)))
    assert (c.value, rows.value) == (13, 16)
    B._check(L.csh_bases_table_policy(C.c_size_t(1000), C.byref(c), C.byref(rows)))
    assert rows.value == 0
    # csh_bases_clone_range: point i of the clone = point offset + i of the source, tables included (placement by range holds 1 / N of a
    # key per GPU): the MSM of a sub-range on the range clone equals the same sub-range on the original, with and without tables
    h3 = C.c_void_p()
    B._check(L.csh_bases_precompute_grouped(h, 12, 4))
    off, cnt = 1234, 3001
    B._check(L.csh_bases_clone_range(h, C.c_size_t(off), C.c_size_t(cnt), 0, C.byref(h3)))
    ln = C.c_size_t(0)
    B._check(L.csh_bases_len(h3, C.byref(ln)))
    assert ln.value == cnt
    d3 = gpu.DeviceBuffer.from_host(sc[off:off + cnt])
    for sub_off, sub_n in ((0, cnt), (17, 2500), (cnt - 1, 1), (5, 0)):
        B._check(L.csh_msm_dev(h, C.c_size_t(off + sub_off), C.c_size_t(sub_n), C.c_void_p(d1.ptr.value + 32 * (off + sub_off)), 1, o1.ctypes.data_as(C.c_void_p), None))
        B._check(L.csh_msm_dev(h3, C.c_size_t(sub_off), C.c_size_t(sub_n), C.c_void_p(d3.ptr.value + 32 * sub_off), 1, o2.ctypes.data_as(C.c_void_p), None))
        assert (o1 == o2).all(), (sub_off, sub_n)
    assert L.csh_bases_clone_range(h, C.c_size_t(n - 10), C.c_size_t(11), 0, C.byref(h2)) != 0      # range past the end
    for x in (h, h2, h3):
        L.csh_bases_free(x)
    for d in (buf, d1, d2, d3):
        d.free()


@pytest.mark.gpu
@pytest.mark.parametrize("curve,generator", [("bn254", 5), ("bls12_381", 7)])
@pytest.mark.parametrize("n_public,n_constraints", [(1, 2), (3, 29), (2, 3000)])
def test_libsnark_reduction_device_matches_oracle(gpu, curve, generator, n_public, n_constraints):
    """LibSnarkReduction::witness_map_from_matrices on the device (csh_groth16_witness_map_libsnark through the host
    mirror) vs the oracle restatement: plain h bit-exact; three Rep3 parties (device ChaCha12 masks) sum to it."""
    import random
    from cosnarks_amd import groth16 as dev
    from oracle import groth16 as g16
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    rng = random.Random(77 + n_constraints)
    A, B, Cm, w = g16.random_r1cs(F, rng, n_public, n_constraints)
    pub, wit = w[:n_public], w[n_public:]
    want = g16.witness_map_libsnark(F, generator, A, B, Cm, n_constraints, g16.PlainDriver(F), pub, wit)
    mont = lambda M: [[(F.to_mont(c), i) for c, i in row] for row in M]
    mats = (mont(A), mont(B), mont(Cm))
    wm = H.pack(F, w)
    got = dev.witness_map(cid, dev.LIBSNARK_REDUCTION, False, mats, n_public, wm)
    assert H.unpack(F, got) == want
    if n_constraints <= 100:
        assert g16.libsnark_identity_holds(F, generator, A, B, Cm, n_constraints, pub, wit, H.unpack(F, got), rng.randrange(F.p))
    hs = dev.witness_map(cid, dev.LIBSNARK_REDUCTION, True, mats, n_public, wm, seed=5)
    parts = [H.unpack(F, hs[p]) for p in range(3)]
    assert [(x + y + z) % F.p for x, y, z in zip(*parts)] == want
    assert parts[0] != want                      # masked shares, not the plain vector three times


@pytest.mark.gpu
def test_libsnark_reduction_device_on_the_reference_penumbra_fixture(gpu):
    """LibSnarkReduction on the device, BLS12-377 Fr, on the reference's own Penumbra test data (13875 constraints, domain
    2^14): h bit-exact against expected.json, the QAP identity at the committed point, three Rep3 parties sum to h."""
    import hashlib
    from cosnarks_amd import groth16 as dev
    from oracle import ntt
    F, A, B, Cm, pub, wit, exp = H.load_penumbra_fixture()
    cid = H.CURVE_IDS["bls12_377"]
    mont = lambda M: [[(F.to_mont(c), i) for c, i in row] for row in M]
    mats = (mont(A), mont(B), mont(Cm))
    wm = H.pack(F, pub + wit)
    got = H.unpack(F, dev.witness_map(cid, dev.LIBSNARK_REDUCTION, False, mats, len(pub), wm))
    assert len(got) == exp["domain_size"]
    assert hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in got)).hexdigest() == exp["h_sha256"]
    assert str(ntt.eval_poly_at(F, got, int(exp["tau"]))) == exp["H_at_tau"]
    hs = dev.witness_map(cid, dev.LIBSNARK_REDUCTION, True, mats, len(pub), wm, seed=11)
    parts = [H.unpack(F, hs[p]) for p in range(3)]
    assert [(x + y + z) % F.p for x, y, z in zip(*parts)] == got


@pytest.mark.gpu
def test_libsnark_reduction_from_the_reference_file_formats(gpu):
    """The host mirror reads the reference's ark-serialize Matrix blobs and wtns container itself (arkwire.hpp)
    and runs LibSnarkReduction on the device: same h as expected.json."""
    import ctypes as C
    import gzip
    import hashlib
    import json
    import os
    import numpy as np
    from cosnarks_amd import groth16 as dev
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bls12_377", "penumbra_output")
    rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
    exp = json.load(open(os.path.join(d, "expected.json")))
    a, b, c, w = rd("a.bin"), rd("b.bin"), rd("c.bin"), rd("witness.wtns")
    out = np.zeros(exp["domain_size"] * 4, dtype=np.uint64)
    L = dev.glib()
    n = L.cog16_libsnark_from_files(3, a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), w, C.c_size_t(len(w)),
                                    C.c_size_t(exp["num_instance_variables"]), out.ctypes.data_as(C.c_void_p), C.c_size_t(exp["domain_size"]))
    assert n == exp["domain_size"], L.cog16_last_error()
    F = H.FR["bls12_377"]
    got = H.unpack(F, out)
    assert hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in got)).hexdigest() == exp["h_sha256"]
    # a truncated blob is rejected, not mis-parsed
    assert L.cog16_libsnark_from_files(3, a[:-5], C.c_size_t(len(a) - 5), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), w, C.c_size_t(len(w)),
                                       C.c_size_t(3), out.ctypes.data_as(C.c_void_p), C.c_size_t(exp["domain_size"])) == -1



@pytest.mark.parametrize("logn", [2, 3, 8, 11, 12, 13, 16, 20, 21])
def test_h_with_fused_tile_passes_equals_two_launch_and_unfused_forms(gpu, logn):
    """Round 6: the last pass of each inverse transform and the first pass of the forward transform that follows it work on
    the same contiguous tiles and run as ONE launch (k_ntt_pass_r4<.., PAIR>, default from 2^20 points; forced here at every size). h of
    random a, b (plain: one component; Rep3: two components + both masks) is bit-identical in the three forms: fused pair, two launches
    (tune ntt_pair = 0), and the reference's step-by-step sequence (tune h_unfused = 1) -- one-pass plans (2^2 .. 2^11), two passes, three."""
    F = H.FR["bn254"]
    n = 1 << logn
    gen = ntt.roots_of_unity(F)[1][logn]
    dom = gpu.Domain(0, logn, H.pack(F, [gen]))
    rs = np.random.RandomState(600 + logn)

    def limbs(k):
        v = rs.randint(0, 1 << 63, size=(k, 4), dtype=np.uint64)
        v[:, 3] >>= np.uint64(3)
        return v

    shift = H.pack(F, [ntt.roots_of_unity(F)[1][logn + 1]])
    for protocol, comp in ((0, 1), (1, 2)):
        a, b = limbs(n * comp), limbs(n * comp)
        mc, mab = (limbs(n), limbs(n)) if protocol == 1 else (None, None)
        outs = {}
        for name, knobs in (("pair", {"ntt_pair": 1, "ntt_pair_min_log": 0}), ("two_launches", {"ntt_pair": 0}), ("unfused", {"h_unfused": 1})):
            with gpu.tuned(**knobs):
                outs[name] = gpu.bindings.groth16_h(dom, shift, protocol, a, b, mc, mab)
        assert np.array_equal(outs["pair"], outs["two_launches"]), (logn, protocol)
        assert np.array_equal(outs["pair"], outs["unfused"]), (logn, protocol)
    dom.free()


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [10, 16])
def test_h_coset_table_kept_with_the_domain(gpu, logn):
    """The scaled coset table of the fused h pipeline stays with the domain after the first witness map (Domain::cs_table, tune
    h_table_cache). h must not depend on it: the first call (builds the table), a second call (reads the kept one), a call with ANOTHER
    shift on the same domain (not the kept one: computed into scratch), the first shift again, and a domain with the cache switched off
    all equal the reference's step-by-step sequence (h_unfused)."""
    F = H.FR["bn254"]
    n = 1 << logn
    gen = ntt.roots_of_unity(F)[1][logn]
    rs = np.random.RandomState(700 + logn)

    def limbs(k):
        v = rs.randint(0, 1 << 63, size=(k, 4), dtype=np.uint64)
        v[:, 3] >>= np.uint64(3)
        return v

    shift1 = H.pack(F, [ntt.roots_of_unity(F)[1][logn + 1]])
    shift2 = H.pack(F, [pow(5, 12345, F.p)])
    a, b = limbs(n), limbs(n)
    ref_dom = gpu.Domain(0, logn, H.pack(F, [gen]))
    with gpu.tuned(h_unfused=1):
        want1 = gpu.bindings.groth16_h(ref_dom, shift1, 0, a, b, None, None)
        want2 = gpu.bindings.groth16_h(ref_dom, shift2, 0, a, b, None, None)
    ref_dom.free()
    assert not np.array_equal(want1, want2)
    dom = gpu.Domain(0, logn, H.pack(F, [gen]))
    for shift, want in ((shift1, want1), (shift1, want1), (shift2, want2), (shift1, want1), (shift2, want2)):
        assert np.array_equal(gpu.bindings.groth16_h(dom, shift, 0, a, b, None, None), want)
    dom.free()
    dom = gpu.Domain(0, logn, H.pack(F, [gen]))
    with gpu.tuned(h_table_cache=0):
        assert np.array_equal(gpu.bindings.groth16_h(dom, shift2, 0, a, b, None, None), want2)
    assert np.array_equal(gpu.bindings.groth16_h(dom, shift1, 0, a, b, None, None), want1)   # first KEPT shift of this domain
    assert np.array_equal(gpu.bindings.groth16_h(dom, shift2, 0, a, b, None, None), want2)
    dom.free()


@pytest.mark.gpu
@pytest.mark.parametrize("curve,generator", [("bn254", 5), ("bls12_381", 7)])
def test_libsnark_proofs_plain_rep3_and_shamir_on_a_random_circuit(gpu, curve, generator):
    """plain_prove::<LibSnarkReduction> and the three-party Rep3 prove over an arkworks ProvingKey on the two north-star curves
    (cog16_prove_libsnark / _rep3: ark ProvingKey + Matrix blobs + wtns in, ark Proof out; the BLS12-377 counterpart on the reference's
    Penumbra circuit is tests/test_gpu_bls12_377.py): a random satisfied R1CS of 300 constraints, key from the restated arkworks generator
    with seeded toxic waste; the device's proof equals the restated prover's, the Rep3 proof equals the plain one, the pairing accepts it."""
    import ctypes as C
    import random
    import numpy as np
    from cosnarks_amd import groth16 as dev
    from oracle import arkfmt, cbridge as cb, curves as cv, fields as fl, groth16 as g16
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    G1, G2 = cv.CURVES[curve]
    rng = random.Random(2024 + cid)
    n_public, n_constraints = 3, 300
    A, B, Cm, w = g16.random_r1cs(F, rng, n_public, n_constraints)
    pub, wit = w[:n_public], w[n_public:]
    toxic = tuple(rng.randrange(1, F.p) for _ in range(5))
    fixed_base = lambda group, sc: cv.unpack_points((G1, G2)[group], cb.fixed_base_mul(cid, group, fl.pack(F, sc, mont=False)))
    key = g16.libsnark_setup(F, generator, G1, G2, A, B, Cm, n_public, len(wit), toxic, fixed_base)
    vk = {"alpha_g1": key["alpha_g1"], "beta_g2": key["beta_g2"], "gamma_g2": key["gamma_g2"], "delta_g2": key["delta_g2"], "ic": key["gamma_abc_g1"]}
    nb = G1.F.nbytes
    pk_bytes = arkfmt.ser_groth16_proving_key(key, G1.F.p, nb)
    a, b, c = (arkfmt.ser_matrix(M) for M in (A, B, Cm))
    wt = arkfmt.ser_wtns_positional(F.p, w)
    r, s = rng.randrange(F.p), rng.randrange(F.p)
    rl, sl = H.pack(F, [r], mont=False), H.pack(F, [s], mont=False)
    L = dev.glib()
    plen = 8 * nb
    plain, rep3 = (C.c_uint8 * 512)(), (C.c_uint8 * 512)()
    args = (cid, a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), wt, C.c_size_t(len(wt)), pk_bytes, C.c_size_t(len(pk_bytes)))
    assert L.cog16_prove_libsnark(*args, rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), plain, C.c_size_t(512), None, C.c_size_t(0)) == plen, L.cog16_last_error()
    got = arkfmt.parse_groth16_proof(bytes(plain[:plen]), G1.F.p, nb)
    want, _ = g16.prove_libsnark_plain(F, generator, G1, G2, key, A, B, Cm, pub, wit, r, s)
    assert got == want
    assert g16.verify(curve, G1, vk, got, pub[1:])
    assert L.cog16_prove_libsnark_rep3(*args, C.c_uint64(99), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), rep3, C.c_size_t(512),
                                       None, C.c_size_t(0)) == plen, L.cog16_last_error()
    assert bytes(rep3[:plen]) == bytes(plain[:plen])
    sham = (C.c_uint8 * 512)()
    for parties, thr in ((3, 1), (5, 2)):
        assert L.cog16_prove_libsnark_shamir(*args, parties, thr, C.c_uint64(5), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), sham,
                                             C.c_size_t(512)) == plen, L.cog16_last_error()
        assert bytes(sham[:plen]) == bytes(plain[:plen]), (parties, thr)
    assert L.cog16_prove_libsnark_shamir(*args, 4, 2, C.c_uint64(5), None, None, sham, C.c_size_t(512)) == -1     # n < 2t + 1
