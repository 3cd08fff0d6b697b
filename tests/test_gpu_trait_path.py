"""GPU parity of the zero-upstream-edit path: csh_groth16_witness_map_masks -- the fused witness map with the two Rep3
mask vectors handed over by the caller, the form rust/co-groth16-hip's HipCircomReduction calls ONCE per witness map -- and the host
mirror's "trait path" (host slices at every seam, five concurrent host-scalar MSMs), against the pinned oracle and the golden h / A / B / C
of the reference's own circuits. Bit-exact."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

from oracle import groth16 as og
from oracle import mpc, ntt
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


def _gold(curve, circ):
    return json.load(open(os.path.join(GOLD, "groth16_golden.json")))[f"{curve}/{circ}"]


def _setup(gpu, curve, circ):
    zk, wt, _vk, _pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    F = zko.Fr
    cid = H.CURVE_IDS[curve]
    w = [x % F.p for x in oz.parse_wtns(wt)]
    npub = zko.num_inputs
    A, B = zko.matrices()
    mk = lambda M: gpu.bindings.Matrix(cid, [[(H.pack(F, [c]), idx) for c, idx in row] for row in M])
    power = zko.domain_size.bit_length() - 1
    gen, shift = ntt.groth16_roots_of_unity(F, power)
    dom = gpu.Domain(cid, power, H.pack(F, [gen]))
    return zko, F, cid, w[:npub], w[npub:], mk(A), mk(B), dom, H.pack(F, [shift])


@pytest.mark.parametrize("curve,circ", CIRCUITS)
def test_witness_map_masks_plain_equals_golden_h(gpu, curve, circ):
    zko, F, cid, pub, wit, MA, MB, dom, shift = _setup(gpu, curve, circ)
    got = gpu.bindings.groth16_witness_map_masks(dom, shift, 0, 0, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack(F, wit))
    assert [str(x) for x in H.unpack(F, got)] == _gold(curve, circ)["h"]
    # masks are ignored by protocol 0 (plain / Shamir shares are multiplied without re-randomisation, shamir/arithmetic.rs:73-79)
    junk = H.pack(F, H.rand_elems(F, dom.n, H.rng(1)))
    again = gpu.bindings.groth16_witness_map_masks(dom, shift, 0, 0, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack(F, wit), junk, junk, fresh_output=False)
    assert np.array_equal(again, got)


@pytest.mark.parametrize("curve,circ", [("bn254", "multiplier2"), ("bn254", "poseidon"), ("bls12_381", "poseidon")])
def test_witness_map_masks_rep3_parties_match_oracle_and_cancel(gpu, curve, circ):
    """Every party's h equals the oracle's reduction.rs:77-193 run with the SAME two mask vectors in the reference's order (:160 then :182);
    with correlated masks (m_p = t_p - t_{p-1}, rngs.rs:103-106) the three h vectors sum to the plain golden h."""
    zko, F, cid, pub, wit, MA, MB, dom, shift = _setup(gpu, curve, circ)
    r = random.Random(9)
    n = dom.n
    shares = mpc.rep3_share_vec(F, wit, lambda: r.randrange(F.p))
    t = [[H.rand_elems(F, n, r) for _ in range(3)] for _ in range(2)]          # two draws (c, ab) of every party's stream
    masks = [[[(t[k][p][i] - t[k][(p + 2) % 3][i]) % F.p for i in range(n)] for k in range(2)] for p in range(3)]
    hs = []
    for p in range(3):
        got = gpu.bindings.groth16_witness_map_masks(dom, shift, 1, p, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack_shares(F, shares[p]),
                                                     H.pack(F, masks[p][0]), H.pack(F, masks[p][1]))
        hs.append(H.unpack(F, got))
        if dom.n <= 512:
            it = iter(masks[p])
            drv = og.Rep3Driver(F, p, mask_fn=lambda k: next(it))
            assert hs[-1] == og.witness_map_circom(zko, drv, pub, shares[p]), p
    assert [str((a + b + c) % F.p) for a, b, c in zip(*hs)] == _gold(curve, circ)["h"]
    assert hs[0] != hs[1]
    # swapping the two vectors is a different (wrong-order) draw: the result must change -- the order is part of the contract
    swapped = gpu.bindings.groth16_witness_map_masks(dom, shift, 1, 0, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack_shares(F, shares[0]),
                                                     H.pack(F, masks[0][1]), H.pack(F, masks[0][0]))
    assert H.unpack(F, swapped) != hs[0]


@pytest.mark.parametrize("curve,circ", [("bn254", "poseidon"), ("bls12_381", "multiplier2")])
def test_witness_map_masks_equals_the_seeded_entry_point(gpu, curve, circ):
    """The caller-mask form and the ChaCha12-seed form are the same map: masks drawn as csh_rep3_masks draws them (= Rep3Rand::
    masking_field_elements_vec, rngs.rs:137-156; chunks [off, off + n) then [off + n, off + 2n)) give bit-identical h."""
    zko, F, cid, pub, wit, MA, MB, dom, shift = _setup(gpu, curve, circ)
    r = random.Random(4)
    shares = mpc.rep3_share_vec(F, wit, lambda: r.randrange(F.p))
    n = dom.n
    keys = [bytes([7 * k + 3] * 32) for k in range(3)]
    off = 11
    for p in range(3):
        k1, k2 = keys[p], keys[(p + 2) % 3]
        seeded = gpu.bindings.groth16_witness_map_seeded(dom, shift, 1, p, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack_shares(F, shares[p]),
                                                         k1, off, k2, off)
        ms = []
        for j in range(2):
            out = np.zeros(4 * n, dtype=np.uint64)
            gpu.bindings._check(gpu.lib().csh_rep3_masks(cid, k1, C.c_uint64(off + j * n), k2, C.c_uint64(off + j * n), out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
            ms.append(out)
        got = gpu.bindings.groth16_witness_map_masks(dom, shift, 1, p, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack_shares(F, shares[p]), ms[0], ms[1])
        assert np.array_equal(got, seeded), p


def test_witness_map_masks_error_behaviour(gpu):
    zko, F, cid, pub, wit, MA, MB, dom, shift = _setup(gpu, "bn254", "multiplier2")
    r = random.Random(2)
    shares = mpc.rep3_share_vec(F, wit, lambda: r.randrange(F.p))
    m = H.pack(F, [1] * dom.n)
    args = (dom, shift, 1, 0, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack_shares(F, shares[0]))
    for mc, mab in ((None, None), (m, None), (None, m)):                      # Rep3 without both masks is refused (unmasked products leak)
        with pytest.raises(gpu.CoSnarksHipError, match="needs its masks"):
            gpu.bindings.groth16_witness_map_masks(*args, mc, mab)
    with pytest.raises(gpu.CoSnarksHipError, match="protocol must be"):
        gpu.bindings.groth16_witness_map_masks(dom, shift, 2, 0, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack(F, wit))
    with pytest.raises(gpu.CoSnarksHipError, match="column index exceeds"):     # a witness shorter than the matrices index
        gpu.bindings.groth16_witness_map_masks(dom, shift, 0, 0, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack(F, wit[:-1]))
    small = gpu.Domain(cid, 1, H.pack(F, [ntt.groth16_roots_of_unity(F, 1)[0]]))
    with pytest.raises(gpu.CoSnarksHipError, match="Polynomial Degree too large"):  # the message of reduction.rs:87-94
        gpu.bindings.groth16_witness_map_masks(small, shift, 0, 0, MA, MB, zko.num_constraints, H.pack(F, pub), H.pack(F, wit))


# ---- the host mirror driven the way the Rust shim drives the ABI ("trait path") --------------------------------------------------------
@pytest.mark.parametrize("curve,circ", CIRCUITS)
def test_trait_path_plain_prove_matches_golden(gpu, curve, circ):
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    with g.trait_path():
        proof, h = g.prove_plain(H.CURVE_IDS[curve], zk, wt, R, S, want_h=True, h_elems=zko.domain_size)
    gold = _gold(curve, circ)
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]
    assert [str(x) for x in H.unpack(zko.Fr, h)] == gold["h"]
    assert og.verify(curve, zko.G1, vk, oz.parse_proof(json.dumps(proof)), pub)


@pytest.mark.parametrize("curve,circ", [("bn254", "multiplier2"), ("bn254", "poseidon"), ("bls12_381", "poseidon")])
def test_trait_path_rep3_and_shamir_match_golden_and_the_device_resident_path(gpu, curve, circ):
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    F = zko.Fr
    cid = H.CURVE_IDS[curve]
    gold = _gold(curve, circ)
    ref_proof, ref_hs = g.prove_rep3(cid, zk, wt, seed=42, r=R, s=S, want_h=True, h_elems=zko.domain_size)
    with g.trait_path():
        proof, hs = g.prove_rep3(cid, zk, wt, seed=42, r=R, s=S, want_h=True, h_elems=zko.domain_size)
        sh = g.prove_shamir(cid, zk, wt, 3, 1, seed=11, r=R, s=S)
        fresh, _ = g.prove_rep3(cid, zk, wt, seed=7)
    for p in (proof, sh):
        assert p["pi_a"][:2] == gold["a"] and p["pi_b"][:2] == gold["b"] and p["pi_c"][:2] == gold["c"]
    n = zko.domain_size
    parts = [H.unpack(F, hs[4 * n * p:4 * n * (p + 1)]) for p in range(3)]
    assert [str((a + b + c) % F.p) for a, b, c in zip(*parts)] == gold["h"]
    # host-drawn masks (masking_field_elements_vec twice) consume the party's streams exactly as the device generator's run of 2n
    # chunks does: the h SHARES, not only their sum, are those of the device-resident prove with the same seeds
    assert np.array_equal(np.asarray(hs), np.asarray(ref_hs)) and proof == ref_proof
    assert og.verify(curve, zko.G1, vk, oz.parse_proof(json.dumps(fresh)), pub)


@pytest.mark.parametrize("curve,circ", [("bn254", "multiplier2"), ("bn254", "poseidon"), ("bls12_381", "poseidon")])
def test_trait_path_rep3_seeded_device_masks_cancel_and_match_golden(gpu, curve, circ):
    """The shim's opt-in all-GPU-parties mode (VERDICT r4 #1c): every party draws ONE pair of fresh seeds through the public
    Rep3Rand::random_seeds() (rngs.rs:233) per witness map and the device generates both mask vectors from them. Party i's rng1 is party
    i+1's rng2, so the seeds -- hence the masks -- stay pairwise correlated: the three h shares sum to the plain golden h and the agreed
    proof is the golden proof. The shares themselves differ from the host-mask mode's (other keystreams): the mode is a session-wide choice."""
    from cosnarks_amd import groth16 as g
    zk, wt, vk, pub = _load(curve, circ)
    zko = oz.parse_zkey(zk)
    F = zko.Fr
    cid = H.CURVE_IDS[curve]
    gold = _gold(curve, circ)
    with g.trait_path():
        _, hs_host = g.prove_rep3(cid, zk, wt, seed=42, r=R, s=S, want_h=True, h_elems=zko.domain_size)
    with g.trait_path(2):
        proof, hs = g.prove_rep3(cid, zk, wt, seed=42, r=R, s=S, want_h=True, h_elems=zko.domain_size)
        again, hs2 = g.prove_rep3(cid, zk, wt, seed=42, r=R, s=S, want_h=True, h_elems=zko.domain_size)
        plain, _ = g.prove_plain(cid, zk, wt, R, S)                  # protocols without masks are untouched by the mode
        fresh, _ = g.prove_rep3(cid, zk, wt, seed=5)
    assert proof["pi_a"][:2] == gold["a"] and proof["pi_b"][:2] == gold["b"] and proof["pi_c"][:2] == gold["c"]
    assert plain["pi_a"][:2] == gold["a"] and plain["pi_c"][:2] == gold["c"]
    n = zko.domain_size
    parts = [H.unpack(F, hs[4 * n * p:4 * n * (p + 1)]) for p in range(3)]
    assert [str((a + b + c) % F.p) for a, b, c in zip(*parts)] == gold["h"]
    assert parts[0] != parts[1]
    assert not np.array_equal(np.asarray(hs), np.asarray(hs_host))   # other masks than the host draw's
    assert np.array_equal(np.asarray(hs), np.asarray(hs2)) and again == proof   # deterministic in the parties' seeds
    assert og.verify(curve, zko.G1, vk, oz.parse_proof(json.dumps(fresh)), pub)


def test_trait_path_synthetic_circuit_closed_form(gpu):
    """The synthetic 2^14 circuit with a known-dlog key through the trait path: the closed-form check of bench.py's prove line."""
    from cosnarks_amd import groth16 as g
    with g.trait_path():
        c = g.SynthCircuit(0, 14)
        ph = c.prove()
        assert c.check()
        c.close()
    assert ph["msm_groups"] > 0 and ph["witness_upload_and_map"] > 0


def test_large_uploads_direct_and_staged_agree(gpu):
    """Round 5: uploads of >= 4 MiB from caller memory go up in one copy (tune host_h2d = 0), staged through the lane's page-locked
    buffers in 2 MiB chunks moved by host threads (1), or direct-and-timed (2, the default: staged for a while after two stalled uploads).
    A host-pointer transform of 2^18 elements (8 MiB up, 8 MiB down), a host-scalar MSM of 2^17 + 3 points (shared-upload path) and a
    trait-path prove of a 2^17-constraint circuit give the same bytes / the closed form in every mode; mode 1 is seen to stage."""
    import ctypes as C
    from cosnarks_amd import bindings as B
    from cosnarks_amd import groth16 as g
    from tests.check_closed_form import closed_form_point
    from tests.test_gpu_msm import _gen_bases
    F = H.FR["bn254"]
    logn = 18
    gen = ntt.roots_of_unity(F)[1][logn]
    dom = gpu.Domain(H.CURVE_IDS["bn254"], logn, H.pack(F, [gen]))
    x = np.random.RandomState(5).randint(0, 1 << 62, size=(1 << logn, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(2)
    n = (1 << 17) + 3
    buf = _gen_bases(gpu, "bn254", 0, 0x77, n)
    hb = C.c_void_p()
    gpu.bindings._check(gpu.lib().csh_bases_upload_dev(0, 0, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(hb)))
    buf.free()
    limbs = np.random.RandomState(6).randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    want_pt = closed_form_point("bn254", 0, 0x77, n, limbs, True)
    G = cv_mod().BN254_G1
    outs = {}
    for mode in (0, 1, 2):
        staged0 = B.tune_get("stat_h2d_staged")
        with gpu.tuned(host_h2d=mode):
            outs[mode] = np.array(dom.ifft_in_to_out(x.copy()), copy=True)
            o = np.zeros(12, dtype=np.uint64)
            gpu.bindings._check(gpu.lib().csh_msm(hb, C.c_size_t(0), C.c_size_t(n), limbs.ctypes.data_as(C.c_void_p), 1, o.ctypes.data_as(C.c_void_p)))
            assert G.eq(H.jac_to_affine(G, o), want_pt), mode
            with g.trait_path():
                c = g.SynthCircuit(0, 17)
                c.prove()
                assert c.check(), mode
                c.close()
        if mode != 2:
            assert (B.tune_get("stat_h2d_staged") > staged0) == (mode == 1), mode
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    gpu.lib().csh_bases_free(hb)
    dom.free()


def cv_mod():
    from oracle import curves
    return curves


def test_large_results_direct_and_staged_copies_agree(gpu):
    """Results of >= 4 MiB reach the caller's pageable memory either by one copy into its pages or staged through the lane's page-locked
    buffer (tune host_d2h: 0 / 1; 2 = direct, timed, staged after a stalled copy): a 2^18-point host-pointer transform and the h of a
    2^17-constraint witness map (inside a trait-path prove, closed-form checked) are the same bytes either way."""
    from cosnarks_amd import bindings as B
    from cosnarks_amd import groth16 as g
    F = H.FR["bn254"]
    logn = 18
    gen = ntt.roots_of_unity(F)[1][logn]
    dom = gpu.Domain(H.CURVE_IDS["bn254"], logn, H.pack(F, [gen]))
    x = np.random.RandomState(3).randint(0, 1 << 62, size=(1 << logn, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(2)
    outs = {}
    for mode in (0, 1, 2):
        staged0 = B.tune_get("stat_d2h_staged")
        with gpu.tuned(host_d2h=mode):
            outs[mode] = np.array(dom.ifft_in_to_out(x.copy()), copy=True)
            with g.trait_path():
                c = g.SynthCircuit(0, 17)
                c.prove()
                assert c.check(), mode
                c.close()
        if mode != 2:  # 2 stages only after stalled copies, which depends on the host
            assert (B.tune_get("stat_d2h_staged") > staged0) == (mode == 1), mode
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.array_equal(np.asarray(dom.fft_out_to_in(outs[1].copy())).reshape(-1), x.reshape(-1))
    dom.free()


def test_concurrent_host_pointer_callers_while_the_copy_mode_flips(gpu):
    """the host result path under the load it is built for -- ten host threads (rayon workers in the reference) calling
    host-pointer entry points of mixed sizes at once (transforms of 2^12 .. 2^20 elements: below and above the 4 MiB threshold of the
    page population / staged copy; witness maps of a 2^17-constraint circuit inside trait-path proves) while another thread flips
    tune host_d2h between direct (0), staged (1) and auto (2) and host_populate on and off. Every result is compared bit for bit with
    the one computed alone beforehand; the staged path must actually have run."""
    import threading
    import time
    from cosnarks_amd import bindings as B
    from cosnarks_amd import groth16 as g
    F = H.FR["bn254"]
    cid = H.CURVE_IDS["bn254"]
    sizes = [12, 16, 17, 18, 19, 20, 18, 20]
    doms, xs, want = {}, {}, {}
    for logn in sorted(set(sizes)):
        gen = ntt.roots_of_unity(F)[1][logn]
        doms[logn] = gpu.Domain(cid, logn, H.pack(F, [gen]))
        x = np.random.RandomState(50 + logn).randint(0, 1 << 62, size=(1 << logn, 4), dtype=np.uint64)
        x[:, 3] >>= np.uint64(2)
        xs[logn] = x
        want[logn] = np.array(doms[logn].ifft_in_to_out(x.copy()), copy=True)
    staged0 = B.tune_get("stat_d2h_staged")
    stop = threading.Event()
    errs = []

    def transformer(logn, rounds):
        try:
            gpu.bindings._check(gpu.lib().csh_init(0))
            for _ in range(rounds):
                got = doms[logn].ifft_in_to_out(xs[logn].copy())                 # a fresh destination every time, as a caller's new Vec is
                if not np.array_equal(np.asarray(got).reshape(-1), want[logn].reshape(-1)):
                    errs.append(("ifft", logn))
                    return
        except Exception as e:  # noqa: BLE001
            errs.append(("ifft", logn, repr(e)))

    def prover(rounds):
        try:
            gpu.bindings._check(gpu.lib().csh_init(0))
            with g.trait_path():
                c = g.SynthCircuit(0, 17)
                for _ in range(rounds):
                    c.prove()
                    if not c.check():
                        errs.append(("prove",))
                        break
                c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(("prove", repr(e)))

    def flipper():
        k = 0
        while not stop.is_set():
            B.tune_set("host_d2h", (1, 0, 2, 1)[k % 4])
            B.tune_set("host_populate", (0x101, 0, 0x101, 0x102)[k % 4])
            B.tune_set("host_h2d", (1, 2, 0, 1, 1)[k % 5])                      # uploads from caller memory: staged / auto / direct
            k += 1
            time.sleep(0.003)

    d2h0, pop0, h2d0 = B.tune_get("host_d2h"), B.tune_get("host_populate"), B.tune_get("host_h2d")
    up_staged0 = B.tune_get("stat_h2d_staged")
    th = [threading.Thread(target=transformer, args=(logn, 12 if logn >= 19 else 30)) for logn in sizes]
    th += [threading.Thread(target=prover, args=(4,)) for _ in range(2)]
    fl = threading.Thread(target=flipper)
    fl.start()
    try:
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
    finally:
        stop.set()
        fl.join()
        B.tune_set("host_d2h", d2h0)
        B.tune_set("host_populate", pop0)
        B.tune_set("host_h2d", h2d0)
    assert not errs, errs
    assert not any(t.is_alive() for t in th)
    assert B.tune_get("stat_d2h_staged") > staged0 and B.tune_get("stat_h2d_staged") > up_staged0
    for d in doms.values():
        d.free()



def test_large_host_transfers_go_through_a_bounded_staging_ring(gpu):
    """staged transfers used to page-lock a buffer of the FULL transfer size per (lane, stream, slot) for the life of
    the process. Round 6: at most 32 MiB per slot -- larger uploads and results cycle through a ring of 2 MiB chunks. Host-pointer vec_add
    (two uploads + one staged result) at sizes around the ring boundaries and well beyond is bit-identical to the CPU restatement, and the
    page-locked memory the library holds afterwards stays within a few ring sizes."""
    from oracle import cbridge
    B = gpu.bindings
    ring = 32 << 20
    before = B.tune_get("stat_pinned_kib")
    rs = np.random.RandomState(99)
    for nbytes in (ring - 32, ring, ring + 32, ring + (16 << 20) + 96, 3 * ring + (2 << 20) + 32, 200 << 20):
        n = nbytes // 32
        a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        b = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        got = gpu.vec_add(0, a, b)
        assert np.array_equal(np.asarray(got).reshape(-1), cbridge.vec_add(0, a, b).reshape(-1)), nbytes
    held = (B.tune_get("stat_pinned_kib") - before) << 10
    assert held <= 6 * ring, held          # uploads use two slots of this lane, the result one: nowhere near the 600 MB of the last case
    assert B.tune_get("stat_h2d_staged") > 0 and B.tune_get("stat_d2h_staged") > 0
