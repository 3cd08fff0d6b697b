"""GPU parity (through the C ABI): MSM over BLS12-377 G1 / G2 -- the curve of the reference's LibSnarkReduction fixtures
(co-circom/co-groth16/src/lib.rs:231-300: Groth16::<Bls12_377>::plain_prove::<LibSnarkReduction> on the Penumbra circuits). The base
field runs in the 14 x 28-bit signed lazy limbs of BLS12-381's; Fq2 = Fq[u]/(u^2 + 5) (field29.hpp Fp2S / curve_pair.hpp Fp2Pair with
NR = 5). Bit-exact on the affine result against the Python oracle, oracle/c and the closed form; the reference-held points of
circuit.vk go through the device too."""
import ctypes as C
import gzip
import os
import struct

import numpy as np
import pytest

from oracle import curves as cv
from tests import helpers as H

pytestmark = pytest.mark.gpu
CURVE = "bls12_377"
CID = H.CURVE_IDS[CURVE]
F = H.FR[CURVE]


def _run(gpu, group, pts, scalars, montgomery=True, offset=0, n=None):
    G = cv.CURVES[CURVE][group]
    bases = gpu.Bases(CID, group, cv.pack_points(G, pts))
    out = bases.msm(H.pack(F, scalars, mont=montgomery), offset=offset, n=n, montgomery=montgomery)
    bases.free()
    return H.jac_to_affine(G, out)


@pytest.mark.parametrize("group", [0, 1])
def test_msm_small_and_edge_cases_match_oracle(gpu, group):
    """msm_unchecked / msm_bigint semantics at n = 0 .. 257, zero / one / r - 1 / small scalars, duplicates, P and -P, infinity
    bases, offset slices (the suites of tests/test_gpu_msm.py on the third pairing curve)."""
    G = cv.CURVES[CURVE][group]
    r = H.rng(3770 + group)
    for n in (0, 1, 2, 3, 33, 257):
        pts = H.rand_points(G, n, r)
        sc = H.rand_elems(F, n, r)
        want = G.msm(pts, sc)
        assert G.eq(_run(gpu, group, pts, sc), want), n
        assert G.eq(_run(gpu, group, pts, sc, montgomery=False), want), n
    n = 64
    pts = H.rand_points(G, n, r, with_inf=True)
    pts[10] = pts[11]
    pts[20] = G.neg(pts[21])
    suites = {
        "zeros": [0] * n,
        "ones": [1] * n,
        "r-1": [F.p - 1] * n,
        "small64": [r.randrange(1 << 64) for _ in range(n)],
        "mixed": [0, 1, F.p - 1, 2, (F.p - 1) // 2, (F.p + 1) // 2] * 10 + [5, 6, 7, 8],
        "equal_cancel": [7] * n,
    }
    for name, sc in suites.items():
        assert G.eq(_run(gpu, group, pts, sc), G.msm(pts, sc)), name
    assert _run(gpu, group, [None] * 8, [3] * 8) is None
    sc = H.rand_elems(F, n, r)
    assert G.eq(_run(gpu, group, pts, sc[:20], offset=7, n=20), G.msm(pts[7:27], sc[:20]))


def _gen_bases(gpu, group, seed, n):
    buf = gpu.DeviceBuffer(n * gpu.point_bytes(CID, group))
    gpu.bindings._check(gpu.lib().csh_util_generate_bases_dev(CID, group, C.c_uint64(seed), C.c_size_t(n), buf.ptr, None))
    return buf


@pytest.mark.parametrize("group", [0, 1])
def test_generated_bases_have_known_dlog_and_equal_the_cpu_restatement(gpu, group):
    """Also the regression test of the small-copy race: the G2 generator kernel (64 double-and-add steps in Fq2) outlives the
    call that queued it on the caller's lane stream; the 57 KB csh_memcpy_d2h behind to_host() must wait for it."""
    from oracle import cbridge as cb
    from tests.check_closed_form import dlogs
    G = cv.CURVES[CURVE][group]
    n = 300
    buf = _gen_bases(gpu, group, 99, n)
    host = buf.to_host()
    buf.free()
    pts = cv.unpack_points(G, host)
    for P, k in list(zip(pts, dlogs(99, n)))[:4]:
        assert G.eq(P, G.mul(G.gen, int(k)))
    assert (np.asarray(host).view(np.uint64).reshape(-1) == cb.generate_bases(CID, group, 99, n).reshape(-1)).all()


@pytest.mark.parametrize("group,logn,tables", [(0, 17, 0), (0, 18, -1), (1, 15, 0), (1, 16, -1)])
def test_msm_closed_form_plain_and_policy_tables(gpu, group, logn, tables):
    """Known-dlog bases: MSM == (sum s_i k_i) G on the plain handle and with the library's table policy (ONE bucket set, 17-bit
    windows, msm_sort_wide.hip); uniform scalars in Montgomery form and as canonical integers."""
    from tests.check_closed_form import closed_form_point
    G = cv.CURVES[CURVE][group]
    n = 1 << logn
    seed = 0x377 + logn
    buf = _gen_bases(gpu, group, seed, n)
    h = C.c_void_p()
    gpu.bindings._check(gpu.lib().csh_bases_upload_dev(CID, group, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
    buf.free()
    if tables:
        c, rows = C.c_int(0), C.c_int(0)
        gpu.bindings._check(gpu.lib().csh_bases_table_policy(C.c_size_t(n), C.byref(c), C.byref(rows)))
        assert c.value >= 17 and rows.value > 0
        gpu.bindings._check(gpu.lib().csh_bases_precompute_grouped(h, c.value, rows.value))
    rs = np.random.RandomState(logn)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)                                  # < 2^252 < r
    out = np.zeros(3 * gpu.point_bytes(CID, group) // 16, dtype=np.uint64)
    for mont in (1, 0):
        gpu.bindings._check(gpu.lib().csh_msm(h, C.c_size_t(0), C.c_size_t(n), limbs.ctypes.data_as(C.c_void_p), mont, out.ctypes.data_as(C.c_void_p)))
        if tables:
            assert gpu.bindings.msm_last_params()[1] == 1, "one bucket set"
        assert G.eq(H.jac_to_affine(G, out), closed_form_point(CURVE, group, seed, n, limbs, bool(mont))), mont
    gpu.lib().csh_bases_free(h)


@pytest.mark.parametrize("group,logn", [(0, 16), (1, 13)])
def test_msm_full_range_points_equal_the_cpu_restatement(gpu, group, logn):
    """Bases with full-width discrete logs (oracle/c: generate_bases_wide), uniform scalars: the device's affine result is bit-identical
    to oracle/c's Pippenger (64-bit __int128 limbs, Jacobian buckets: a different arithmetic from the device's)."""
    from oracle import cbridge as cb
    G = cv.CURVES[CURVE][group]
    n = 1 << logn
    pts = cb.generate_bases_wide(CID, group, 0x377, n)
    rs = np.random.RandomState(7 + logn)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    want = cv.unpack_points(G, cb.msm_fast(CID, group, pts, limbs))[0]
    bases = gpu.Bases(CID, group, pts)
    assert G.eq(H.jac_to_affine(G, bases.msm(limbs)), want)
    bases.precompute(17, 0)
    assert G.eq(H.jac_to_affine(G, bases.msm(limbs)), want)
    bases.free()


@pytest.mark.parametrize("group", [0, 1])
def test_msm_fixed_base_tables_every_row_layout(gpu, group):
    G = cv.CURVES[CURVE][group]
    r = H.rng(1377 + group)
    n = 1100
    pts = H.rand_points(G, n, r, with_inf=True)
    sc = H.rand_elems(F, n, r)
    sk = [1] * 300 + [F.p - 1] * 300 + [0] * 100 + H.rand_elems(F, n - 700, r)
    want_full, want_sk = G.msm(pts, sc), G.msm(pts, sk)
    want_off = G.msm(pts[37:37 + 900], sc[:900])
    for c, groups in ((0, 0), (15, 2), (11, 5), (20, 0)):
        bases = gpu.Bases(CID, group, cv.pack_points(G, pts)).precompute(c, groups)
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc))), want_full), c
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sk))), want_sk), c
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc[:900]), offset=37, n=900)), want_off), c
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc, mont=False), montgomery=False)), want_full), c
        bases.free()


def test_reference_vk_points_through_the_device(gpu):
    """The G1 / G2 points the reference commits in test_vectors/Groth16/bls12_377/penumbra_output/circuit.vk (tests/golden copy):
    vk_x = gamma_abc[0] + sum pub_i gamma_abc[i] (what Groth16::verify folds, co-circom/co-groth16/src/lib.rs:287) as a device MSM ==
    the oracle's; and beta / gamma / delta (G2) under random scalars."""
    from oracle import arkfmt
    G1, G2 = cv.CURVES[CURVE]
    q = G1.F.p
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bls12_377", "penumbra_output")
    vk = gzip.open(os.path.join(d, "circuit.vk.gz"), "rb").read()
    alpha, off = arkfmt.parse_g1(vk, 0, q, 48)
    g2 = []
    for _ in range(3):
        pt, off = arkfmt.parse_g2(vk, off, q, 48)
        g2.append(pt)
    (k,) = struct.unpack_from("<Q", vk, off)
    off += 8
    abc = []
    for _ in range(k):
        pt, off = arkfmt.parse_g1(vk, off, q, 48)
        abc.append(pt)
    assert all(G2.is_on_curve(P) and G2.mul(P, G2.order) is None for P in g2)
    r = H.rng(377)
    pub = [1] + H.rand_elems(F, k - 1, r)
    assert G1.eq(_run(gpu, 0, abc, pub), G1.msm(abc, pub))
    sc = H.rand_elems(F, 3, r)
    assert G2.eq(_run(gpu, 1, g2, sc), G2.msm(g2, sc))
    assert G1.eq(_run(gpu, 0, [alpha] + abc, [F.p - 1] + pub), G1.msm([alpha] + abc, [F.p - 1] + pub))


@pytest.mark.parametrize("group", [0, 1])
def test_split_msm_ranges_fold_to_the_whole(gpu, group):
    """csh_msm_partial over 4 contiguous ranges + csh_msm_fold_partials (the exchange format of the split MSM, msm_split.hip) == one MSM."""
    G = cv.CURVES[CURVE][group]
    r = H.rng(2377 + group)
    n = 2000
    pts = H.rand_points(G, n, r)
    sc = H.rand_elems(F, n, r)
    bases = gpu.Bases(CID, group, cv.pack_points(G, pts))
    dsc = gpu.DeviceBuffer.from_host(H.pack(F, sc))
    pb = gpu.msm_partial_bytes(CID, group)
    host = np.zeros(4 * pb, dtype=np.uint8)
    bounds = [0, 300, 1000, 1000, n]                             # one empty range
    for i in range(4):
        lo, hi = bounds[i], bounds[i + 1]
        out = gpu.DeviceBuffer(pb)
        bases.msm_partial_dev(C.c_void_p(dsc.ptr.value + 32 * lo), hi - lo, out, offset=lo)
        host[pb * i:pb * (i + 1)] = out.to_host(np.uint8)
        out.free()
    got = gpu.msm_fold_partials(CID, group, host, 4)
    assert G.eq(H.jac_to_affine(G, got), G.msm(pts, sc))
    dsc.free()
    bases.free()


def test_libsnark_proof_on_the_reference_penumbra_circuit(gpu):
    """proof_libsnark_penumbra_output_bls12_377 (co-circom/co-groth16/src/lib.rs:231-290, 297-299) through the host mirror and the device:
    ark ProvingKey + a / b / c.bin + witness.wtns in (cog16_prove_libsnark reads the reference's formats itself), LibSnarkReduction's h and
    the five query MSMs (G1 and G2 of BLS12-377) on the GPU, ark Proof out -- bit-identical to the oracle's restated prover and accepted by
    the oracle's pairing check under the key's vk. (The key comes from the restated arkworks generator: the reference's circuit.pk is
    absent upstream.)"""
    import hashlib
    from cosnarks_amd import groth16 as dev
    from oracle import arkfmt, groth16 as g16
    (F_, A, B, Cm, pub, wit, exp), key, vk, pk_bytes, msm = H.penumbra_libsnark_key()
    G1, G2 = cv.CURVES[CURVE]
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bls12_377", "penumbra_output")
    rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
    a, b, c, w = rd("a.bin"), rd("b.bin"), rd("c.bin"), rd("witness.wtns")
    r, s = 0x1234567890ABCDEF % F.p, (F.p - 5)
    rl, sl = H.pack(F, [r], mont=False), H.pack(F, [s], mont=False)
    out = (C.c_uint8 * 512)()
    hbuf = np.zeros(exp["domain_size"] * 4, dtype=np.uint64)
    L = dev.glib()
    n = L.cog16_prove_libsnark(CID, a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), w, C.c_size_t(len(w)), pk_bytes,
                               C.c_size_t(len(pk_bytes)), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), out, C.c_size_t(512),
                               hbuf.ctypes.data_as(C.c_void_p), C.c_size_t(exp["domain_size"]))
    assert n == 96 + 192 + 96, L.cog16_last_error()
    got = arkfmt.parse_groth16_proof(bytes(out[:n]), G1.F.p, 48)
    hh = H.unpack(F, hbuf)
    assert hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in hh)).hexdigest() == exp["h_sha256"]
    want, _ = g16.prove_libsnark_plain(F, exp["generator"], G1, G2, key, A, B, Cm, pub, wit, r, s, msm=msm, h=hh)
    assert got == want
    assert g16.verify(CURVE, G1, vk, got, pub[1:])
    # the same files under another curve id are rejected (48-byte coordinates read as 32-byte ones: non-canonical values / lengths off)
    assert L.cog16_prove_libsnark(0, a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), w, C.c_size_t(len(w)), pk_bytes,
                                  C.c_size_t(len(pk_bytes)), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), out, C.c_size_t(512),
                                  None, C.c_size_t(0)) == -1
    # a witness of the wrong length is an error, as in prove_inner (groth16.rs:131-146)
    w_short = arkfmt.ser_wtns_positional(F.p, (pub + wit)[:-3])
    assert L.cog16_prove_libsnark(CID, a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), w_short, C.c_size_t(len(w_short)), pk_bytes,
                                  C.c_size_t(len(pk_bytes)), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), out, C.c_size_t(512),
                                  None, C.c_size_t(0)) == -1
    assert b"private witness" in L.cog16_last_error()
    # a key cut short is rejected, not mis-parsed
    assert L.cog16_prove_libsnark(CID, a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), w, C.c_size_t(len(w)), pk_bytes,
                                  C.c_size_t(len(pk_bytes) - 7), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), out, C.c_size_t(512),
                                  None, C.c_size_t(0)) == -1


def test_rep3_and_shamir_libsnark_proofs_equal_the_plain_proof(gpu):
    """Rep3CoGroth16::prove::<LibSnarkReduction> (co-circom/co-groth16/src/groth16.rs:360-379 with the LibSnark reduction) on the reference's
    Penumbra circuit, three in-process parties as in the reference's Rep3 tests (tests/tests/circom/e2e_tests/rep3.rs:57-69): witness shared
    from a seed, LibSnark witness map with device masks and the five query MSMs per party on BLS12-377, the parties' proofs agree, equal
    the plain proof for the same r, s bit for bit, and verify by pairing; the three h half-share vectors sum to the plain h."""
    import hashlib
    from cosnarks_amd import groth16 as dev
    from oracle import arkfmt, groth16 as g16
    (F_, A, B, Cm, pub, wit, exp), key, vk, pk_bytes, msm = H.penumbra_libsnark_key()
    G1, G2 = cv.CURVES[CURVE]
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bls12_377", "penumbra_output")
    rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
    a, b, c, w = rd("a.bin"), rd("b.bin"), rd("c.bin"), rd("witness.wtns")
    r, s = 0x0123456789ABCDEF0123 % F.p, (F.p - 77)
    rl, sl = H.pack(F, [r], mont=False), H.pack(F, [s], mont=False)
    L = dev.glib()
    n_dom = exp["domain_size"]
    plain, rep3 = (C.c_uint8 * 512)(), (C.c_uint8 * 512)()
    hs = np.zeros(3 * n_dom * 4, dtype=np.uint64)
    args = (CID, a, C.c_size_t(len(a)), b, C.c_size_t(len(b)), c, C.c_size_t(len(c)), w, C.c_size_t(len(w)), pk_bytes, C.c_size_t(len(pk_bytes)))
    n0 = L.cog16_prove_libsnark(*args, rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), plain, C.c_size_t(512), None, C.c_size_t(0))
    assert n0 == 384, L.cog16_last_error()
    n1 = L.cog16_prove_libsnark_rep3(*args, C.c_uint64(4242), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), rep3, C.c_size_t(512),
                                     hs.ctypes.data_as(C.c_void_p), C.c_size_t(3 * n_dom))
    assert n1 == 384, L.cog16_last_error()
    assert bytes(rep3[:384]) == bytes(plain[:384])
    assert g16.verify(CURVE, G1, vk, arkfmt.parse_groth16_proof(bytes(rep3[:384]), G1.F.p, 48), pub[1:])
    parts = [H.unpack(F, hs[4 * n_dom * p:4 * n_dom * (p + 1)]) for p in range(3)]
    h = [(x + y + z) % F.p for x, y, z in zip(*parts)]
    assert hashlib.sha256(b"".join(x.to_bytes(32, "little") for x in h)).hexdigest() == exp["h_sha256"]
    assert parts[0] != h
    # ShamirCoGroth16::prove with the LibSnark reduction (groth16.rs:439-463), 3 parties / threshold 1: the same proof again
    sham = (C.c_uint8 * 512)()
    n2 = L.cog16_prove_libsnark_shamir(*args, 3, 1, C.c_uint64(777), rl.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), sham, C.c_size_t(512))
    assert n2 == 384, L.cog16_last_error()
    assert bytes(sham[:384]) == bytes(plain[:384])
