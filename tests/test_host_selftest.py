"""CPU tests: the host-executed instantiations of the kernels' field/curve templates vs the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import curves as cv
from oracle import fields as fl
from tests import helpers as H

FIELDS = [fl.BN254_FQ, fl.BN254_FR, fl.BLS381_FQ, fl.BLS381_FR, fl.BLS377_FQ, fl.BLS377_FR]


def _fop(hip, F, op, a, b=None):
    L = hip.lib()
    pa = H.pack(F, [a], mont=False)
    pb = H.pack(F, [b], mont=False) if b is not None else None
    out = np.zeros(F.nlimbs, dtype=np.uint64)
    rc = L.csh_selftest_field_op(H.FIELD_IDS[F.name], op, pa.ctypes.data_as(C.c_void_p),
                                 pb.ctypes.data_as(C.c_void_p) if pb is not None else None, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return fl.limbs_to_int(out)


@pytest.mark.parametrize("F", FIELDS, ids=lambda f: f.name)
def test_field_ops_match_oracle(hip, F):
    r = H.rng(11)
    vals = H.edge_elems(F) + H.rand_elems(F, 40, r)
    vals = [v % F.p for v in vals]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        am, bm = F.to_mont(a), F.to_mont(b)
        assert _fop(hip, F, 0, am, bm) == F.to_mont((a + b) % F.p)
        assert _fop(hip, F, 1, am, bm) == F.to_mont((a - b) % F.p)
        assert _fop(hip, F, 2, am, bm) == F.to_mont(a * b % F.p)
        assert _fop(hip, F, 7, am) == F.to_mont(a * a % F.p)
        assert _fop(hip, F, 6, am) == F.to_mont(-a % F.p)
        assert _fop(hip, F, 4, am) == a          # from_mont
        assert _fop(hip, F, 5, a) == am          # to_mont
    for a in vals[:12]:
        if a:
            assert _fop(hip, F, 3, F.to_mont(a)) == F.to_mont(pow(a, -1, F.p))


def test_bn254_fr_mul_known_answer(hip):
    """The four products hard-coded by the reference: tests/tests/mpc/rep3.rs:286-345."""
    F = fl.BN254_FR
    x = [13839525561076761625780930844889299788193703994911163378019280196128582690055,
         19302971480864839163158232064620707211435225928426123775531639309944891593977,
         8048717310762513532550620831072439583505607813129662608591015555880153427210,
         2585271390974436123003027749932103593962191064365118925254473311197989280023]
    y = [2688648969035332064113669477511029957484512453056743431884706385750388613065,
         13632770404954969699480437686769008635735921498648460325387842712839596176806,
         19199593902803943133889170931116903997086625101975591190159463567024116566625,
         8255472466884305547009533395117607586789669747151273739964395707537515634749]
    z = [14012338922664984944451142760937475581748095944353358534203030914664561190462,
         4297594441150501195973997511775989720904927516253689527653694984160382713321,
         7875903949174289914141782934879682497141865775307179984684659764891697566272,
         6646526994769136778802685410292764833027657364709823469005920616147071273574]
    for a, b, c in zip(x, y, z):
        assert a * b % F.p == c                                   # oracle arithmetic
        assert _fop(hip, F, 2, F.to_mont(a), F.to_mont(b)) == F.to_mont(c)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_fp2_ops(hip, curve):
    G2 = cv.CURVES[curve][1]
    F2, Fq = G2.F, G2.F.base
    L = hip.lib()
    r = H.rng(5)
    for _ in range(20):
        a = (r.randrange(Fq.p), r.randrange(Fq.p))
        b = (r.randrange(Fq.p), r.randrange(Fq.p))
        pa, pb = H.pack(Fq, list(a)), H.pack(Fq, list(b))
        for op, want in [(0, F2.add(a, b)), (1, F2.sub(a, b)), (2, F2.mul(a, b)), (7, F2.sqr(a)), (3, F2.inv(a)), (6, F2.neg(a))]:
            out = np.zeros(2 * Fq.nlimbs, dtype=np.uint64)
            assert L.csh_selftest_fp2_op(H.CURVE_IDS[curve], op, pa.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p)) == 0
            assert tuple(H.unpack(Fq, out)) == want


def _xyzz_to_affine(hip, curve_id, group, curve, xyzz):
    out = np.zeros(hip.point_bytes(curve_id, group) // 8, dtype=np.uint64)
    assert hip.lib().csh_selftest_curve_op(curve_id, group, 4, xyzz.ctypes.data_as(C.c_void_p), None, 0, out.ctypes.data_as(C.c_void_p)) == 0
    return cv.unpack_points(curve, out)[0]


@pytest.mark.parametrize("curve,group", [("bn254", 0), ("bn254", 1), ("bls12_381", 0), ("bls12_381", 1), ("bls12_377", 0), ("bls12_377", 1)])
def test_xyzz_group_law(hip, curve, group):
    G = cv.CURVES[curve][group]
    cid = H.CURVE_IDS[curve]
    L = hip.lib()
    r = H.rng(3)
    pts = H.rand_points(G, 6, r)
    pb = hip.point_bytes(cid, group)
    acc = np.zeros(2 * pb // 8, dtype=np.uint64)     # XYZZ infinity (all zero)
    want = None
    # sequence covers: inf + P, P + Q, P + P (doubling branch), P + (-P) (-> inf), inf handling
    seq = [pts[0], pts[1], pts[1], pts[2], None, pts[3]]
    for P in seq:
        ap = cv.pack_points(G, [P]).reshape(-1)
        out = np.zeros_like(acc)
        assert L.csh_selftest_curve_op(cid, group, 0, acc.ctypes.data_as(C.c_void_p), ap.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p)) == 0
        acc = out
        want = G.add(want, P)
        assert G.eq(_xyzz_to_affine(hip, cid, group, G, acc), want)
    # P + P via madd when acc == P
    accP = np.zeros_like(acc)
    ap = cv.pack_points(G, [pts[4]]).reshape(-1)
    assert L.csh_selftest_curve_op(cid, group, 5, ap.ctypes.data_as(C.c_void_p), None, 0, accP.ctypes.data_as(C.c_void_p)) == 0
    out = np.zeros_like(acc)
    assert L.csh_selftest_curve_op(cid, group, 0, accP.ctypes.data_as(C.c_void_p), ap.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p)) == 0
    assert G.eq(_xyzz_to_affine(hip, cid, group, G, out), G.double(pts[4]))
    # P + (-P)
    an = cv.pack_points(G, [G.neg(pts[4])]).reshape(-1)
    assert L.csh_selftest_curve_op(cid, group, 0, accP.ctypes.data_as(C.c_void_p), an.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p)) == 0
    assert _xyzz_to_affine(hip, cid, group, G, out) is None
    # general add, doubling, small multiples
    out2 = np.zeros_like(acc)
    assert L.csh_selftest_curve_op(cid, group, 1, acc.ctypes.data_as(C.c_void_p), accP.ctypes.data_as(C.c_void_p), 0, out2.ctypes.data_as(C.c_void_p)) == 0
    assert G.eq(_xyzz_to_affine(hip, cid, group, G, out2), G.add(want, pts[4]))
    assert L.csh_selftest_curve_op(cid, group, 1, acc.ctypes.data_as(C.c_void_p), acc.ctypes.data_as(C.c_void_p), 0, out2.ctypes.data_as(C.c_void_p)) == 0
    assert G.eq(_xyzz_to_affine(hip, cid, group, G, out2), G.double(want))
    assert L.csh_selftest_curve_op(cid, group, 2, acc.ctypes.data_as(C.c_void_p), None, 0, out2.ctypes.data_as(C.c_void_p)) == 0
    assert G.eq(_xyzz_to_affine(hip, cid, group, G, out2), G.double(want))
    for k in [0, 1, 2, 3, 7, 1000, 32768, 65535]:
        assert L.csh_selftest_curve_op(cid, group, 3, acc.ctypes.data_as(C.c_void_p), None, k, out2.ctypes.data_as(C.c_void_p)) == 0
        assert G.eq(_xyzz_to_affine(hip, cid, group, G, out2), G.mul(want, k))


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
@pytest.mark.parametrize("c", [2, 3, 4, 7, 11, 13, 16, 17, 20])
def test_signed_digit_recoding(hip, curve, c):
    F = H.FR[curve]
    r = H.rng(c)
    vals = [0, 1, F.p - 1, F.p - 2, (1 << (c - 1)), (1 << (c - 1)) + 1, (1 << c) - 1, (F.p - 1) // 2] + H.rand_elems(F, 50, r)
    for s in vals:
        digits = (C.c_int32 * 200)()
        W = C.c_int(0)
        sc = H.pack(F, [s], mont=False)
        assert hip.lib().csh_selftest_digits(H.CURVE_IDS[curve], sc.ctypes.data_as(C.c_void_p), c, digits, C.byref(W)) == 0
        d = list(digits)[:W.value]
        assert all(abs(x) <= 1 << (c - 1) for x in d)
        assert sum(x << (c * w) for w, x in enumerate(d)) == s


def test_lazy29_mul_matches_montgomery32(hip):
    F = fl.BN254_FQ
    r = H.rng(29)
    vals = [v % F.p for v in H.edge_elems(F)] + H.rand_elems(F, 200, r)
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 1) % len(vals)]
        pa, pb = H.pack(F, [a]), H.pack(F, [b])
        out = np.zeros(4, dtype=np.uint64)
        assert hip.lib().csh_test_mul29_host(pa.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
        assert H.unpack(F, out) == [a * b % F.p]


def test_signed_lazy_field_ops(hip):
    """FpS (signed 29-bit lazy field used by the bucket kernels): (a +/- b) * c and its zero test."""
    F = fl.BN254_FQ
    r = H.rng(31)
    vals = [v % F.p for v in H.edge_elems(F)] + H.rand_elems(F, 300, r)
    L = hip.lib()
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 1) % len(vals)]
        c = vals[(i * 11 + 2) % len(vals)]
        for op, want in [(0, (a + b) * c % F.p), (1, (a - b) * c % F.p)]:
            out = np.zeros(4, dtype=np.uint64)
            z = L.csh_selftest_lazys_op(op, H.pack(F, [a]).ctypes.data_as(C.c_void_p), H.pack(F, [b]).ctypes.data_as(C.c_void_p),
                                        H.pack(F, [c]).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            assert H.unpack(F, out) == [want]
            assert z == (1 if ((a + b) if op == 0 else (a - b)) % F.p == 0 else 0)
    # exact zero / p-multiples must be detected: a - a, a + (p - a)
    for a in vals[:20]:
        out = np.zeros(4, dtype=np.uint64)
        pa = H.pack(F, [a]).ctypes.data_as(C.c_void_p)
        assert L.csh_selftest_lazys_op(1, pa, pa, pa, out.ctypes.data_as(C.c_void_p)) == 1
        pn = H.pack(F, [(-a) % F.p]).ctypes.data_as(C.c_void_p)
        assert L.csh_selftest_lazys_op(0, pa, pn, pa, out.ctypes.data_as(C.c_void_p)) == 1


@pytest.mark.parametrize("curve,group", [("bn254", 0), ("bn254", 1), ("bls12_381", 0), ("bls12_381", 1), ("grumpkin", 0), ("bls12_377", 0), ("bls12_377", 1)])
def test_lazy_bucket_accumulation_matches_group_law(hip, curve, group):
    """lazy_madd chain (incl. duplicates -> doubling, P + (-P) -> empty, infinity bases, long chains)."""
    G = cv.CURVES[curve][group]
    cid = H.CURVE_IDS[curve]
    r = H.rng(33 + group)
    pts = H.rand_points(G, 40, r)
    seqs = [
        ([pts[0]], [0]),
        ([pts[0], pts[0]], [0, 0]),                                  # doubling
        ([pts[0], pts[0]], [0, 1]),                                  # cancellation
        ([pts[0], pts[0], pts[1]], [0, 1, 0]),                       # empty then restart
        ([pts[0], pts[1], G.add(pts[0], pts[1])], [0, 0, 0]),        # acc == next point -> doubling mid-chain
        ([pts[0], pts[1], G.add(pts[0], pts[1])], [0, 0, 1]),        # acc == -next -> empty
        ([None, pts[2], None, pts[3]], [0, 1, 0, 1]),
        (pts, [r.randrange(2) for _ in pts]),
        ([pts[5]] * 33, [0] * 33),                                   # 33 * P
    ]
    for seq, neg in seqs:
        want = None
        for P, ng in zip(seq, neg):
            want = G.add(want, G.neg(P) if ng else P)
        ap = cv.pack_points(G, seq).reshape(-1)
        ngb = np.array(neg, dtype=np.uint8)
        out = np.zeros(2 * hip.point_bytes(cid, group) // 8, dtype=np.uint64)
        assert hip.lib().csh_selftest_lazy_accumulate(cid, group, ap.ctypes.data_as(C.c_void_p), ngb.ctypes.data_as(C.c_void_p),
                                                      C.c_size_t(len(seq)), out.ctypes.data_as(C.c_void_p)) == 0
        got = _xyzz_to_affine(hip, cid, group, G, out)
        assert G.eq(got, want)


@pytest.mark.parametrize("curve,group", [("bn254", 0), ("bn254", 1), ("bls12_381", 0), ("bls12_381", 1), ("grumpkin", 0), ("bls12_377", 0), ("bls12_377", 1)])
def test_lazy_point_tree_matches_group_law(hip, curve, group):
    """General XYZZ + XYZZ / doubling / small scalar in the lazy field (what the merge and reduce kernels run):
    group sums folded pairwise, incl. equal sums (-> doubling), opposite sums (-> infinity), empty groups."""
    G = cv.CURVES[curve][group]
    cid = H.CURVE_IDS[curve]
    r = H.rng(91 + group)
    pts = H.rand_points(G, 24, r)
    cases = [
        (pts, [r.randrange(2) for _ in pts], 3, 1),
        (pts, [0] * len(pts), 5, 29),
        (pts[:4] + pts[:4], [0] * 8, 4, 7),                 # two equal group sums -> lazy_dbl inside lazy_add
        (pts[:4] + pts[:4], [0] * 4 + [1] * 4, 4, 3),       # S + (-S) -> infinity
        (pts[:4] + pts[:4] + pts[4:8], [0] * 4 + [1] * 4 + [0] * 4, 4, 32767),
        ([None, None, pts[0], pts[1], None, None], [0] * 6, 2, 12345),  # empty groups around a real one
        ([pts[0]], [1], 1, 65535),
        (pts[:2], [0, 0], 1, 0),
    ]
    for seq, neg, glen, weight in cases:
        want = None
        for P, ng in zip(seq, neg):
            want = G.add(want, G.neg(P) if ng else P)
        want = G.mul(want, weight)
        ap = cv.pack_points(G, seq).reshape(-1)
        ngb = np.array(neg, dtype=np.uint8)
        out = np.zeros(2 * hip.point_bytes(cid, group) // 8, dtype=np.uint64)
        assert hip.lib().csh_selftest_lazy_tree(cid, group, ap.ctypes.data_as(C.c_void_p), ngb.ctypes.data_as(C.c_void_p),
                                                C.c_size_t(len(seq)), C.c_size_t(glen), C.c_uint32(weight),
                                                out.ctypes.data_as(C.c_void_p)) == 0
        got = _xyzz_to_affine(hip, cid, group, G, out)
        assert G.eq(got, want), (glen, weight)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_rep3_mask_generator_matches_rngs_rs(hip, curve):
    """The device mask generator's code (run on the host) vs mpc-core/src/protocols/rep3/rngs.rs:137-156 restated:
    two ChaCha12 keystreams, 32-byte chunks, from_be_bytes_mod_order(a) - from_be_bytes_mod_order(b)."""
    from oracle import chacha, mpc
    F = H.FR[curve]
    s1, s2 = bytes([1] * 32), bytes([2] * 32)        # the fixed seeds of the reference's determinism test (rngs.rs:331-360)
    for (e1, e2, n) in [(0, 0, 9), (1, 4, 8), (1000001, 77, 5)]:
        out = np.zeros(4 * n, dtype=np.uint64)
        assert hip.lib().csh_selftest_rep3_masks_host(H.CURVE_IDS[curve], s1, C.c_uint64(e1), s2, C.c_uint64(e2),
                                                     out.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == 0
        a = chacha.keystream(s1, 32 * n, start_byte=32 * e1)
        b = chacha.keystream(s2, 32 * n, start_byte=32 * e2)
        assert H.unpack(F, out) == mpc.masks_from_streams(F, a, b, n)

@pytest.mark.parametrize("curve,group", [("bn254", 0), ("bn254", 1), ("bls12_381", 0), ("bls12_381", 1), ("grumpkin", 0), ("bls12_377", 0), ("bls12_377", 1)])
def test_host_window_fold_64bit_limbs(hip, curve, group):
    """csh_msm_fold_partials (host only: Horner over window sums in 64-bit limbs, host_fp64.hpp) for every group:
    two partials with the same layout (summed window-wise first), one with another layout, one empty."""
    import struct
    G = cv.CURVES[curve][group]
    cid = H.CURVE_IDS[curve]
    r = H.rng(404 + group)
    pb = hip.point_bytes(cid, group)                 # affine bytes; an XYZZ entry is twice that
    one = cv.pack_points(G, [G.gen])                 # only used for its shape
    F = G.F
    base = F.base if hasattr(F, "base") else F

    def xyzz(P):
        if P is None:
            return bytes(2 * pb)
        xy = cv.pack_points(G, [P]).tobytes()
        k = F.ncoeff()
        mont_one = base.to_mont(1).to_bytes(base.nbytes, "little") + bytes(base.nbytes * (k - 1))
        return xy + mont_one + mont_one

    hdr_len = hip.msm_partial_bytes(cid, group) - 128 * 2 * pb

    def partial(c, pts):
        hdr = struct.pack("<4I", 0x4D534D50, c, len(pts), 0).ljust(hdr_len, b"\0")
        return hdr + b"".join(xyzz(P) for P in pts) + bytes(2 * pb * (128 - len(pts)))

    def horner(c, pts):
        acc = None
        for P in reversed(pts):
            for _ in range(c):
                acc = G.add(acc, acc)
            acc = G.add(acc, P)
        return acc

    pa = [G.mul(G.gen, r.randrange(1, 1 << 64)) for _ in range(5)]
    pb_ = [G.mul(G.gen, r.randrange(1, 1 << 64)) for _ in range(5)]
    pb_[2] = None                                    # an empty window
    pc = [G.mul(G.gen, r.randrange(1, 1 << 64)) for _ in range(3)]
    empty = partial(0, [])
    buf = partial(7, pa) + partial(7, pb_) + empty + partial(11, pc)
    want = G.add(G.add(horner(7, pa), horner(7, pb_)), horner(11, pc))
    out = np.zeros(3 * pb // 16, dtype=np.uint64)
    assert hip.lib().csh_msm_fold_partials(cid, group, buf, C.c_size_t(4), out.ctypes.data_as(C.c_void_p)) == 0
    assert G.eq(H.jac_to_affine(G, out), want)
    # P + (-P) -> infinity encoding (1, 1, 0)
    buf = partial(7, pa) + partial(7, [G.neg(P) for P in pa])
    assert hip.lib().csh_msm_fold_partials(cid, group, buf, C.c_size_t(2), out.ctypes.data_as(C.c_void_p)) == 0
    assert H.jac_to_affine(G, out) is None


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_lazy_ntt_butterfly_arithmetic(hip, curve):
    """The lazy-field butterfly chain of the NTT kernels (k_ntt_pass_lazy) run on the host with limb-bound assertions:
    a +/- k * b * w for k up to 24 stages of drift, edge values included, reduced by canonical_wide()."""
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(1234)
    vals = [0, 1, F.p - 1, F.p - 2, (F.p - 1) // 2] + H.rand_elems(F, 12, r)
    L = hip.lib()
    for i, a in enumerate(vals):
        b = vals[(3 * i + 1) % len(vals)]
        w = vals[(5 * i + 2) % len(vals)]
        for k in (0, 1, 2, 11, 24):
            for neg in (0, 1):
                out = np.zeros(4, dtype=np.uint64)
                rc = L.csh_selftest_lazy_fr_chain(cid, H.pack(F, [a]).ctypes.data_as(C.c_void_p), H.pack(F, [b]).ctypes.data_as(C.c_void_p),
                                                  H.pack(F, [w]).ctypes.data_as(C.c_void_p), k, neg, out.ctypes.data_as(C.c_void_p))
                assert rc == 0
                want = (a - k * b * w) % F.p if neg else (a + k * b * w) % F.p
                assert H.unpack(F, out) == [want], (a, b, w, k, neg)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_lazy_canonicalisation_at_the_multiples_of_p(hip, curve):
    """canonical_wide() = fold_top() + the branch-free canonical_narrow(): results that land on, just below and just above a multiple of p
    (a +/- k with a in {0, 1, p - 1, ...}: the product b w = 1 comes out of a Montgomery reduction as 1 or 1 + p, so the drifted sum sits
    within a few units of k' p for every k' the drift reaches) must come out exact and canonical."""
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    L = hip.lib()
    one = H.pack(F, [1]).ctypes.data_as(C.c_void_p)
    for a in (0, 1, 2, 23, 24, 25, F.p - 1, F.p - 2, F.p - 24, F.p - 25, (F.p - 1) // 2, (F.p + 1) // 2):
        for k in list(range(0, 26)):
            for neg in (0, 1):
                out = np.zeros(4, dtype=np.uint64)
                rc = L.csh_selftest_lazy_fr_chain(cid, H.pack(F, [a]).ctypes.data_as(C.c_void_p), one, one, k, neg, out.ctypes.data_as(C.c_void_p))
                assert rc == 0
                assert H.unpack(F, out) == [(a - k) % F.p if neg else (a + k) % F.p], (a, k, neg)
                assert int.from_bytes(out.tobytes(), "little") == F.to_mont((a - k) % F.p if neg else (a + k) % F.p)   # the canonical encoding itself


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_lazy_share_vector_products(hip, curve):
    """The lazy-field products of the share-vector kernels (vec_ops.hip; operands re-sliced as they are, one scaled by
    2^5 = R'/2^256) run on the host with limb-bound assertions: a*b and the Rep3 local multiplication, edge values included."""
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(4321)
    vals = [0, 1, 2, F.p - 1, F.p - 2, (F.p - 1) // 2, (F.p + 1) // 2] + H.rand_elems(F, 10, r)
    L = hip.lib()
    pk = lambda v: H.pack(F, [v]).ctypes.data_as(C.c_void_p)
    for i, a in enumerate(vals):
        for j, b in enumerate(vals):
            c, d, m = vals[(i + 2 * j + 1) % len(vals)], vals[(3 * i + j + 2) % len(vals)], vals[(i * j + 3) % len(vals)]
            out = np.zeros(4, dtype=np.uint64)
            assert L.csh_selftest_lazy_vec(cid, 0, pk(a), pk(b), pk(c), pk(d), pk(m), out.ctypes.data_as(C.c_void_p)) == 0
            assert H.unpack(F, out) == [a * b % F.p]
            assert L.csh_selftest_lazy_vec(cid, 1, pk(a), pk(b), pk(c), pk(d), pk(m), out.ctypes.data_as(C.c_void_p)) == 0
            assert H.unpack(F, out) == [(a * (c + d) + b * c + m) % F.p]



@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377"])
def test_lazy_fp2_products_at_the_edge_of_their_column_bound(hip, curve):
    """Fp2S mul / sqr / sqr_sub / mul_sub on RAW signed limbs (csh_selftest_fp2s_raw): limbs 0 .. NL-2 of every operand at +/-(2^B + 8) (the top limb within the value contract, |x| < 8p) -- the
    normalised-operand bound the 63-bit column sums were derived for (BLS12-377: (1 + 5) 14 2^56 products + 14 2^56 of the reduction) --,
    mixed signs, and random limbs in that range; the result equals big-integer arithmetic on the values the limbs spell (an overflowing
    column would show here, on the host run of the same template code the kernels use)."""
    G2 = cv.CURVES[curve][1]
    F2, Fq = G2.F, G2.F.base
    p = Fq.p
    B, NL = (29, 9) if curve == "bn254" else (28, 14)
    lim = (1 << B) + 8
    Rinv = pow(1 << (B * NL), -1, p)
    val = lambda l: sum(int(x) << (B * i) for i, x in enumerate(l))
    rep = lambda e: (val(e[0]) * Rinv % p, val(e[1]) * Rinv % p)
    r = H.rng(12377)
    top = 4 * (p >> (B * (NL - 1)))          # the value contract: operands within (-8p, 8p) -- the top limb carries sign and size, the rest the bound
    tl = lambda: [r.randrange(-top, top + 1)]
    patterns = [
        lambda: [lim] * (NL - 1) + tl(), lambda: [-lim] * (NL - 1) + tl(), lambda: [lim if i % 2 else -lim for i in range(NL - 1)] + tl(),
        lambda: [r.choice((lim, -lim)) for _ in range(NL - 1)] + tl(), lambda: [r.randrange(-lim, lim + 1) for _ in range(NL - 1)] + tl(),
    ]
    L = hip.lib()
    for trial in range(60):
        els = [(patterns[r.randrange(5)](), patterns[r.randrange(5)]()) for _ in range(4)]
        if trial < 5:
            els = [(patterns[trial](), patterns[trial]()) for _ in range(4)]
        flat = np.array([x for e in els for comp in e for x in comp], dtype=np.int32)
        a, b, c, d = (rep(e) for e in els)
        want = [F2.mul(a, b), F2.sqr(a), F2.sub(F2.sqr(a), b), F2.sub(F2.mul(a, b), F2.mul(c, d))]
        for op in range(4):
            out = np.zeros(2 * Fq.nlimbs, dtype=np.uint64)
            assert L.csh_selftest_fp2s_raw(H.CURVE_IDS[curve], op, flat.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
            assert tuple(H.unpack(Fq, out)) == want[op], (trial, op)
