"""GPU parity (through the C ABI): MSM vs the oracle on all four pairing groups and Grumpkin. Bit-exact on the affine result."""
import numpy as np
import pytest

from oracle import curves as cv
from tests import helpers as H

pytestmark = pytest.mark.gpu
GROUPS = [("bn254", 0), ("bn254", 1), ("bls12_381", 0), ("bls12_381", 1), ("grumpkin", 0)]


def _run(gpu, curve, group, pts, scalars, montgomery=True, offset=0, n=None):
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    bases = gpu.Bases(H.CURVE_IDS[curve], group, cv.pack_points(G, pts))
    out = bases.msm(H.pack(F, scalars, mont=montgomery), offset=offset, n=n, montgomery=montgomery)
    bases.free()
    return H.jac_to_affine(G, out)


@pytest.mark.parametrize("curve,group", GROUPS)
@pytest.mark.parametrize("n", [0, 1, 2, 3, 33, 257])
def test_msm_small_matches_oracle(gpu, curve, group, n):
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    r = H.rng(1000 + n + group)
    pts = H.rand_points(G, n, r)
    sc = H.rand_elems(F, n, r)
    assert G.eq(_run(gpu, curve, group, pts, sc), G.msm(pts, sc))
    # msm_bigint semantics: canonical (non-Montgomery) scalars give the same group element
    assert G.eq(_run(gpu, curve, group, pts, sc, montgomery=False), G.msm(pts, sc))


@pytest.mark.parametrize("curve,group", GROUPS)
def test_msm_edge_cases(gpu, curve, group):
    """Edge suites of SURVEY 8d config 2: zero/one/r-1/small scalars, duplicates, P and -P, infinity bases
    (the zkey queries contain points at infinity: multiplier2 A-query[3])."""
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    r = H.rng(42 + group)
    n = 64
    pts = H.rand_points(G, n, r, with_inf=True)
    pts[10] = pts[11]                      # duplicate point
    pts[20] = G.neg(pts[21])               # P and -P
    suites = {
        "zeros": [0] * n,
        "ones": [1] * n,
        "r-1": [F.p - 1] * n,
        "small64": [r.randrange(1 << 64) for _ in range(n)],
        "mixed": [0, 1, F.p - 1, 2, (F.p - 1) // 2, (F.p + 1) // 2] * 10 + [5, 6, 7, 8],
        "equal_cancel": [7] * n,
    }
    for name, sc in suites.items():
        got = _run(gpu, curve, group, pts, sc)
        assert G.eq(got, G.msm(pts, sc)), name
    # all-infinity bases and empty slices
    assert _run(gpu, curve, group, [None] * 8, [3] * 8) is None
    # offset / sub-slice ("unchecked": the caller passes the shorter length, honk_curve.rs:33-34)
    sc = H.rand_elems(F, n, r)
    assert G.eq(_run(gpu, curve, group, pts, sc[:20], offset=7, n=20), G.msm(pts[7:27], sc[:20]))


def test_msm_window_sizes(gpu):
    """Every window width the heuristic can choose gives the same element (csh_tune_set("msm_c") forces c)."""
    G = cv.BN254_G1
    F = H.FR["bn254"]
    r = H.rng(8)
    n = 300
    pts = H.rand_points(G, n, r, with_inf=True)
    sc = H.rand_elems(F, n, r)
    want = G.msm(pts, sc)
    for c in [2, 3, 5, 8, 11, 13, 16]:
        with gpu.tuned(msm_c=c):
            assert G.eq(_run(gpu, "bn254", 0, pts, sc), want), c
    # both scatter variants (single-level LDS cursors / two-level tile sort), uniform and skewed digits
    sk2 = [1] * 120 + [F.p - 1] * 80 + [5 << 200] * 60 + sc[:40]
    want2 = G.msm(pts, sk2)
    # msm_variant 32: level 2 of the two-level sort with one block per partition instead of one per tile-sized slice (the default)
    for mode, variant in [(0, 0), (1, 0), (1, 32)]:
        for c in [11, 14, 16]:
            with gpu.tuned(sort_two_level=mode, msm_c=c, msm_variant=variant):
                assert G.eq(_run(gpu, "bn254", 0, pts, sc), want), (mode, variant, c)
                assert G.eq(_run(gpu, "bn254", 0, pts, sk2), want2), (mode, variant, c)
    # tiny task length: forces many tasks per bucket (the skew path) on a skewed scalar set
    with gpu.tuned(msm_l=3):
        sk = [1] * 150 + [F.p - 1] * 100 + sc[:50]
        assert G.eq(_run(gpu, "bn254", 0, pts, sk), G.msm(pts, sk))


@pytest.mark.parametrize("curve,group", [("bn254", 1), ("bls12_381", 1), ("bn254", 0), ("bls12_381", 0)])
@pytest.mark.parametrize("variant", [1, 2, 4, 6, 16, 17])
def test_msm_kernel_form_variants(gpu, curve, group, variant):
    """Every form of the bucket kernels that is not the default of its group (tune "msm_variant"): G2: bit 1 = the other accumulate
    form of the group (two lanes per point, csrc/curve_pair.hpp, on BN254 G2; whole points per lane on BLS12-381 G2, whose default is the pair), bit 0 / bit 2 = four-lane / lane-serial window reduction instead of the two-lane
    one; G1: bit 0 = lane-serial window reduction instead of the four-lane one; bit 4 (16) = the reduction merges each bucket's
    partial slots itself instead of reading the merge launch's dense array (17 on G2: the same with the four-lane reduction). Same
    group element as the oracle on random points with duplicates, P / -P, points at infinity and the edge scalars (r - 1 on every
    point = one giant bucket: the queue of k_msm_mark_giant), at several window widths (the doubling and cancellation paths run
    through the DPP exchanges too)."""
    if group == 0 and variant not in (1, 16):
        pytest.skip("bits 1 and 2 only select G2 kernels")
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    r = H.rng(700 + variant)
    n = 300
    pts = H.rand_points(G, n, r, with_inf=True)
    pts[10] = pts[11]
    pts[20] = G.neg(pts[21])
    pts[30:40] = [pts[30]] * 10            # a run of equal points: doublings inside one bucket
    for name, sc in {"random": H.rand_elems(F, n, r), "ones": [1] * n, "r-1": [F.p - 1] * n,
                     "mixed": ([0, 1, F.p - 1, 2, (F.p - 1) // 2, 7] * 50)}.items():
        want = G.msm(pts, sc)
        for c in (0, 4, 11):
            with gpu.tuned(msm_variant=variant, msm_c=c):
                assert G.eq(_run(gpu, curve, group, pts, sc), want), (name, c)


def test_msm_arkworks_affine_stride(gpu):
    """Bases passed with a stride (arkworks Affine = x, y, infinity flag + padding) need no repacking."""
    G = cv.BN254_G1
    F = H.FR["bn254"]
    r = H.rng(77)
    n = 50
    pts = H.rand_points(G, n, r)
    sc = H.rand_elems(F, n, r)
    packed = cv.pack_points(G, pts)                          # (n, 8) u64
    strided = np.zeros((n, 9), dtype=np.uint64)              # 72-byte stride
    strided[:, :8] = packed
    strided[:, 8] = 0xDEADBEEF                               # junk where the flag/padding lives
    bases = gpu.Bases(0, 0, strided, stride_bytes=72)
    got = H.jac_to_affine(G, bases.msm(H.pack(F, sc)))
    assert G.eq(got, G.msm(pts, sc))


@pytest.mark.parametrize("curve,group,logn", [("bn254", 0, 20), ("bn254", 1, 16), ("bls12_381", 0, 16), ("bls12_381", 1, 14)])
def test_msm_large_tiled_bases(gpu, curve, group, logn):
    """BASELINE config 2 size (2^20 for BN254 G1): k distinct bases tiled to n entries, uniform scalars.
    MSM = sum_j (sum_{i = j mod k} s_i) * P_j, which the oracle evaluates as a k-point MSM."""
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    r = H.rng(2020 + group)
    k = 256
    n = 1 << logn
    pts = H.rand_points(G, k, r, with_inf=True)
    tiled = np.tile(cv.pack_points(G, pts), (n // k, 1))
    rs = np.random.RandomState(7)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)                             # canonical values < 2^253 < r
    bases = gpu.Bases(H.CURVE_IDS[curve], group, tiled)
    got = H.jac_to_affine(G, bases.msm(limbs, montgomery=False))
    # column sums of the scalars, exact big-int arithmetic via 32-bit halves
    lo = (limbs & np.uint64(0xFFFFFFFF)).reshape(n // k, k, 4)
    hi = (limbs >> np.uint64(32)).reshape(n // k, k, 4)
    slo, shi = lo.sum(axis=0, dtype=np.uint64), hi.sum(axis=0, dtype=np.uint64)   # < 2^32 * 2^12: no overflow
    sums = []
    for j in range(k):
        v = 0
        for l in range(4):
            v += (int(slo[j, l]) + (int(shi[j, l]) << 32)) << (64 * l)
        sums.append(v % F.p)
    assert G.eq(got, G.msm(pts, sums))


def _gen_bases(gpu, curve, group, seed, n):
    import ctypes as C
    cid = H.CURVE_IDS[curve]
    buf = gpu.DeviceBuffer(n * gpu.point_bytes(cid, group))
    gpu.bindings._check(gpu.lib().csh_util_generate_bases_dev(cid, group, C.c_uint64(seed), C.c_size_t(n), buf.ptr, None))
    gpu.bindings.sync()
    return buf


@pytest.mark.parametrize("curve,group", GROUPS)
def test_generated_bases_have_known_dlog(gpu, curve, group):
    from tests.check_closed_form import dlogs
    G = cv.CURVES[curve][group]
    n = 5
    buf = _gen_bases(gpu, curve, group, 99, n)
    pts = cv.unpack_points(G, buf.to_host())
    ks = dlogs(99, n)
    for P, k in zip(pts, ks):
        assert G.eq(P, G.mul(G.gen, int(k)))


@pytest.mark.parametrize("curve,group,logn", [("bn254", 0, 20), ("bn254", 0, 22), ("bn254", 0, 24), ("bls12_381", 0, 18), ("bls12_381", 0, 22),
                                               ("bn254", 1, 17), ("bn254", 1, 20), ("bls12_381", 1, 16), ("bls12_381", 1, 20),
                                               ("grumpkin", 0, 18)])
def test_msm_closed_form_full_size(gpu, curve, group, logn):
    """Known-dlog bases: MSM == (sum s_i k_i) G at BASELINE sizes, uniform scalars, Montgomery input."""
    import ctypes as C
    from tests.check_closed_form import closed_form_point
    G = cv.CURVES[curve][group]
    cid = H.CURVE_IDS[curve]
    n = 1 << logn
    seed = 0xC0FFEE + logn
    buf = _gen_bases(gpu, curve, group, seed, n)
    h = C.c_void_p()
    gpu.bindings._check(gpu.lib().csh_bases_upload_dev(cid, group, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
    buf.free()
    rs = np.random.RandomState(logn)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    out = np.zeros(3 * gpu.point_bytes(cid, group) // 16, dtype=np.uint64)
    gpu.bindings._check(gpu.lib().csh_msm(h, C.c_size_t(0), C.c_size_t(n), limbs.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p)))
    want = closed_form_point(curve, group, seed, n, limbs, True)
    assert G.eq(H.jac_to_affine(G, out), want)
    if group == 0 and curve == "bn254":
        # both intermediate record formats of the two-level sort (4 bytes up to 2^23 entries, 8 bytes beyond / when forced)
        # must order the same entries into the same buckets
        with gpu.tuned(msm_variant=8):
            out2 = np.zeros_like(out)
            gpu.bindings._check(gpu.lib().csh_msm(h, C.c_size_t(0), C.c_size_t(n), limbs.ctypes.data_as(C.c_void_p), 1, out2.ctypes.data_as(C.c_void_p)))
        assert G.eq(H.jac_to_affine(G, out2), want)
    gpu.lib().csh_bases_free(h)


@pytest.mark.parametrize("curve,group,logn", [("bn254", 0, 13), ("bn254", 0, 15), ("bn254", 0, 16), ("bn254", 0, 17), ("bn254", 0, 18), ("bn254", 0, 19),
                                               ("bls12_381", 0, 16), ("bn254", 1, 15), ("bls12_381", 1, 14), ("grumpkin", 0, 16)])
def test_msm_balanced_windows_at_proving_key_sizes(gpu, curve, group, logn):
    """Balanced windows (round 5; msm_impl.hpp choose_windows): W windows share the bits + 1 bits evenly -- widths c and c - 1 -- instead of
    c-bit windows with whatever is left on top (3 bits of 12 at 2^16). Known-dlog bases, uniform Montgomery scalars: the default plan, the
    uniform plan of rounds 1-4 (msm_balanced = 0) and every forced W around the model's choice give (sum s_i k_i) G, bit-identical to each
    other; odd lengths and an offset into the handle ride along."""
    import ctypes as C
    from tests.check_closed_form import closed_form_point, dlogs, weighted_sum
    G = cv.CURVES[curve][group]
    cid = H.CURVE_IDS[curve]
    F = H.FR[curve]
    n = (1 << logn) + 77
    seed = 0xBA1A + logn
    buf = _gen_bases(gpu, curve, group, seed, n)
    h = C.c_void_p()
    gpu.bindings._check(gpu.lib().csh_bases_upload_dev(cid, group, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
    buf.free()
    rs = np.random.RandomState(100 + logn)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    limbs[5] = 0
    limbs[6] = np.array([((F.p - 1) >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)   # R^-1-scaled r - 1: a top-heavy canonical value
    out = np.zeros(3 * gpu.point_bytes(cid, group) // 16, dtype=np.uint64)

    def run(count, offset=0, sc=limbs):
        gpu.bindings._check(gpu.lib().csh_msm(h, C.c_size_t(offset), C.c_size_t(count), sc.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p)))
        return out.copy()

    want = closed_form_point(curve, group, seed, n, limbs, True)
    got = run(n)
    assert G.eq(H.jac_to_affine(G, got), want)
    c, w, _l, _s = gpu.bindings.msm_last_params()
    assert w * (c - 1) < F.p.bit_length() + 1 <= w * c                     # the windows cover bits + 1 with less than one bit each to spare
    with gpu.tuned(msm_balanced=0):
        assert np.array_equal(run(n), got)
        cu, wu, _l, _s = gpu.bindings.msm_last_params()
    assert w == wu and c <= cu                                            # the tuned window count, its bits spread evenly
    for fw in sorted({max(16, w - 2), max(16, w - 1), w, w + 1, w + 3, 2 * w, 127}):
        with gpu.tuned(msm_w=fw):
            assert np.array_equal(run(n), got), fw
            assert gpu.bindings.msm_last_params()[1] == fw
    # a sub-range of the handle with its own (smaller) plan
    m, off = (1 << (logn - 2)) + 3, 41
    sub = np.ascontiguousarray(limbs[:m])
    S = weighted_sum(sub, dlogs(seed, m, off)) % F.p * F.Rinv % F.p
    assert G.eq(H.jac_to_affine(G, run(m, off, sub)), G.mul(G.gen, S))
    gpu.lib().csh_bases_free(h)


def test_concurrent_callers_share_the_device(gpu):
    """The reference calls the hot path from rayon workers and scoped threads at once (5 MSM closures, 3 NTT pipelines,
    SURVEY 8b "Threading"): eight host threads issue MSMs of different sizes on two curves plus NTT round trips
    concurrently, several rounds each (stream lanes and arenas are leased per thread); every result must be exact."""
    import threading
    from oracle import ntt as ontt
    jobs = []
    r = H.rng(99)
    for i, (curve, group, n) in enumerate([("bn254", 0, 3000), ("bn254", 0, 41), ("bn254", 1, 500), ("bls12_381", 0, 900),
                                           ("grumpkin", 0, 1200), ("bn254", 0, 7000)]):
        G = cv.CURVES[curve][group]
        F = H.FR[curve]
        pts = H.rand_points(G, n, r, with_inf=True)
        sc = H.rand_elems(F, n, r)
        jobs.append(("msm", curve, group, pts, sc, G.msm(pts, sc)))
    Fr = H.FR["bn254"]
    for logn in (9, 12):
        v = H.rand_elems(Fr, 1 << logn, r)
        jobs.append(("ntt", logn, v))
    errors = []

    def run(job):
        try:
            for _ in range(4):
                if job[0] == "msm":
                    _, curve, group, pts, sc, want = job
                    assert cv.CURVES[curve][group].eq(_run(gpu, curve, group, pts, sc), want)
                else:
                    _, logn, v = job
                    dom = gpu.Domain(gpu.BN254, logn, H.pack(Fr, [ontt.Domain.snarkjs(Fr, 1 << logn).gen]))
                    x = dom.ifft_in_to_out(H.pack(Fr, v))
                    assert H.unpack(Fr, dom.fft_out_to_in(x)) == v
                    dom.free()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("tables,share", [(0, 2), (17, 2), (17, 1)])
def test_concurrent_host_scalar_calls_share_one_upload(gpu, tables, share):
    """share = 2 (the default, round 6): the calls that arrive while the first one uploads are RUN by it as one csh_msm_multi_dev (one digit
    sort for the handles of equal length and offset: here the three G1 handles and the G2 handle); share = 1: they share the upload only.
    tables = 17: the same on handles that carry the round-6 fixed-base tables (one bucket set, the wide sort stage on five streams at once).
    Round 5: concurrent csh_msm calls handed the SAME host scalar slice (the reference's rayon_join5: A, B/G1, B/G2 and L all read
    aux_assignment, groth16.rs:227-294) share one upload -- a call that finds another one in flight with the same (device, pointer,
    length) reads that call's device copy. Known-dlog bases of four groups / seeds, 2^17 + 5 scalars (4 MiB: above the sharing threshold),
    six rounds of four concurrent callers + one caller on a different slice of the same vector: every result equals the closed form and
    the one computed alone with sharing off; the counter shows that copies were shared; nothing is shared between calls that do not overlap."""
    import ctypes as C
    import threading
    from tests.check_closed_form import closed_form_point, dlogs, weighted_sum
    from cosnarks_amd import bindings as B
    n = (1 << 17) + 5
    F = H.FR["bn254"]
    jobs = [("bn254", 0, 0x51), ("bn254", 0, 0x52), ("bn254", 1, 0x53), ("bn254", 0, 0x54)]
    handles = []
    for curve, group, seed in jobs:
        buf = _gen_bases(gpu, curve, group, seed, n)
        h = C.c_void_p()
        gpu.bindings._check(gpu.lib().csh_bases_upload_dev(H.CURVE_IDS[curve], group, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
        buf.free()
        if tables:
            gpu.bindings._check(gpu.lib().csh_bases_precompute(h, tables))
        handles.append(h)
    rs = np.random.RandomState(17)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    want = [closed_form_point(c, g, s, n, limbs, True) for c, g, s in jobs]
    m, off = n - 4097, 4097                                      # the fifth caller: another slice (other pointer), bases of job 0 from `off`
    sub = limbs[off:]
    S = weighted_sum(sub, dlogs(jobs[0][2], m, off)) % F.p * F.Rinv % F.p
    want_sub = cv.BN254_G1.mul(cv.BN254_G1.gen, S)

    def call(i, out, scalars=limbs, count=n, offset=0):
        gpu.bindings._check(gpu.lib().csh_init(0))
        gpu.bindings._check(gpu.lib().csh_msm(handles[i], C.c_size_t(offset), C.c_size_t(count), scalars.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p)))

    outs = [np.zeros(3 * gpu.point_bytes(H.CURVE_IDS[c], g) // 16, dtype=np.uint64) for c, g, _ in jobs]
    alone = []
    with gpu.tuned(msm_share_uploads=0):
        for i in range(4):
            call(i, outs[i])
            alone.append(outs[i].copy())
    shared0 = B.tune_get("stat_uploads_shared")
    for i in range(4):                                           # sequential calls never overlap: nothing to share
        call(i, outs[i])
    assert B.tune_get("stat_uploads_shared") == shared0
    errs = []
    gpu.bindings.tune_set("msm_share_uploads", share)
    for _ in range(6):
        o5 = np.zeros_like(outs[0])
        res = [np.zeros_like(o) for o in outs]

        def guarded(fn, *a, **k):
            try:
                fn(*a, **k)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        th = [threading.Thread(target=guarded, args=(call, i, res[i])) for i in range(4)]
        th.append(threading.Thread(target=guarded, args=(call, 0, o5), kwargs=dict(scalars=sub, count=m, offset=off)))
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i, (c, g, _s) in enumerate(jobs):
            G = cv.CURVES[c][g]
            assert G.eq(H.jac_to_affine(G, res[i]), want[i]) and np.array_equal(res[i], alone[i]), i
        assert cv.BN254_G1.eq(H.jac_to_affine(cv.BN254_G1, o5), want_sub)
    gpu.bindings.tune_set("msm_share_uploads", 2)
    assert B.tune_get("stat_uploads_shared") > shared0
    for h in handles:
        gpu.lib().csh_bases_free(h)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_msm_multi_shares_one_sort(gpu, curve):
    """csh_msm_multi_dev: several MSMs (G1 and G2 mixed, different offsets) over one scalar vector, one digit sort;
    every result equals the separate MSM / the oracle. Includes a skewed scalar set and n = 0."""
    import ctypes as C
    G1, G2 = cv.CURVES[curve]
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(2025)
    n = 600
    sets = [(G1, 0, H.rand_points(G1, n + 5, r, with_inf=True), 5), (G1, 0, H.rand_points(G1, n, r), 0),
            (G2, 1, H.rand_points(G2, n + 2, r), 2), (G1, 0, H.rand_points(G1, n + 9, r), 9)]
    handles = [gpu.Bases(cid, g, cv.pack_points(G, pts)) for G, g, pts, _ in sets]
    L = gpu.lib()
    for sc in (H.rand_elems(F, n, r), [1] * 200 + [F.p - 1] * 200 + H.rand_elems(F, 200, r), []):
        m = len(sc)
        dsc = gpu.DeviceBuffer.from_host(H.pack(F, sc) if m else np.zeros(4, dtype=np.uint64))
        outs = [np.zeros(3 * gpu.point_bytes(cid, g) // 16, dtype=np.uint64) for _, g, _, _ in sets]
        hs = (C.c_void_p * 4)(*[h.h.value for h in handles])
        offs = (C.c_size_t * 4)(*[o for *_, o in sets])
        po = (C.c_void_p * 4)(*[o.ctypes.data for o in outs])
        gpu.bindings._check(L.csh_msm_multi_dev(hs, offs, C.c_size_t(4), C.c_size_t(m), dsc.ptr, 1, po, None))
        for (G, g, pts, off), out in zip(sets, outs):
            assert G.eq(H.jac_to_affine(G, out), G.msm(pts[off:off + m], sc)), (g, off, m)
        dsc.free()
    for h in handles:
        h.free()


TABLE_LAYOUTS = ((0, 0), (13, 0), (16, 2), (13, 4), (11, 5))       # (c, groups); groups = 0: one row per window
TABLE_LAYOUTS_LONG = ((8, 0), (16, 0), (15, 2), (9, 3))            # -m gpu_long (tests/test_gpu_long.py)


@pytest.mark.parametrize("curve,group", GROUPS)
def test_msm_fixed_base_tables(gpu, curve, group, layouts=TABLE_LAYOUTS):
    """csh_bases_precompute / csh_bases_precompute_grouped: with tables on the handle every MSM (full, prefix, offset, skewed
    scalars, canonical scalars) returns the same group element as the oracle; several table widths and row counts (2, 3, 4, 5
    rows: windows w and w + W' k share a bucket set; the last group is ragged when rows x W' > windows)."""
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(555 + group)
    n = 1100                                                   # >= 1024: tables are built
    pts = H.rand_points(G, n, r, with_inf=True)
    sc = H.rand_elems(F, n, r)
    sk = [1] * 300 + [F.p - 1] * 300 + [0] * 100 + H.rand_elems(F, n - 700, r)
    want_full, want_sk = G.msm(pts, sc), G.msm(pts, sk)
    want_off = G.msm(pts[37:37 + 900], sc[:900])
    for c, groups in layouts:
        bases = gpu.Bases(cid, group, cv.pack_points(G, pts)).precompute(c, groups)
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc))), want_full), c
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sk))), want_sk), c
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc[:900]), offset=37, n=900)), want_off), c
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc, mont=False), montgomery=False)), want_full), c
        # a tiny MSM on the same handle takes the ordinary path
        assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc[:20]), n=20)), G.msm(pts[:20], sc[:20])), c
        bases.free()


@pytest.mark.parametrize("curve,group", GROUPS)
def test_msm_wide_window_tables(gpu, curve, group):
    """Fixed-base tables with ONE bucket set and windows of 17 .. 22 bits (csh_bases_precompute with c > 16: msm_sort_wide.hip, 2^16 ..
    2^21 buckets, 32-bit digit keys sorted in two levels): full, prefix-with-offset, skewed (0 / 1 / -1-heavy) and canonical scalars equal
    the oracle for every partition width of the second level (tune msm_wide_lb) and both record sizes (msm_variant bit 3)."""
    G = cv.CURVES[curve][group]
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(777 + group)
    n = 1300
    pts = H.rand_points(G, n, r, with_inf=True)
    sc = H.rand_elems(F, n, r)
    sk = [1] * 300 + [F.p - 1] * 300 + [0] * 100 + H.rand_elems(F, n - 700, r)
    want_full, want_sk = G.msm(pts, sc), G.msm(pts, sk)
    want_off = G.msm(pts[37:37 + 900], sc[:900])
    try:
        for c, lb, variant in ((17, 0, 0), (19, 9, 0), (20, 11, 8), (22, 10, 0), (22, 0, 8)):
            bases = gpu.Bases(cid, group, cv.pack_points(G, pts)).precompute(c, 0)
            gpu.bindings.tune_set("msm_wide_lb", lb)
            gpu.bindings.tune_set("msm_variant", variant)
            assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc))), want_full), c
            assert gpu.bindings.msm_last_params()[:2] == [c, 1], "the wide single-bucket-set plan must be the one that ran"
            assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sk))), want_sk), c
            assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc[:900]), offset=37, n=900)), want_off), c
            assert G.eq(H.jac_to_affine(G, bases.msm(H.pack(F, sc, mont=False), montgomery=False)), want_full), c
            bases.free()
    finally:
        gpu.bindings.tune_set("msm_wide_lb", 0)
        gpu.bindings.tune_set("msm_variant", 0)


def test_msm_fixed_base_tables_closed_form_and_multi(gpu):
    """2^19 known-dlog bases with tables (2^20 on table handles: tests/test_gpu_fullsize.py, the bench line): closed form; the shared-sort multi-MSM over handles with tables (mixed G1 / G2,
    different offsets) equals the oracle."""
    import ctypes as C
    from tests.check_closed_form import closed_form_point
    G = cv.BN254_G1
    n = 1 << 19
    seed = 0xFEED
    buf = _gen_bases(gpu, "bn254", 0, seed, n)
    h = C.c_void_p()
    L = gpu.lib()
    gpu.bindings._check(L.csh_bases_upload_dev(0, 0, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
    buf.free()
    rs = np.random.RandomState(7)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    want = closed_form_point("bn254", 0, seed, n, limbs, True)
    for c, groups in ((0, 0), (17, 0)):   # automatic width (c = 16, one row per window), the policy's one bucket set at this size (round 5's grouped rows: test_msm_fixed_base_tables, every layout against the oracle)
        if groups:
            gpu.bindings._check(L.csh_bases_precompute_grouped(h, c, groups))
        else:
            gpu.bindings._check(L.csh_bases_precompute(h, c))
        out = np.zeros(12, dtype=np.uint64)
        gpu.bindings._check(L.csh_msm(h, C.c_size_t(0), C.c_size_t(n), limbs.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p)))
        assert G.eq(H.jac_to_affine(G, out), want), (c, groups)
    L.csh_bases_free(h)
    # multi over tables
    G1, G2 = cv.CURVES["bn254"]
    F = H.FR["bn254"]
    r = H.rng(31337)
    m = 1500
    sets = [(G1, 0, H.rand_points(G1, m + 3, r), 3), (G1, 0, H.rand_points(G1, m + 3, r), 3), (G2, 1, H.rand_points(G2, m + 3, r), 3),
            (G1, 0, H.rand_points(G1, m, r), 0)]
    sc = H.rand_elems(F, m, r)
    for c, groups in ((0, 2), (0, 0), (18, 0)):   # (18, 0): one bucket set of 2^17 buckets, the wide sort stage shared by the four MSMs
        handles = [gpu.Bases(0, g, cv.pack_points(Gx, pts)).precompute(c, groups) for Gx, g, pts, _ in sets]
        _multi_over_tables(gpu, sets, handles, sc, m)
        for x in handles:
            x.free()


def _multi_over_tables(gpu, sets, handles, sc, m):
    import ctypes as C
    L = gpu.lib()
    F = H.FR["bn254"]
    dsc = gpu.DeviceBuffer.from_host(H.pack(F, sc))
    outs = [np.zeros(3 * gpu.point_bytes(0, g) // 16, dtype=np.uint64) for _, g, _, _ in sets]
    hs = (C.c_void_p * 4)(*[x.h.value for x in handles])
    offs = (C.c_size_t * 4)(*[o for *_, o in sets])
    po = (C.c_void_p * 4)(*[o.ctypes.data for o in outs])
    gpu.bindings._check(L.csh_msm_multi_dev(hs, offs, C.c_size_t(4), C.c_size_t(m), dsc.ptr, 1, po, None))
    for (Gx, g, pts, off), o in zip(sets, outs):
        assert Gx.eq(H.jac_to_affine(Gx, o), Gx.msm(pts[off:off + m], sc)), (g, off)
    dsc.free()


def test_plain_c_caller_of_the_boundary(gpu, tmp_path):
    """examples/msm_ntt_from_c.c: MSM and NTT through the C ABI from a C11 program built with gcc."""
    import subprocess
    from tests.test_abi_cpu import _build_c_example
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "ok: NTT round trip" in r.stdout
