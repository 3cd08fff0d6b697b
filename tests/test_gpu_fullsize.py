"""GPU parity at the BASELINE sizes by DIRECT comparison with the CPU restatement (oracle/c), not only through properties:
NTT 2^20 / 2^22 (config 3) full-vector equality + 16 Horner indices, MSM 2^20 (config 2) on full-range random points on
three groups, the share-vector kernels at 2^24 + 5 elements (the grid-stride branch), and the reference's BN254 Fr products
(tests/tests/mpc/rep3.rs:286-345) through the device kernels. Everything through the C ABI; bit-exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import cbridge, ntt
from oracle import curves as cv
from oracle import fields as fl
from oracle import mpc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _uniform_limbs(rs, n):
    """n canonical values < 2^253 (< r on every curve here), used as Montgomery encodings of uniform field elements."""
    v = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    v[:, 3] >>= np.uint64(3)
    return v


G2_FULL_RANGE_DEFAULT = [("bn254", 20), ("bls12_381", 19)]
G2_FULL_RANGE_LONG = [("bls12_381", 20), ("bls12_381", 22)]          # tests/test_gpu_long.py (-m gpu_long)


@pytest.mark.parametrize("curve,logn", G2_FULL_RANGE_DEFAULT)
def test_msm_g2_full_range_points_equals_cpu_restatement(gpu, curve, logn):
    """G2 at BASELINE sizes on full-range points (oracle/c's progression family: blocks S_b + j D_b with 253-bit discrete logs --
    k G per point would take minutes on G2), uniform scalars: the affine result is bit-identical to oracle/c's independent
    Pippenger. 2^22 on BLS12-381 G2 is config 5's kernel at a size where the two-level sort and 16 windows are in play."""
    cid = H.CURVE_IDS[curve]
    n = 1 << logn
    pts = cbridge.generate_bases_progression(cid, 1, 0xD1CE + logn, n)
    sc = _uniform_limbs(np.random.RandomState(31 + logn), n)
    sc[:16] = 0
    bases = gpu.Bases(cid, 1, pts)
    got = bases.msm(sc, montgomery=True)
    w = got.size // 3
    got_aff = np.zeros(2 * w, dtype=np.uint64) if not got[2 * w:].any() else got[:2 * w]
    want = cbridge.msm_fast(cid, 1, pts, sc, montgomery=True)
    assert np.array_equal(got_aff, want), (curve, logn)
    bases.free()


@pytest.mark.parametrize("variant", [0, 32])
def test_msm_2p20_witness_like_scalars_equals_cpu_restatement(gpu, variant):
    """A 0/1-heavy witness at BASELINE config 2's size (canonical scalars, msm_bigint): a quarter zero, a quarter one (every "1" lands
    in bucket 1 of window 0), a quarter one repeated 253-bit value (one bucket per window holds a quarter of the window's entries), a
    quarter uniform. Bit-identical to oracle/c. variant 0: level 2 of the sort with one block per tile-sized slice of a partition (the
    default: a skewed partition is spread over the chip); 32: one block per partition."""
    cid = H.CURVE_IDS["bn254"]
    n = 1 << 20
    pts = cbridge.hash_points_bn254_g1(0xFEED, n)
    rs = np.random.RandomState(99)
    sc = _uniform_limbs(rs, n)
    kind = rs.randint(0, 4, size=n)
    sc[kind == 0] = 0
    sc[kind == 1] = np.array([1, 0, 0, 0], dtype=np.uint64)
    sc[kind == 2] = sc[0]
    bases = gpu.Bases(cid, 0, pts)
    with gpu.tuned(msm_variant=variant):
        got = bases.msm(sc, montgomery=False)
    bases.free()
    w = got.size // 3
    got_aff = np.zeros(2 * w, dtype=np.uint64) if not got[2 * w:].any() else got[:2 * w]
    assert np.array_equal(got_aff, cbridge.msm_fast(cid, 0, pts, sc, montgomery=False))


@pytest.mark.parametrize("curve,group", [("bn254", 0), ("bn254", 1), ("bls12_381", 0)])
@pytest.mark.parametrize("case", ["one_value", "many_values"])
def test_msm_sliced_giant_buckets_equal_cpu_restatement(gpu, curve, group, case):
    """Buckets with hundreds to thousands of partial sums (csrc/msm_impl.hpp: k_msm_giant_slices + k_msm_merge_giant). one_value: every
    scalar equal, one entry per accumulate lane (msm_l = 1): ONE bucket per window holds all 3000 partials -- sliced on every window.
    many_values: 400 distinct scalars repeated 300 times each at a narrow window: far more than GIANT_BIG_CAP = 128 buckets of >= 256
    partials, so the first 128 to register are sliced and the rest take the one-block path in the same launch. Bit-identical to oracle/c."""
    cid = H.CURVE_IDS[curve]
    rs = np.random.RandomState(5 + group)
    if case == "one_value":
        n, knobs = 3000, {"msm_l": 1, "msm_c": 12}
        sc = np.repeat(_uniform_limbs(rs, 1), n, axis=0)
    else:
        n, knobs = 120000, {"msm_l": 1, "msm_c": 11}
        sc = np.repeat(_uniform_limbs(rs, 400), 300, axis=0)
    pts = cbridge.generate_bases_progression(cid, group, 0xC0DE + group, n)
    bases = gpu.Bases(cid, group, pts)
    with gpu.tuned(**knobs):
        got = bases.msm(sc, montgomery=False)
    bases.free()
    w = got.size // 3
    got_aff = np.zeros(2 * w, dtype=np.uint64) if not got[2 * w:].any() else got[:2 * w]
    assert np.array_equal(got_aff, cbridge.msm_fast(cid, group, pts, sc, montgomery=False)), (curve, group, case)


NTT_FULL_DEFAULT = [(20, 1), (20, 2), (21, 1), (21, 2), (22, 1)]
NTT_FULL_LONG = [(22, 2), (23, 1)]                          # tests/test_gpu_long.py (-m gpu_long)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("logn,ncomp", NTT_FULL_DEFAULT)
def test_ntt_full_size_equals_cpu_restatement(gpu, curve, logn, ncomp):
    """BASELINE config 3 (BN254, 2^22) and the Rep3 two-component form: both directions bit-identical to oracle/c's radix-2
    NTT over the whole vector, the round trip, and 16 output indices re-derived by Horner (no NTT code involved). 2^20 / 2^21 run as two
    sweeps (128- and 64-byte runs in the strided pass, 32-byte ones for share pairs at 2^20), 2^22 / 2^23 as three: every pass plan."""
    if curve == "bls12_381" and (logn, ncomp) not in ((22, 1), (23, 1), (22, 2)):
        pytest.skip("one full-size BLS12-381 case is enough")
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    n = 1 << logn
    gen = ntt.roots_of_unity(F)[1][logn]                      # snarkjs root (reduction.rs:38-43 via groth16 roots)
    pg = H.pack(F, [gen])
    dom = gpu.Domain(cid, logn, pg)
    x = _uniform_limbs(np.random.RandomState(1000 + logn + ncomp), n * ncomp)
    coeffs = dom.ifft_in_to_out(x, ncomp=ncomp)
    want = cbridge.ntt(cid, x, logn, pg, ncomp=ncomp, dif=True)
    assert np.array_equal(coeffs.reshape(-1), want.reshape(-1))
    evals = dom.fft_out_to_in(coeffs, ncomp=ncomp)
    assert np.array_equal(evals.reshape(-1), x.reshape(-1))                                     # bit-exact round trip
    assert np.array_equal(cbridge.ntt(cid, want, logn, pg, ncomp=ncomp, dif=False).reshape(-1), x.reshape(-1))
    # Horner: x[k] (component comp) = sum_i c[bitrev(i)] w^(ik); natural-order coefficients through the CPU bit reversal
    nat = cbridge.bit_reverse(coeffs, logn, ncomp=ncomp)
    r = H.rng(logn)
    for _ in range(16):
        k, comp = r.randrange(n), r.randrange(ncomp)
        wk = H.pack(F, [pow(gen, k, F.p)])
        got = cbridge.eval_poly(cid, nat, wk, stride=ncomp, offset=comp)
        assert np.array_equal(got, x.reshape(n, ncomp, 4)[k, comp]), (k, comp)
    dom.free()


NTT_BEYOND_DEFAULT = [(24, 1)]
NTT_BEYOND_LONG = [(24, 2), (26, 1)]              # tests/test_gpu_long.py (-m gpu_long): 28 s of CPU restatement + 2 GiB buffers


@pytest.mark.parametrize("logn,ncomp", NTT_BEYOND_DEFAULT)
def test_ntt_beyond_2p23_equals_cpu_restatement(gpu, logn, ncomp):
    """VERDICT r4 missing #3: the reference accepts any domain up to TWO_ADICITY = 28 (groth16/reduction.rs:84-94); 2^24 (the size DESIGN
    quotes a time for) and 2^26 (three sweeps of 10 + 8 + 8 stages), BN254, both directions bit-identical to oracle/c's radix-2 NTT
    over the whole vector, the bit-exact round trip, and output indices re-derived by Horner. Buffers are dropped as soon as they have
    been compared (2 GiB each at 2^26)."""
    F = H.FR["bn254"]
    cid = H.CURVE_IDS["bn254"]
    n = 1 << logn
    gen = ntt.roots_of_unity(F)[1][logn]
    pg = H.pack(F, [gen])
    dom = gpu.Domain(cid, logn, pg)
    x = _uniform_limbs(np.random.RandomState(2000 + logn + ncomp), n * ncomp)
    coeffs = dom.ifft_in_to_out(x, ncomp=ncomp)
    want = cbridge.ntt(cid, x, logn, pg, ncomp=ncomp, dif=True)
    assert np.array_equal(coeffs.reshape(-1), want.reshape(-1))
    del want
    evals = dom.fft_out_to_in(coeffs, ncomp=ncomp)
    assert np.array_equal(evals.reshape(-1), x.reshape(-1))                                     # bit-exact round trip
    del evals
    fwd = dom.fft_out_to_in(x, ncomp=ncomp)                                                     # the other direction on its own input
    want = cbridge.ntt(cid, x, logn, pg, ncomp=ncomp, dif=False)
    assert np.array_equal(fwd.reshape(-1), want.reshape(-1))
    del fwd, want
    nat = cbridge.bit_reverse(coeffs, logn, ncomp=ncomp)
    del coeffs
    r = H.rng(logn)
    for _ in range(4 if logn >= 25 else 8):
        k, comp = r.randrange(n), r.randrange(ncomp)
        wk = H.pack(F, [pow(gen, k, F.p)])
        got = cbridge.eval_poly(cid, nat, wk, stride=ncomp, offset=comp)
        assert np.array_equal(got, x.reshape(n, ncomp, 4)[k, comp]), (k, comp)
    dom.free()


RANDOM_POINTS_DEFAULT = [("bn254", 0, "hashed"), ("bn254", 1, "wide")]
RANDOM_POINTS_LONG = [("bls12_381", 0, "wide"), ("bls12_381", 1, "wide20"), ("bls12_381", 0, "wide20"), ("bn254", 0, "wide20"), ("bn254", 1, "wide20")]   # tests/test_gpu_long.py (-m gpu_long)


@pytest.mark.parametrize("curve,group,family", RANDOM_POINTS_DEFAULT)
def test_msm_2p20_random_points_equals_cpu_restatement(gpu, curve, group, family):
    """BASELINE config 2 size on full-range points with no exploitable structure: BN254 G1 points hashed to the curve
    (SURVEY 8d family i), and k G with 253-bit k on every group (incl. points at infinity); uniform scalars in Montgomery form
    and canonical (msm_bigint). The affine result is bit-identical to oracle/c's independent Pippenger (Booth / XYZZ)."""
    cid = H.CURVE_IDS[curve]
    # BASELINE config 2 (BN254 G1, 2^20) is the HASHED family here; the k G family runs at 2^19 / 2^18 (G1) and 2^17 (G2) in the metered
    # suite and at 2^20 under -m gpu_long ("wide20")
    logn = 20 if family == "hashed" else ((19 if curve == "bn254" else 18) if group == 0 else 17)
    if family == "wide20":
        logn, family = (20 if group == 0 else 18), "wide"
    n = 1 << logn
    pts = cbridge.hash_points_bn254_g1(0xA11CE, n) if family == "hashed" else cbridge.generate_bases_wide(cid, group, 0xB0B + group, n)
    sc = _uniform_limbs(np.random.RandomState(7 + group), n)
    sc[:64] = 0                                                # zero scalars
    sc[64:128] = sc[128:192]                                   # repeated scalars
    bases = gpu.Bases(cid, group, pts)
    for mont in (True, False):
        got = bases.msm(sc, montgomery=mont)
        w = got.size // 3
        got_aff = np.zeros(2 * w, dtype=np.uint64) if not got[2 * w:].any() else got[:2 * w]
        want = cbridge.msm_fast(cid, group, pts, sc, montgomery=mont)
        assert np.array_equal(got_aff, want), (curve, group, family, mont)
    # round 6: the same MSM with the library's fixed-base tables on the handle (ONE bucket set, 17-bit windows at this size; 20-bit: the
    # 2^24 policy) -- bit-identical to the CPU restatement on points with no structure, canonical scalars (the last `want`)
    for c in (17, 20):
        bases.precompute(c, 0)
        got = bases.msm(sc, montgomery=False)
        assert gpu.bindings.msm_last_params()[:2] == [c, 1]
        got_aff = np.zeros(2 * w, dtype=np.uint64) if not got[2 * w:].any() else got[:2 * w]
        assert np.array_equal(got_aff, want), (curve, group, family, "tables", c)
    bases.free()


SHARE_VECTOR_CURVES_LONG = ["bls12_381"]          # tests/test_gpu_long.py (-m gpu_long)


@pytest.mark.parametrize("curve", ["bn254"])
def test_share_vector_kernels_beyond_one_launch_width(gpu, curve):
    """2^24 + 5 elements: more than 65536 workgroups x 256 lanes, so every kernel takes its grid-stride branch; each result is
    compared element for element with oracle/c."""
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    n = (1 << 24) + 5
    rs = np.random.RandomState(99)
    a2, b2 = _uniform_limbs(rs, 2 * n), _uniform_limbs(rs, 2 * n)
    m = _uniform_limbs(rs, n)
    eq = lambda x, y: np.array_equal(np.asarray(x).reshape(-1), np.asarray(y).reshape(-1))
    assert eq(gpu.rep3_local_mul_vec(cid, a2, b2, m), cbridge.rep3_local_mul_vec(cid, a2, b2, m))
    a1, b1 = a2[:n], b2[:n]
    assert eq(gpu.vec_mul(cid, a1, b1), cbridge.vec_mul(cid, a1, b1))
    assert eq(gpu.vec_sub(cid, a1, b1), cbridge.vec_sub(cid, a1, b1))
    assert eq(gpu.vec_add(cid, a1, b1), cbridge.vec_add(cid, a1, b1))
    if curve == "bn254":
        assert eq(gpu.vec_sub(cid, a2, b2, ncomp=2), cbridge.vec_sub(cid, a2, b2))
        assert eq(gpu.vec_mul_table(cid, a1, m), cbridge.vec_mul_table(cid, a1, m))
        assert eq(gpu.vec_mul_table(cid, a2, m, ncomp=2), cbridge.vec_mul_table(cid, a2, m, ncomp=2))
        x, y = mpc.rep3_to_shamir_points(F, 1)
        px, py = H.pack(F, [x]), H.pack(F, [y])
        assert eq(gpu.rep3_to_shamir_vec(cid, a2, px, py), cbridge.rep3_to_shamir_vec(cid, a2, px, py))
        co = H.pack(F, [3, F.p - 2, 12345])
        assert eq(gpu.lincomb(cid, [a1, b1, m], co), cbridge.lincomb(cid, [a1, b1, m], co))


KAT_X = [13839525561076761625780930844889299788193703994911163378019280196128582690055,
         19302971480864839163158232064620707211435225928426123775531639309944891593977,
         8048717310762513532550620831072439583505607813129662608591015555880153427210,
         2585271390974436123003027749932103593962191064365118925254473311197989280023]
KAT_Y = [2688648969035332064113669477511029957484512453056743431884706385750388613065,
         13632770404954969699480437686769008635735921498648460325387842712839596176806,
         19199593902803943133889170931116903997086625101975591190159463567024116566625,
         8255472466884305547009533395117607586789669747151273739964395707537515634749]
KAT_Z = [14012338922664984944451142760937475581748095944353358534203030914664561190462,
         4297594441150501195973997511775989720904927516253689527653694984160382713321,
         7875903949174289914141782934879682497141865775307179984684659764891697566272,
         6646526994769136778802685410292764833027657364709823469005920616147071273574]


def test_reference_bn254_fr_products_through_the_device_kernels(gpu):
    """The four products hard-coded by the reference (tests/tests/mpc/rep3.rs:286-345, the same in shamir.rs): through
    csh_vec_mul (plain / Shamir local_mul_vec), and through three Rep3 parties -- share_field_elements semantics, each party's
    csh_rep3_local_mul_vec with correlated masks, opened with csh_lincomb -- exactly the flow of that test (mul_vec then open)."""
    F = fl.BN254_FR
    cid = H.CURVE_IDS["bn254"]
    px, py = H.pack(F, KAT_X), H.pack(F, KAT_Y)
    assert H.unpack(F, gpu.vec_mul(cid, px, py)) == KAT_Z
    r = H.rng(2024)
    rnd = lambda: r.randrange(F.p)
    xs = [mpc.rep3_share(F, v, rnd(), rnd()) for v in KAT_X]   # per value: three (a, b) shares
    ys = [mpc.rep3_share(F, v, rnd(), rnd()) for v in KAT_Y]
    keys = [rnd() for _ in range(12)]                          # masks m_p = k_p - k_{p-1} per element: they cancel
    prods = []
    for p in range(3):
        lhs = H.pack_shares(F, [xs[i][p] for i in range(4)])
        rhs = H.pack_shares(F, [ys[i][p] for i in range(4)])
        mask = H.pack(F, [(keys[4 * p + i] - keys[4 * ((p + 2) % 3) + i]) % F.p for i in range(4)])
        prods.append(gpu.rep3_local_mul_vec(cid, lhs, rhs, mask))
    opened = gpu.lincomb(cid, prods, H.pack(F, [1, 1, 1]))
    assert H.unpack(F, opened) == KAT_Z


FUZZ_DEFAULT = [("bn254", 0, 12), ("bn254", 1, 4), ("bls12_381", 0, 5), ("bls12_381", 1, 3)]
FUZZ_LONG = [("bn254", 0, 40), ("bn254", 1, 8), ("bls12_381", 0, 10), ("bls12_381", 1, 5)]   # tests/test_gpu_long.py (-m gpu_long)


@pytest.mark.parametrize("curve,group,rounds", FUZZ_DEFAULT)
def test_msm_fuzz_sizes_and_plans_vs_cpu_restatement(gpu, curve, group, rounds):
    """Random sizes around the plan boundaries (single- / two-level sort, chunk counts, lane lengths) with random forced window
    widths and lane lengths, zero / repeated / extreme scalars and points at infinity mixed in: every result bit-identical to
    oracle/c's independent Pippenger."""
    cid = H.CURVE_IDS[curve]
    F = H.FR[curve]
    r = H.rng(4242 + group + 10 * cid)
    nmax = 40000 if group == 0 else 9000
    pts_all = cbridge.generate_bases_wide(cid, group, 0xF00D + group, nmax)
    rs = np.random.RandomState(11 + group)
    for it in range(rounds):
        n = r.choice([1, 2, 63, 64, 65, 255, 256, 257, 1023, 1025, 4095, 4097, r.randrange(1, nmax), r.randrange(1, nmax)])
        off = r.randrange(0, nmax - n + 1)
        sc = _uniform_limbs(rs, n)
        k = r.randrange(0, 4)
        if k == 0 and n > 8:
            sc[: n // 3] = sc[0]                          # one giant bucket per window
        elif k == 1 and n > 8:
            sc[::2] = 0
            sc[1::4] = np.array([1, 0, 0, 0], dtype=np.uint64)
        elif k == 2:
            pm1 = np.array([((F.p - 1) >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
            sc[: max(1, n // 5)] = pm1                    # canonical r - 1
        knobs = {"msm_c": r.choice([0, 0, 3, 7, 10, 12, 13, 14, 15, 16]), "msm_l": r.choice([0, 0, 1, 5, 16, 64]),
                 "sort_two_level": r.choice([-1, -1, 0, 1]), "msm_variant": r.choice([0, 0, 32, 64, 96]),
                 # balanced windows (round 5): W windows sharing the bits evenly (widths c and c - 1), forced W incl. the extremes, or off
                 "msm_balanced": r.choice([1, 1, 1, 0]), "msm_w": r.choice([0, 0, 0, 16, 17, 19, 22, 25, 31, 40, 64, 85, 127])}
        bases = gpu.Bases(cid, group, pts_all[off:off + n])
        # round 6: a third of the rounds run on fixed-base tables -- one bucket set with 17 .. 22-bit windows (the wide sort stage, every
        # second-level partition width and both record sizes) or the grouped / full-merge forms of rounds 2-5 (tables need >= 1024 points)
        tbl = r.choice([None, None, (17, 0), (18, 0), (19, 0), (20, 0), (21, 0), (22, 0), (16, 0), (13, 3)]) if n >= 1024 else None
        if tbl:
            bases.precompute(*tbl)
            knobs.update({"msm_wide_lb": r.choice([0, 8, 9, 10, 11]), "msm_wide_chunks": r.choice([0, 1, 3, 64]), "msm_variant": r.choice([0, 8])})
        with gpu.tuned(**knobs):
            got = bases.msm(sc, montgomery=False)
            if tbl and tbl[0] > 16:
                assert gpu.bindings.msm_last_params()[:2] == [tbl[0], 1], (tbl, gpu.bindings.msm_last_params())
        bases.free()
        w = got.size // 3
        got_aff = np.zeros(2 * w, dtype=np.uint64) if not got[2 * w:].any() else got[:2 * w]
        want = cbridge.msm_fast(cid, group, pts_all[off:off + n], sc, montgomery=False)
        assert np.array_equal(got_aff, want), (curve, group, it, n, off, k, knobs)


NTT_VARIANTS_DEFAULT = [0, 2, 1, 0x102, 0x401, 0x100801]
NTT_VARIANTS_LONG = [0x101, 0x201, 0x302, 0x402]   # tests/test_gpu_long.py (-m gpu_long): the other forced tile sizes


@pytest.mark.parametrize("variant", NTT_VARIANTS_DEFAULT)
@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_ntt_every_size_up_to_2p19_vs_cpu_restatement(gpu, curve, variant):
    """Every domain size 2^1 .. 2^19 (all pass plans: one, two, three and more sweeps; even and odd stage counts per pass), both
    directions, ncomp 1 and 2, against oracle/c's radix-2 NTT over the whole vector. variant 0: the default (tile size by transform
    size, radix-2 passes below 2^20 points); 2: the radix-2 pass everywhere, 1: the radix-4 pass everywhere; bits 8-10 = v force
    2^(12-v)-element tiles (tune ntt_variant), each with either pass form; 0x100801: the radix-4 pass with the unit-twiddle rounds and the
    LDS bank swizzle switched off (the plain form both are measured against)."""
    if curve == "bls12_381" and variant in (0x101, 0x302):
        pytest.skip("the forced tile sizes run on both fields with one pass form each")
    with gpu.tuned(ntt_variant=variant):
        _ntt_every_size(gpu, curve)


def _ntt_every_size(gpu, curve):
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    rs = np.random.RandomState(77)
    roots = ntt.roots_of_unity(F)[1]
    for logn in range(1, 20):
        n = 1 << logn
        gen = roots[logn]
        pg = H.pack(F, [gen])
        dom = gpu.Domain(cid, logn, pg)
        for ncomp in ((1, 2) if logn % 3 == 0 or logn >= 14 else (1,)):
            x = _uniform_limbs(rs, n * ncomp)
            assert np.array_equal(dom.ifft_in_to_out(x, ncomp=ncomp).reshape(-1), cbridge.ntt(cid, x, logn, pg, ncomp=ncomp, dif=True, threads=8).reshape(-1)), (logn, ncomp, "ifft")
            assert np.array_equal(dom.fft_out_to_in(x, ncomp=ncomp).reshape(-1), cbridge.ntt(cid, x, logn, pg, ncomp=ncomp, dif=False, threads=8).reshape(-1)), (logn, ncomp, "fft")
        dom.free()
