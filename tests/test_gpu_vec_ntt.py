"""GPU parity (through the C ABI): share-vector arithmetic and NTT vs the oracle. Bit-exact."""
import numpy as np
import pytest

from oracle import fields as fl
from oracle import mpc, ntt
from tests import helpers as H

pytestmark = pytest.mark.gpu
CURVES = ["bn254", "bls12_381"]


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n", [0, 1, 7, 1000, 4097])
def test_vec_ops(gpu, curve, n):
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(100 + n)
    edge = [v % F.p for v in H.edge_elems(F)]
    a = (edge + H.rand_elems(F, n, r))[:n]
    b = (edge[::-1] + H.rand_elems(F, n, r))[:n]
    pa, pb = H.pack(F, a), H.pack(F, b)
    assert H.unpack(F, gpu.vec_mul(cid, pa, pb)) == [x * y % F.p for x, y in zip(a, b)]
    assert H.unpack(F, gpu.vec_add(cid, pa, pb)) == [(x + y) % F.p for x, y in zip(a, b)]
    assert H.unpack(F, gpu.vec_sub(cid, pa, pb)) == [(x - y) % F.p for x, y in zip(a, b)]
    assert H.unpack(F, gpu.vec_mul_table(cid, pa, pb)) == [x * y % F.p for x, y in zip(a, b)]


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n", [0, 1, 10, 4095, 4096, 16384])   # sizes of tests/tests/mpc/bridges.rs:11-...
def test_rep3_ops(gpu, curve, n):
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(200 + n)
    lhs = [(r.randrange(F.p), r.randrange(F.p)) for _ in range(n)]
    rhs = [(r.randrange(F.p), r.randrange(F.p)) for _ in range(n)]
    masks = H.rand_elems(F, n, r)
    pl, pr, pm = H.pack_shares(F, lhs), H.pack_shares(F, rhs), H.pack(F, masks)
    assert H.unpack(F, gpu.rep3_local_mul_vec(cid, pl, pr, pm)) == mpc.rep3_local_mul_vec(F, lhs, rhs, masks)
    if n:   # an unmasked Rep3 product is refused (it leaks cross terms once opened); the product build has no override (round 6)
        with pytest.raises(gpu.CoSnarksHipError, match="needs its masks"):
            gpu.rep3_local_mul_vec(cid, pl, pr, None)
        with pytest.raises(gpu.CoSnarksHipError, match="CSH_EXPERIMENTS"):
            gpu.bindings.tune_set("allow_unmasked_rep3", 1)
    assert H.unpack(F, gpu.rep3_local_mul_vec(cid, pl, pr, H.pack(F, [0] * n))) == mpc.rep3_local_mul_vec(F, lhs, rhs, [0] * n)
    tbl = H.rand_elems(F, n, r)
    got = H.unpack_shares(F, gpu.vec_mul_table(cid, pl, H.pack(F, tbl), ncomp=2))
    assert got == [mpc.rep3_mul_public(F, s, t) for s, t in zip(lhs, tbl)]
    assert H.unpack_shares(F, gpu.vec_sub(cid, pl, pr, ncomp=2)) == [mpc.rep3_sub(F, x, y) for x, y in zip(lhs, rhs)]
    for party in range(3):
        x, y = mpc.rep3_to_shamir_points(F, party)
        got = H.unpack(F, gpu.rep3_to_shamir_vec(cid, pl, H.pack(F, [x]), H.pack(F, [y])))
        assert got == mpc.rep3_to_shamir_vec(F, lhs, party)


@pytest.mark.parametrize("curve", CURVES)
def test_rep3_mul_and_shamir_reconstruct_roundtrip(gpu, curve):
    """share -> local mul on 3 'parties' -> combine == plain product (tests/tests/mpc/rep3.rs:286-367);
    Shamir share -> lincomb with Lagrange coefficients == secret (mpc-core shamir.rs:611-742)."""
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(77)
    n = 513
    xs, ys = H.rand_elems(F, n, r), H.rand_elems(F, n, r)
    rnd = lambda: r.randrange(F.p)
    sx, sy = mpc.rep3_share_vec(F, xs, rnd), mpc.rep3_share_vec(F, ys, rnd)
    t = [H.rand_elems(F, n, r) for _ in range(3)]
    outs = []
    for p in range(3):
        mask = [(t[p][i] - t[(p + 2) % 3][i]) % F.p for i in range(n)]
        outs.append(gpu.rep3_local_mul_vec(cid, H.pack_shares(F, sx[p]), H.pack_shares(F, sy[p]), H.pack(F, mask)))
    comb = gpu.lincomb(cid, outs, H.pack(F, [1, 1, 1]))
    assert H.unpack(F, comb) == [a * b % F.p for a, b in zip(xs, ys)]
    for (nparties, deg) in [(3, 1), (10, 6)]:
        shares = [mpc.shamir_share(F, v, nparties, deg, rnd) for v in xs]      # per value: list of party shares
        use = list(range(1, deg + 2))
        lag = mpc.lagrange_from_coeff(F, use)
        per_party = [H.pack(F, [s[p - 1] for s in shares]) for p in use]
        assert H.unpack(F, gpu.lincomb(cid, per_party, H.pack(F, lag))) == xs


def _domain(gpu, curve, logn):
    F = H.FR[curve]
    gen = ntt.roots_of_unity(F)[1][logn]
    return gpu.Domain(H.CURVE_IDS[curve], logn, H.pack(F, [gen])), ntt.Domain(F, 1 << logn, gen)


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("logn", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13])
def test_ntt_matches_oracle(gpu, curve, logn):
    F = H.FR[curve]
    r = H.rng(300 + logn)
    n = 1 << logn
    dg, do = _domain(gpu, curve, logn)
    v = H.rand_elems(F, n, r)
    pv = H.pack(F, v)
    assert H.unpack(F, dg.ifft_in_to_out(pv)) == do.ifft_in_to_out(v)
    assert H.unpack(F, dg.fft_out_to_in(pv)) == do.fft_out_to_in(v)
    assert H.unpack(F, dg.fft(pv)) == do.fft(v)
    assert H.unpack(F, dg.ifft(pv)) == do.ifft(v)
    shift = ntt.groth16_roots_of_unity(F, logn)[1] if logn < F.two_adicity else 5
    assert H.unpack(F, dg.coset_table(H.pack(F, [shift]))) == ntt.bit_reversed_coset_table(F, shift, n)
    assert H.unpack(F, gpu.bindings.bit_reverse(H.CURVE_IDS[curve], pv, logn)) == ntt.bit_reverse(v)
    # Rep3 shares: DomainCoeff on {a, b} = component-wise transform
    sh = [(r.randrange(F.p), r.randrange(F.p)) for _ in range(n)]
    got = H.unpack_shares(F, dg.ifft_in_to_out(H.pack_shares(F, sh), ncomp=2))
    wa, wb = do.ifft_in_to_out([s[0] for s in sh]), do.ifft_in_to_out([s[1] for s in sh])
    assert got == list(zip(wa, wb))
    got = H.unpack_shares(F, dg.fft_out_to_in(H.pack_shares(F, sh), ncomp=2))
    wa, wb = do.fft_out_to_in([s[0] for s in sh]), do.fft_out_to_in([s[1] for s in sh])
    assert got == list(zip(wa, wb))


def test_ntt_default_root_is_arkworks(gpu):
    """Domain::new (reduction.rs:249): arkworks 2-adic root = GENERATOR^TRACE squared down; on BN254 it equals
    the snarkjs root (both derive from 5), on BLS12-381 (generator 7) it does not."""
    for curve, g in [("bn254", 5), ("bls12_381", 7)]:
        F = H.FR[curve]
        logn = 6
        root = pow(ntt.arkworks_two_adic_root(F, g), 1 << (F.two_adicity - logn), F.p)
        dg = gpu.Domain(H.CURVE_IDS[curve], logn, None)
        do = ntt.Domain(F, 1 << logn, root)
        v = H.rand_elems(F, 1 << logn, H.rng(9))
        assert H.unpack(F, dg.fft_out_to_in(H.pack(F, v))) == do.fft_out_to_in(v)
    assert pow(ntt.arkworks_two_adic_root(fl.BN254_FR, 5), 1 << (28 - 6), fl.BN254_FR.p) == ntt.roots_of_unity(fl.BN254_FR)[1][6]


def test_ntt_degree_too_large(gpu):
    with pytest.raises(gpu.CoSnarksHipError, match="Polynomial Degree too large"):
        gpu.Domain(H.CURVE_IDS["bn254"], 29, None)


@pytest.mark.parametrize("logn", [16, 22])
def test_ntt_roundtrip_and_spot_checks_full_size(gpu, logn):
    """BASELINE config 3: BN254 NTT 2^22 -- bit-exact round trip + random outputs re-derived by Horner."""
    F = fl.BN254_FR
    n = 1 << logn
    rs = np.random.RandomState(1234)
    # uniform-ish canonical values < 2^253 < r, stored as-is (any canonical residue is a valid Montgomery form)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    dg, do = _domain(gpu, "bn254", logn)
    coeffs = dg.ifft_in_to_out(limbs)
    back = dg.fft_out_to_in(coeffs)
    assert np.array_equal(back.reshape(n, 4), limbs)
    # spot-check: evaluations X[k] = sum_i c_i w^{ik}, with c in bit-reversed storage (Horner in the C restatement: 16 indices at
    # 2^22 cost what two cost in a Python loop, which was 20 s of the metered suite)
    from oracle import cbridge
    nat = cbridge.bit_reverse(coeffs, logn)
    r = H.rng(5)
    for k in [0, 1, n - 1] + [r.randrange(n) for _ in range(13)]:
        got = cbridge.eval_poly(0, nat, H.pack(F, [pow(do.gen, k, F.p)]))
        assert np.array_equal(np.asarray(got).reshape(-1), limbs[k]), k


@pytest.mark.parametrize("curve", CURVES)
def test_rep3_masks_on_device(gpu, curve):
    """csh_rep3_masks == Rep3Rand::masking_field_elements_vec (rngs.rs:137-156) and the three parties' masks cancel."""
    import ctypes as C
    from oracle import chacha
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    keys = [bytes([k + 1] * 32) for k in range(3)]
    n = 1000
    outs = []
    for p in range(3):                      # party p: rng1 = own key, rng2 = previous party's key
        out = np.zeros(4 * n, dtype=np.uint64)
        gpu.bindings._check(gpu.lib().csh_rep3_masks(cid, keys[p], C.c_uint64(5), keys[(p + 2) % 3], C.c_uint64(5),
                                                      out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        a = chacha.keystream(keys[p], 32 * n, start_byte=160)
        b = chacha.keystream(keys[(p + 2) % 3], 32 * n, start_byte=160)
        vals = H.unpack(F, out)
        assert vals == mpc.masks_from_streams(F, a, b, n)
        outs.append(vals)
    assert all((x + y + z) % F.p == 0 for x, y, z in zip(*outs))


@pytest.mark.parametrize("curve", CURVES)
def test_sparse_constraint_evaluation_on_device(gpu, curve):
    """csh_evaluate_constraints_dev vs the driver row kernels restated in the oracle (mpc/plain.rs:28-43,
    mpc/rep3.rs:31-49): plain and the three Rep3 parties, ragged rows, empty rows, zero-padded tail."""
    from oracle import groth16 as og
    F = H.FR[curve]
    cid = H.CURVE_IDS[curve]
    r = H.rng(41)
    n_pub, n_wit, n_rows, n_out = 3, 40, 50, 64
    rows = []
    for i in range(n_rows):
        k = 0 if i % 7 == 3 else r.randrange(1, 6)
        rows.append([(r.randrange(F.p), r.randrange(n_pub + n_wit)) for _ in range(k)])
    pub = [1] + H.rand_elems(F, n_pub - 1, r)
    wit = H.rand_elems(F, n_wit, r)
    M = gpu.bindings.Matrix(cid, [[(H.pack(F, [c]), idx) for c, idx in row] for row in rows])
    drv = og.PlainDriver(F)
    want = [drv.eval_row(row, pub, wit) for row in rows] + [0] * (n_out - n_rows)
    assert H.unpack(F, M.evaluate(0, 0, H.pack(F, pub), H.pack(F, wit), n_out)) == want
    shares = mpc.rep3_share_vec(F, wit, lambda: r.randrange(F.p))
    for party in range(3):
        d3 = og.Rep3Driver(F, party)
        want3 = [d3.eval_row(row, pub, shares[party]) for row in rows] + [(0, 0)] * (n_out - n_rows)
        got = H.unpack_shares(F, M.evaluate(1, party, H.pack(F, pub), H.pack_shares(F, shares[party]), n_out))
        assert got == want3


def test_extract_component_and_current_device(gpu):
    """csh_extract_component_dev (the to_half_share map over a Rep3 share vector, groth16.rs:159-163 / mpc/rep3.rs:120-122)
    and csh_current_device."""
    import ctypes as C
    L = gpu.lib()
    dev = C.c_int(-1)
    gpu.bindings._check(L.csh_current_device(C.byref(dev)))
    assert dev.value == 0
    rs = np.random.RandomState(3)
    n = 1000
    shares = rs.randint(0, 1 << 62, size=(n, 2, 4), dtype=np.uint64)
    src = gpu.DeviceBuffer.from_host(shares)
    dst = gpu.DeviceBuffer(n * 32)
    for comp in (0, 1):
        gpu.bindings._check(L.csh_extract_component_dev(src.ptr, 2, comp, C.c_size_t(n), dst.ptr, None))
        gpu.bindings.sync()
        assert (dst.to_host(count=4 * n).reshape(n, 4) == shares[:, comp, :]).all()
    assert L.csh_extract_component_dev(src.ptr, 2, 2, C.c_size_t(n), dst.ptr, None) != 0   # component out of range
    src.free()
    dst.free()



@pytest.mark.parametrize("curve", CURVES)
def test_ntt_32bit_pass_rebuilds_its_tables_on_demand(gpu, curve):
    """The 32-bit CIOS pass (tune ntt_lazy = 0, kept for A/B runs) reads the natural-order twiddle tables, which a domain releases once
    its staged tables exist: asking for that pass rebuilds them, and both passes give the oracle's transform on the same domain."""
    F = H.FR[curve]
    r = H.rng(9)
    for logn in (3, 11, 13):
        n = 1 << logn
        dg, do = _domain(gpu, curve, logn)
        v = H.rand_elems(F, n, r)
        pv = H.pack(F, v)
        want_i, want_f = do.ifft_in_to_out(v), do.fft_out_to_in(v)
        assert H.unpack(F, dg.ifft_in_to_out(pv)) == want_i
        with gpu.tuned(ntt_lazy=0):
            assert H.unpack(F, dg.ifft_in_to_out(pv)) == want_i
            assert H.unpack(F, dg.fft_out_to_in(pv)) == want_f
        assert H.unpack(F, dg.fft_out_to_in(pv)) == want_f


def test_inherited_experiment_environment_cannot_change_a_transform(gpu):
    """with CSH_NTT_VARIANT carrying the "skip the butterflies" / "run one pass" experiment bits in the environment the
    product library still returns the oracle's transform (the bits are dropped at load; radix-4 and radix-2 pass forms, 2^12 and 2^20)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import random\n"
        "import cosnarks_amd as hip\n"
        "from oracle import fields as fl, ntt\n"
        "from oracle import cbridge\n"
        "from tests import helpers as H\n"
        "F = fl.BN254_FR\n"
        "r = random.Random(5)\n"
        "for logn in (12, 20):\n"
        "    gen = ntt.roots_of_unity(F)[1][logn]\n"
        "    dg = hip.Domain(hip.BN254, logn, H.pack(F, [gen]))\n"
        "    import numpy as np\n"
        "    v = np.random.RandomState(logn).randint(0, 1 << 62, size=(1 << logn, 4), dtype=np.uint64)\n"
        "    v[:, 3] >>= np.uint64(3)\n"
        "    want = cbridge.ntt(0, v, logn, H.pack(F, [gen]), dif=True)\n"
        "    assert np.array_equal(np.asarray(dg.ifft_in_to_out(v)).reshape(-1), np.asarray(want).reshape(-1)), logn\n"
        "print('ok')\n")
    env = dict(os.environ, CSH_NTT_VARIANT="0x11000", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
