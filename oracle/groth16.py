"""Groth16 prove/verify restated from co-groth16 (oracle; test infrastructure only).

Follows, line by line where cited (paths relative to /root/reference):

* ``CircomReduction::witness_map_from_matrices``  co-circom/co-groth16/src/groth16/reduction.rs:77-193
* ``evaluate_constraint`` + driver row kernels      reduction.rs:196-210, mpc/plain.rs:28-43,
                                                    mpc/rep3.rs:31-49, mpc/shamir.rs:29-49
* ``calculate_coeff``                               groth16.rs:179-203
* ``create_proof_with_assignment``                  groth16.rs:207-338
* Rep3 protocol steps                               mpc/rep3.rs:87-161, mpc-core rep3/pointshare.rs:119-155,
                                                    rep3/pointshare/ops.rs:95-102
* verification equation (snarkjs / ark-groth16)     verifier.rs:17-30

``r``/``s`` are taken as inputs (the reference draws them from ``thread_rng``, mpc/plain.rs:23-26, so its
proofs are not byte-reproducible; parity criterion there is "the proof verifies").
"""
from __future__ import annotations

from . import mpc
from . import ntt
from .pairing import pairing_product_is_one
from .zkey import ZKey


# ------------------------------------------------------------------ drivers (share semantics)
class PlainDriver:
    """mpc/plain.rs: share = half share = field element."""

    def __init__(self, F):
        self.F = F
        self.party = 0

    def zero_share(self):
        return 0

    def eval_row(self, row, public_inputs, witness):
        p = self.F.p
        acc = 0
        npub = len(public_inputs)
        for coeff, idx in row:
            v = public_inputs[idx] if idx < npub else witness[idx - npub]
            acc = (acc + coeff * v) % p
        return acc

    def eval_row_half(self, row, public_inputs, witness):
        """mpc/plain.rs:45-60 (and mpc/shamir.rs:51-68): same value as eval_row."""
        return PlainDriver.eval_row(self, row, public_inputs, witness)

    def promote(self, vals):
        return [v % self.F.p for v in vals]

    def local_mul_vec(self, a, b):
        return [x * y % self.F.p for x, y in zip(a, b)]

    def mul_table(self, v, table):
        return [x * t % self.F.p for x, t in zip(v, table)]

    def ntt(self, fn, v):
        return fn(v)

    def to_half(self, x):
        return x


class Rep3Driver:
    """mpc/rep3.rs: share = (a, b); half share = additive share."""

    def __init__(self, F, party: int, mask_fn=None):
        self.F = F
        self.party = party
        self.mask_fn = mask_fn or (lambda n: [0] * n)

    def zero_share(self):
        return (0, 0)

    def eval_row(self, row, public_inputs, witness):
        F = self.F
        acc = (0, 0)
        npub = len(public_inputs)
        for coeff, idx in row:
            if idx < npub:
                acc = mpc.rep3_add_public(F, acc, public_inputs[idx] * coeff % F.p, self.party)
            else:
                acc = mpc.rep3_add(F, acc, mpc.rep3_mul_public(F, witness[idx - npub], coeff))
        return acc

    def eval_row_half(self, row, public_inputs, witness):
        """mpc/rep3.rs:51-74: public terms on party 0 only, witness terms through the `a` component."""
        p = self.F.p
        acc = 0
        npub = len(public_inputs)
        for coeff, idx in row:
            if idx < npub:
                if self.party == 0:
                    acc = (acc + public_inputs[idx] * coeff) % p
            else:
                acc = (acc + witness[idx - npub][0] * coeff) % p
        return acc

    def promote(self, vals):
        return [mpc.rep3_promote(self.F, v, self.party) for v in vals]

    def local_mul_vec(self, a, b):
        return mpc.rep3_local_mul_vec(self.F, a, b, self.mask_fn(len(a)))

    def mul_table(self, v, table):
        return [mpc.rep3_mul_public(self.F, x, t) for x, t in zip(v, table)]

    def ntt(self, fn, v):
        # DomainCoeff on a Rep3 share: the NTT is linear, applied component-wise
        a = fn([x[0] for x in v])
        b = fn([x[1] for x in v])
        return list(zip(a, b))

    def to_half(self, x):
        return x[0]


class ShamirDriver:
    """mpc/shamir.rs: share = evaluation of the sharing polynomial; public values are added to every
    party's share (constant polynomial); half share = degree-2t share."""

    def __init__(self, F, party: int):
        self.F = F
        self.party = party

    def zero_share(self):
        return 0

    eval_row = PlainDriver.eval_row
    eval_row_half = PlainDriver.eval_row_half
    promote = PlainDriver.promote
    local_mul_vec = PlainDriver.local_mul_vec
    mul_table = PlainDriver.mul_table
    ntt = PlainDriver.ntt
    to_half = PlainDriver.to_half


# ------------------------------------------------------------------ R1CSToQAP (CircomReduction)
def witness_map_circom(zk: ZKey, drv, public_inputs, witness):
    """reduction.rs:77-193. Returns h as a list of half shares (natural order, length domain_size)."""
    F = zk.Fr
    num_constraints = zk.num_constraints
    num_inputs = zk.num_inputs
    domain_size = 1
    while domain_size < num_constraints + num_inputs:
        domain_size *= 2
    power = domain_size.bit_length() - 1
    if power > F.two_adicity:
        raise ValueError("Polynomial Degree too large")
    group_gen, coset_shift = ntt.groth16_roots_of_unity(F, power)
    dom = ntt.Domain(F, domain_size, group_gen)
    coset_table = ntt.bit_reversed_coset_table(F, coset_shift, domain_size)
    A, B = zk.matrices()

    def evaluate(matrix):
        res = [drv.eval_row(row, public_inputs, witness) for row in matrix]
        res += [drv.zero_share()] * (domain_size - len(res))
        return res

    a = evaluate(A)
    promoted = drv.promote(public_inputs)
    a[num_constraints:num_constraints + num_inputs] = promoted[:num_inputs]
    b = evaluate(B)

    def coset_eval(v):
        v = drv.ntt(dom.ifft_in_to_out, v)
        v = drv.mul_table(v, coset_table)
        return drv.ntt(dom.fft_out_to_in, v)

    a_c = coset_eval(a)
    b_c = coset_eval(b)
    ab = drv.local_mul_vec(a, b)                       # half shares from here on
    ab = dom.ifft_in_to_out(ab)
    ab = [x * t % F.p for x, t in zip(ab, coset_table)]
    c_c = dom.fft_out_to_in(ab)
    out = drv.local_mul_vec(a_c, b_c)
    return [(x - y) % F.p for x, y in zip(out, c_c)]


# ------------------------------------------------------------------ R1CSToQAP (LibSnarkReduction)
def witness_map_libsnark(F, generator: int, A, B, Cm, num_constraints: int, drv, public_inputs, witness):
    """reduction.rs:241-342. A, B, Cm: rows of (coeff, index) over public_inputs || witness. `generator` =
    P::ScalarField::GENERATOR (5 for BN254 Fr, 7 for BLS12-381 Fr). Returns the natural-order coefficients of
    H = (AB - C) / Z as half shares. No fixture of the reference tree exercises this path (its only tests need the
    absent Penumbra BLS12-377 keys): parity unpinned; tests check H(t) Z(t) = A(t) B(t) - C(t) instead."""
    p = F.p
    num_inputs = len(public_inputs)
    domain_size = 1
    while domain_size < num_constraints + num_inputs:          # Domain::new (:249): next power of two
        domain_size *= 2
    power = domain_size.bit_length() - 1
    if power > F.two_adicity:
        raise ValueError("Polynomial Degree too large")
    root = pow(ntt.arkworks_two_adic_root(F, generator), 1 << (F.two_adicity - power), p)
    dom = ntt.Domain(F, domain_size, root)
    coset_table = ntt.bit_reversed_coset_table(F, generator, domain_size)      # :255

    def coset_eval(v):
        v = drv.ntt(dom.ifft_in_to_out, v)
        v = drv.mul_table(v, coset_table)
        return drv.ntt(dom.fft_out_to_in, v)

    def evaluate(matrix):
        res = [drv.eval_row(row, public_inputs, witness) for row in matrix]
        return res + [drv.zero_share()] * (domain_size - len(res))

    a = evaluate(A)                                                             # :260-272
    promoted = drv.promote(public_inputs)
    a[num_constraints:num_constraints + num_inputs] = promoted[:num_inputs]
    a = coset_eval(a)
    b = coset_eval(evaluate(B))                                                 # :276-286
    ab = drv.local_mul_vec(a, b)                                                # :289
    c = [drv.eval_row_half(row, public_inputs, witness) for row in Cm]          # :292-306
    c += [0] * (domain_size - len(c))
    c = dom.ifft_in_to_out(c)
    c = [x * t % p for x, t in zip(c, coset_table)]
    c = dom.fft_out_to_in(c)
    v = pow((pow(generator, domain_size, p) - 1) % p, -1, p)                    # :311-314
    ab = [(x - y) * v % p for x, y in zip(ab, c)]                               # :316-322
    ab = ntt.bit_reverse(dom.ifft_in_to_out(ab))                                # :327-328
    g_inv = pow(generator, -1, p)                                               # :329-340
    cur = 1
    out = []
    for x in ab:
        out.append(x * cur % p)
        cur = cur * g_inv % p
    return out


def libsnark_identity_holds(F, generator: int, A, B, Cm, num_constraints: int, public_inputs, witness, h, tau: int) -> bool:
    """H(tau) * (tau^n - 1) == A(tau) * B(tau) - C(tau), with A, B, C interpolated from their evaluations on the arkworks
    domain (plain values). Independent of the NTT code: Lagrange basis evaluated directly."""
    p = F.p
    drv = PlainDriver(F)
    num_inputs = len(public_inputs)
    n = 1
    while n < num_constraints + num_inputs:
        n *= 2
    power = n.bit_length() - 1
    root = pow(ntt.arkworks_two_adic_root(F, generator), 1 << (F.two_adicity - power), p)
    ev = lambda M: [drv.eval_row(r, public_inputs, witness) for r in M] + [0] * (n - len(M))
    a, b, c = ev(A), ev(B), ev(Cm)
    a[num_constraints:num_constraints + num_inputs] = [v % p for v in public_inputs]
    z = (pow(tau, n, p) - 1) % p
    n_inv = pow(n, -1, p)

    def interp(vals):
        acc, w = 0, 1
        for v in vals:
            if v:
                acc = (acc + v * w % p * pow((tau - w) % p, -1, p)) % p
            w = w * root % p
        return acc * z % p * n_inv % p

    H = ntt.eval_poly_at(F, h, tau)
    return H * z % p == (interp(a) * interp(b) - interp(c)) % p


def random_r1cs(F, rng, n_public: int, n_constraints: int, fan: int = 3):
    """A satisfied R1CS for the reduction tests: variable 0 = 1, n_public - 1 public inputs, then one fresh variable per
    constraint holding (A_j . w)(B_j . w), with C_j selecting it. Returns (A, B, C, w) with rows of (coeff, index)."""
    p = F.p
    w = [1] + [rng.randrange(p) for _ in range(n_public - 1)] + [rng.randrange(p) for _ in range(2)]
    A, B, Cm = [], [], []
    for _ in range(n_constraints):
        ra = [(rng.randrange(1, p), rng.randrange(len(w))) for _ in range(rng.randrange(1, fan + 1))]
        rb = [(rng.randrange(1, p), rng.randrange(len(w))) for _ in range(rng.randrange(1, fan + 1))]
        va = sum(c * w[i] for c, i in ra) % p
        vb = sum(c * w[i] for c, i in rb) % p
        k = rng.randrange(1, p)                      # C row: k * w_new = va * vb
        w.append(va * vb % p * pow(k, -1, p) % p)
        A.append(ra)
        B.append(rb)
        Cm.append([(k, len(w) - 1)])
    return A, B, Cm, w


# ------------------------------------------------------------------ plain prove
def _calculate_coeff(G, add_public: bool, initial, query, vk_param, input_assignment, aux):
    """groth16.rs:179-203. ``add_public`` = this party adds public points (party 0 / plain / every
    Shamir party)."""
    pub_len = len(input_assignment)
    priv_acc = G.msm(query[1 + pub_len:], aux)
    res = initial
    if add_public:
        pub_acc = G.msm(query[1:1 + pub_len], input_assignment)
        res = G.add(res, query[0])
        res = G.add(res, vk_param)
        res = G.add(res, pub_acc)
    return G.add(res, priv_acc)


def prove_plain(zk: ZKey, wtns, r: int, s: int):
    """``Groth16::plain_prove::<CircomReduction>`` (groth16.rs:484-490) with r, s supplied.
    Returns (proof dict of affine points, h)."""
    F = zk.Fr
    G1, G2 = zk.G1, zk.G2
    npub = zk.num_inputs
    public_inputs = [w % F.p for w in wtns[:npub]]
    witness = [w % F.p for w in wtns[npub:]]
    drv = PlainDriver(F)
    h = witness_map_circom(zk, drv, public_inputs, witness)
    inp = public_inputs[1:]
    aux = witness
    r_g1 = _calculate_coeff(G1, True, G1.mul(zk.delta_g1, r), zk.a_query, zk.alpha_g1, inp, aux)
    s_g1 = _calculate_coeff(G1, True, G1.mul(zk.delta_g1, s), zk.b_g1_query, zk.beta_g1, inp, aux)
    s_g2 = _calculate_coeff(G2, True, G2.mul(zk.delta_g2, s), zk.b_g2_query, zk.beta_g2, inp, aux)
    l_acc = G1.msm(zk.l_query, aux)
    h_acc = G1.msm(zk.h_query, h)
    rs = r * s % F.p
    g_a = r_g1
    g_c = G1.mul(g_a, s)
    g_c = G1.add(g_c, G1.mul(s_g1, r))
    g_c = G1.add(g_c, G1.neg(G1.mul(zk.delta_g1, rs)))
    g_c = G1.add(g_c, l_acc)
    g_c = G1.add(g_c, h_acc)
    return {"a": g_a, "b": s_g2, "c": g_c}, h


# ------------------------------------------------------------------ Rep3 3-party prove (in-process)
def prove_rep3(zk: ZKey, wtns, r: int, s: int, rng, with_masks: bool = True):
    """Three parties executed in one process (the reference's tests do the same with three threads
    and LocalNetwork: tests/tests/circom/e2e_tests/rep3.rs:57-69). ``rng()`` yields uniform field
    elements for sharing and masks. Returns (proof, [h_0, h_1, h_2])."""
    F = zk.Fr
    G1, G2 = zk.G1, zk.G2
    npub = zk.num_inputs
    public_inputs = [w % F.p for w in wtns[:npub]]
    witness_shares = mpc.rep3_share_vec(F, [w % F.p for w in wtns[npub:]], rng)
    r_sh = mpc.rep3_share(F, r, rng(), rng())
    s_sh = mpc.rep3_share(F, s, rng(), rng())

    # correlated masks: party i holds PRF streams (k_i, k_{i-1}); mask_i = f(k_i) - f(k_{i-1});
    # they sum to zero (rngs.rs:103-106). Model: three random vectors per call.
    def make_mask_fns():
        cache = {}

        def for_party(party):
            calls = {"n": 0}

            def fn(n):
                key = (calls["n"], n)
                calls["n"] += 1
                if key not in cache:
                    cache[key] = [[rng() if with_masks else 0 for _ in range(n)] for _ in range(3)]
                t = cache[key]
                return [(t[party][i] - t[(party + 2) % 3][i]) % F.p for i in range(n)]
            return fn
        return [for_party(k) for k in range(3)]

    mask_fns = make_mask_fns()
    inp = public_inputs[1:]
    hs, A_p, B1_p, B2_p, L_p, H_p, rs_p = [], [], [], [], [], [], []
    for party in range(3):
        drv = Rep3Driver(F, party, mask_fns[party])
        h = witness_map_circom(zk, drv, public_inputs, witness_shares[party])
        hs.append(h)
        aux = [drv.to_half(x) for x in witness_shares[party]]
        rp, sp = r_sh[party][0], s_sh[party][0]
        A_p.append(_calculate_coeff(G1, party == 0, G1.mul(zk.delta_g1, rp), zk.a_query, zk.alpha_g1, inp, aux))
        B1_p.append(_calculate_coeff(G1, party == 0, G1.mul(zk.delta_g1, sp), zk.b_g1_query, zk.beta_g1, inp, aux))
        B2_p.append(_calculate_coeff(G2, party == 0, G2.mul(zk.delta_g2, sp), zk.b_g2_query, zk.beta_g2, inp, aux))
        L_p.append(G1.msm(zk.l_query, aux))
        H_p.append(G1.msm(zk.h_query, h))
        rs_p.append(drv.local_mul_vec([r_sh[party]], [s_sh[party]])[0])
    # network round 1: open A (broadcast), reshare B1 then local scalar mul (mask omitted: sums to 0)
    g_a = G1.add(G1.add(A_p[0], A_p[1]), A_p[2])
    proofs_c = []
    for party in range(3):
        pa, pb = B1_p[party], B1_p[(party + 2) % 3]          # (own, received from prev)
        ra, rb = r_sh[party]
        r_g1_b = G1.add(G1.add(G1.mul(pa, ra), G1.mul(pb, ra)), G1.mul(pa, rb))  # pointshare/ops.rs:95-102
        g_c = G1.mul(g_a, s_sh[party][0])
        g_c = G1.add(g_c, r_g1_b)
        g_c = G1.add(g_c, G1.neg(G1.mul(zk.delta_g1, rs_p[party])))
        g_c = G1.add(g_c, L_p[party])
        g_c = G1.add(g_c, H_p[party])
        proofs_c.append(g_c)
    g_c = G1.add(G1.add(proofs_c[0], proofs_c[1]), proofs_c[2])
    g_b = G2.add(G2.add(B2_p[0], B2_p[1]), B2_p[2])
    return {"a": g_a, "b": g_b, "c": g_c}, hs


# ------------------------------------------------------------------ LibSnark keys and proofs (arkworks' own QAP)
def libsnark_setup(F, generator: int, G1, G2, A, B, Cm, num_instance: int, num_witness: int, toxic, fixed_base):
    """ark-groth16 0.6 ``generate_parameters_with_qap::<LibsnarkReduction>`` (un-vendored; the generator behind the ``circuit.pk`` /
    ``circuit.vk`` of the reference's LibSnark tests, co-circom/co-groth16/src/lib.rs:231-300) with the toxic waste SUPPLIED:
    ``toxic = (tau, alpha, beta, gamma, delta)``. QAP evaluations per variable as ``LibsnarkReduction::instance_map_with_evaluation``
    takes them: u_j = L_j(tau) on the arkworks domain of size >= num_constraints + num_instance; a_i = sum_j A[j][i] u_j (+ u_(nc + i)
    for the instance variables: the input-consistency rows), b_i, c_i alike without the extra term; gamma_abc_i = (beta a_i + alpha b_i
    + c_i) / gamma, l_i = the same over delta; h_query[k] = tau^k Z(tau) / delta for k < domain - 1 (``h_query_scalars(m_raw - 1, ..)``).
    ``fixed_base(group, scalars)`` returns [k G] for the group's generator (oracle/c: cbridge.fixed_base_mul). The key exists only because
    the reference's own ``circuit.pk`` files are absent upstream: it is checked by the pairing equation, not against a reference file.
    Returns the key as a dict of affine points (vk fields + queries)."""
    p = F.p
    tau, alpha, beta, gamma, delta = (x % p for x in toxic)
    nc = len(A)
    n = 1
    while n < nc + num_instance:
        n *= 2
    power = n.bit_length() - 1
    root = pow(ntt.arkworks_two_adic_root(F, generator), 1 << (F.two_adicity - power), p)
    zt = (pow(tau, n, p) - 1) % p
    assert zt, "tau lies in the domain"
    zn = zt * pow(n, -1, p) % p
    u, w = [], 1
    for _ in range(n):                                           # evaluate_all_lagrange_coefficients(tau)
        u.append(zn * w % p * pow((tau - w) % p, -1, p) % p)
        w = w * root % p
    nv = num_instance + num_witness
    a, b, c = [0] * nv, [0] * nv, [0] * nv
    for i in range(num_instance):
        a[i] = u[nc + i]
    for M, acc in ((A, a), (B, b), (Cm, c)):
        for j, row in enumerate(M):
            for coeff, idx in row:
                acc[idx] = (acc[idx] + u[j] * coeff) % p
    gi, di = pow(gamma, -1, p), pow(delta, -1, p)
    comb = [(beta * x + alpha * y + z) % p for x, y, z in zip(a, b, c)]
    gamma_abc = [v * gi % p for v in comb[:num_instance]]
    l = [v * di % p for v in comb[num_instance:]]
    hq, cur = [], zt * di % p
    for _ in range(n - 1):
        hq.append(cur)
        cur = cur * tau % p
    g1 = fixed_base(0, [alpha, beta, delta] + a + b + gamma_abc + l + hq)
    g2 = fixed_base(1, [beta, gamma, delta] + b)
    o = 3
    key = {"alpha_g1": g1[0], "beta_g1": g1[1], "delta_g1": g1[2], "beta_g2": g2[0], "gamma_g2": g2[1], "delta_g2": g2[2]}
    for name, cnt in (("a_query", nv), ("b_g1_query", nv), ("gamma_abc_g1", num_instance), ("l_query", num_witness), ("h_query", n - 1)):
        key[name] = g1[o:o + cnt]
        o += cnt
    key["b_g2_query"] = g2[3:]
    return key


def prove_libsnark_plain(F, generator: int, G1, G2, key, A, B, Cm, public_inputs, witness, r: int, s: int, msm=None, h=None):
    """``Groth16::<P>::plain_prove::<LibSnarkReduction>`` (co-circom/co-groth16/src/lib.rs:280-282 -> groth16.rs:125-177, 207-338) over an
    arkworks ``ProvingKey`` (dict as libsnark_setup returns it) with r, s supplied. ``public_inputs`` includes the constant one.
    ``msm(G, points, scalars)`` defaults to the oracle's own; ``msm_unchecked`` zips to the shorter side (h has one coefficient more than
    h_query). Returns (proof, h)."""
    msm = msm or (lambda G, pts, sc: G.msm(pts, sc))
    drv = PlainDriver(F)
    if h is None:
        h = witness_map_libsnark(F, generator, A, B, Cm, len(A), drv, public_inputs, witness)
    inp = public_inputs[1:]

    def coeff(G, initial, query, vk_param):                      # calculate_coeff, groth16.rs:179-203
        npub = len(inp)
        res = G.add(G.add(initial, query[0]), vk_param)
        res = G.add(res, G.msm(query[1:1 + npub], inp))
        return G.add(res, msm(G, query[1 + npub:], witness))

    r_g1 = coeff(G1, G1.mul(key["delta_g1"], r), key["a_query"], key["alpha_g1"])
    s_g1 = coeff(G1, G1.mul(key["delta_g1"], s), key["b_g1_query"], key["beta_g1"])
    s_g2 = coeff(G2, G2.mul(key["delta_g2"], s), key["b_g2_query"], key["beta_g2"])
    l_acc = msm(G1, key["l_query"], witness)
    k = min(len(h), len(key["h_query"]))
    h_acc = msm(G1, key["h_query"][:k], h[:k])
    g_c = G1.add(G1.mul(r_g1, s), G1.mul(s_g1, r))
    g_c = G1.add(g_c, G1.neg(G1.mul(key["delta_g1"], r * s % F.p)))
    g_c = G1.add(G1.add(g_c, l_acc), h_acc)
    return {"a": r_g1, "b": s_g2, "c": g_c}, h


# ------------------------------------------------------------------ verify
def verify(curve: str, G1, vk, proof, public) -> bool:
    """e(-A, B) e(alpha, beta) e(IC0 + sum pub_i IC_{i+1}, gamma) e(C, delta) == 1."""
    acc = vk["ic"][0]
    assert len(public) + 1 == len(vk["ic"])
    for x, P in zip(public, vk["ic"][1:]):
        acc = G1.add(acc, G1.mul(P, x % G1.order))
    pairs = [
        (G1.neg(proof["a"]), proof["b"]),
        (vk["alpha_g1"], vk["beta_g2"]),
        (acc, vk["gamma_g2"]),
        (proof["c"], vk["delta_g2"]),
    ]
    return pairing_product_is_one(curve, pairs)
