"""Element-wise secret-shared field arithmetic of mpc-core (oracle; test infrastructure only).

Every function cites the reference lines it restates (paths relative to /root/reference).
Rep3 shares are ``(a, b)`` tuples of ints; Shamir shares are ints.
"""
from __future__ import annotations

from .fields import PrimeField


# ------------------------------------------------------------------ Rep3
def rep3_share(F: PrimeField, val: int, a: int, b: int):
    """mpc-core/src/protocols/rep3.rs:281-292 with the two random elements supplied."""
    c = (val - a - b) % F.p
    return [(a, c), (b, a), (c, b)]


def rep3_share_vec(F: PrimeField, vals, rng):
    """rep3.rs:375-389. ``rng()`` returns a uniform field element."""
    s = [[], [], []]
    for v in vals:
        sh = rep3_share(F, v, rng(), rng())
        for k in range(3):
            s[k].append(sh[k])
    return s


def rep3_combine(F: PrimeField, s1, s2, s3):
    """rep3.rs:583-605: x1.a + x2.a + x3.a."""
    return [(x[0] + y[0] + z[0]) % F.p for x, y, z in zip(s1, s2, s3)]


def rep3_open(F: PrimeField, share, c):
    """rep3/arithmetic.rs:249-252: a + b + c (c = the previous party's b)."""
    return (share[0] + share[1] + c) % F.p


def rep3_local_mul(F: PrimeField, x, y):
    """rep3/arithmetic/ops.rs:69-76: a*a' + a*b' + b*a'."""
    return (x[0] * y[0] + x[0] * y[1] + x[1] * y[0]) % F.p


def rep3_local_mul_vec(F: PrimeField, lhs, rhs, masks):
    """rep3/arithmetic.rs:132-146: lhs*rhs + mask, the masks (rngs.rs:137-156) supplied."""
    assert len(lhs) == len(rhs) == len(masks)
    return [(rep3_local_mul(F, x, y) + m) % F.p for x, y, m in zip(lhs, rhs, masks)]


def rep3_mul_public(F: PrimeField, x, k):
    """rep3/arithmetic/ops.rs:78-87, 109-114."""
    return (x[0] * k % F.p, x[1] * k % F.p)


def rep3_add(F: PrimeField, x, y):
    return ((x[0] + y[0]) % F.p, (x[1] + y[1]) % F.p)


def rep3_sub(F: PrimeField, x, y):
    return ((x[0] - y[0]) % F.p, (x[1] - y[1]) % F.p)


def rep3_add_public(F: PrimeField, x, k, party: int):
    """rep3/arithmetic.rs:52-58: party 0 adds to a, party 1 to b."""
    if party == 0:
        return ((x[0] + k) % F.p, x[1])
    if party == 1:
        return (x[0], (x[1] + k) % F.p)
    return x


def rep3_promote(F: PrimeField, v, party: int):
    """rep3/arithmetic/types.rs:69-82 promote_from_trivial."""
    return [(v % F.p, 0), (0, v % F.p), (0, 0)][party]


def masks_from_streams(F: PrimeField, stream1: bytes, stream2: bytes, n: int):
    """rep3/rngs.rs:137-156: from_be_bytes_mod_order(a_i) - from_be_bytes_mod_order(b_i) over
    ceil(MODULUS_BIT_SIZE/8)-byte chunks of the two keystreams."""
    fs = (F.p.bit_length() + 7) // 8
    out = []
    for i in range(n):
        a = int.from_bytes(stream1[i * fs:(i + 1) * fs], "big") % F.p
        b = int.from_bytes(stream2[i * fs:(i + 1) * fs], "big") % F.p
        out.append((a - b) % F.p)
    return out


# ------------------------------------------------------------------ Shamir
def evaluate_poly(F: PrimeField, poly, x):
    """shamir.rs:333-343 (Horner)."""
    acc = 0
    for c in reversed(poly):
        acc = (acc * x + c) % F.p
    return acc


def shamir_share(F: PrimeField, secret, num_shares, degree, rng):
    """shamir.rs:359-376: random degree-``degree`` poly, shares = evaluations at 1..n."""
    coeffs = [secret % F.p] + [rng() for _ in range(degree)]
    return [evaluate_poly(F, coeffs, i) for i in range(1, num_shares + 1)]


def lagrange_from_coeff(F: PrimeField, coeffs):
    """shamir.rs:442-460."""
    res = []
    for i in coeffs:
        num, den = 1, 1
        for j in coeffs:
            if i != j:
                num = num * j % F.p
                den = den * (j - i) % F.p
        res.append(num * pow(den, -1, F.p) % F.p)
    return res


def shamir_reconstruct(F: PrimeField, shares, lagrange):
    """shamir.rs:483-491: sum s_i * l_i."""
    return sum(s * l for s, l in zip(shares, lagrange)) % F.p


def shamir_local_mul_vec(F: PrimeField, a, b):
    """shamir/arithmetic.rs:73-79."""
    return [x * y % F.p for x, y in zip(a, b)]


def rep3_to_shamir_points(F: PrimeField, party: int):
    """bridges/rep3_to_shamir.rs:14-28. f(X) = 1 - X/z is the degree-1 poly with f(0)=1, f(z)=0
    (shamir.rs interpolate_poly_from_secret_and_zeros with one zero point)."""
    e = party + 1
    z1 = 3 if party == 0 else party
    z2 = 1 if party == 2 else party + 2
    x = (1 - e * pow(z1, -1, F.p)) % F.p
    y = (1 - e * pow(z2, -1, F.p)) % F.p
    return x, y


def rep3_to_shamir_vec(F: PrimeField, shares, party: int):
    """bridges/rep3_to_shamir.rs:43-62: a*x + b*y."""
    x, y = rep3_to_shamir_points(F, party)
    return [(s[0] * x + s[1] * y) % F.p for s in shares]
