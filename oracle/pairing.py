"""Optimal-ate pairing product check for BN254 and BLS12-381 (oracle; test infrastructure only).

This is the independent checker that pins the oracle: it must accept the snarkjs proofs the
reference commits under ``test_vectors/Groth16/*/*/circom.proof`` (what the reference's own tests
assert with ``Groth16::verify``: ``co-circom/co-groth16/src/lib.rs:72-91, 123-161``;
``verifier.rs:17-30``) and the proofs produced by the restated prover.

Generic affine Miller loop over Fp12 = Fp[w]/(w^12 + c6 w^6 + c0) (textbook, py_ecc-shaped):
BN254:     w^12 - 18 w^6 + 82, ate loop 29793968203157093288, two Frobenius line steps;
BLS12-381: w^12 -  2 w^6 +  2, ate loop 15132376222941642752.
"""
from __future__ import annotations

from . import fields as fl


class Fp12Ctx:
    def __init__(self, p, mod_c0, mod_c6):
        # w^12 = -(mod_c0) - (mod_c6) w^6
        self.p = p
        self.c0 = mod_c0 % p
        self.c6 = mod_c6 % p

    def one(self):
        return [1] + [0] * 11

    def zero(self):
        return [0] * 12

    def add(self, a, b):
        p = self.p
        return [(x + y) % p for x, y in zip(a, b)]

    def sub(self, a, b):
        p = self.p
        return [(x - y) % p for x, y in zip(a, b)]

    def neg(self, a):
        p = self.p
        return [(-x) % p for x in a]

    def scal(self, a, k):
        p = self.p
        return [x * k % p for x in a]

    def mul(self, a, b):
        p = self.p
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    t[i + j] += x * y
        for i in range(22, 11, -1):
            v = t[i] % p
            if v:
                t[i - 12] -= v * self.c0
                t[i - 6] -= v * self.c6
        return [x % p for x in t[:12]]

    def eq(self, a, b):
        return all((x - y) % self.p == 0 for x, y in zip(a, b))

    def pow(self, a, e):
        r = self.one()
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r

    def inv(self, a):
        """Solve a*x = 1 via the 12x12 multiplication matrix (Gaussian elimination mod p)."""
        p = self.p
        cols = []
        basis = self.one()
        for k in range(12):
            e = [0] * 12
            e[k] = 1
            cols.append(self.mul(a, e))
        # matrix M[row][col] = cols[col][row]; solve M x = e0
        M = [[cols[c][r] for c in range(12)] + [1 if r == 0 else 0] for r in range(12)]
        for i in range(12):
            piv = next(r for r in range(i, 12) if M[r][i] % p)
            M[i], M[piv] = M[piv], M[i]
            iv = pow(M[i][i], -1, p)
            M[i] = [x * iv % p for x in M[i]]
            for r in range(12):
                if r != i and M[r][i]:
                    f = M[r][i]
                    M[r] = [(x - f * y) % p for x, y in zip(M[r], M[i])]
        del basis
        return [M[r][12] for r in range(12)]

    def is_zero(self, a):
        return all(x % self.p == 0 for x in a)


class PairingParams:
    def __init__(self, name, p, r, c0, c6, xi_a, loop, bn_frobenius, twist_mul):
        self.name = name
        self.p = p
        self.r = r
        self.K = Fp12Ctx(p, c0, c6)
        self.xi_a = xi_a          # Fp2 a+bi  ->  (a - xi_a*b) + b w^6
        self.loop = loop
        self.bn_frobenius = bn_frobenius
        self.twist_mul = twist_mul  # True: (x w^2, y w^3); False: (x / w^2, y / w^3)


BN254 = PairingParams("bn254", fl.BN254_Q, fl.BN254_R, 82, -18, 9, 29793968203157093288, True, True)
BLS381 = PairingParams("bls12_381", fl.BLS381_Q, fl.BLS381_R, 2, -2, 1, 15132376222941642752, False, False)
# BLS12-377: Fq12 = Fq[w] / (w^12 + 5) (w^6 = u, u^2 = -5), optimal ate loop x = 0x8508C00000000001 > 0, D-type twist (x w^2, y w^3)
BLS377 = PairingParams("bls12_377", fl.BLS377_Q, fl.BLS377_R, 5, 0, 0, 0x8508C00000000001, False, True)
PARAMS = {"bn254": BN254, "bls12_381": BLS381, "bls12_377": BLS377}


def _embed_fp(K, a):
    return [a % K.p] + [0] * 11


def _embed_fp2(pp, a):
    K = pp.K
    v = [0] * 12
    v[0] = (a[0] - pp.xi_a * a[1]) % K.p
    v[6] = a[1] % K.p
    return v


def _wpow(K, k):
    v = [0] * 12
    v[k] = 1
    return v


def _twist(pp, Q):
    K = pp.K
    x, y = _embed_fp2(pp, Q[0]), _embed_fp2(pp, Q[1])
    if pp.twist_mul:
        return (K.mul(x, _wpow(K, 2)), K.mul(y, _wpow(K, 3)))
    return (K.mul(x, K.inv(_wpow(K, 2))), K.mul(y, K.inv(_wpow(K, 3))))


def _double(K, P):
    x, y = P
    lam = K.mul(K.scal(K.mul(x, x), 3), K.inv(K.scal(y, 2)))
    nx = K.sub(K.mul(lam, lam), K.scal(x, 2))
    ny = K.sub(K.mul(lam, K.sub(x, nx)), y)
    return (nx, ny)


def _add(K, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if K.eq(x1, x2):
        if K.eq(y1, y2):
            return _double(K, P)
        return None
    lam = K.mul(K.sub(y2, y1), K.inv(K.sub(x2, x1)))
    nx = K.sub(K.sub(K.mul(lam, lam), x1), x2)
    ny = K.sub(K.mul(lam, K.sub(x1, nx)), y1)
    return (nx, ny)


def _line(K, P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if not K.eq(x1, x2):
        m = K.mul(K.sub(y2, y1), K.inv(K.sub(x2, x1)))
        return K.sub(K.mul(m, K.sub(xt, x1)), K.sub(yt, y1))
    if K.eq(y1, y2):
        m = K.mul(K.scal(K.mul(x1, x1), 3), K.inv(K.scal(y1, 2)))
        return K.sub(K.mul(m, K.sub(xt, x1)), K.sub(yt, y1))
    return K.sub(xt, x1)


def miller_loop(pp: PairingParams, Q2, P1):
    """Q2: affine G2 point (Fp2 coords); P1: affine G1 point. Returns unreduced Fp12 value."""
    K = pp.K
    if Q2 is None or P1 is None:
        return K.one()
    Q = _twist(pp, Q2)
    P = (_embed_fp(K, P1[0]), _embed_fp(K, P1[1]))
    R = Q
    f = K.one()
    nbits = pp.loop.bit_length()
    for i in range(nbits - 2, -1, -1):
        f = K.mul(K.mul(f, f), _line(K, R, R, P))
        R = _double(K, R)
        if (pp.loop >> i) & 1:
            f = K.mul(f, _line(K, R, Q, P))
            R = _add(K, R, Q)
    if pp.bn_frobenius:
        Q1 = (K.pow(Q[0], pp.p), K.pow(Q[1], pp.p))
        nQ2 = (K.pow(Q1[0], pp.p), K.neg(K.pow(Q1[1], pp.p)))
        f = K.mul(f, _line(K, R, Q1, P))
        R = _add(K, R, Q1)
        f = K.mul(f, _line(K, R, nQ2, P))
    return f


def pairing_product_is_one(curve: str, pairs) -> bool:
    """prod_i e(P_i, Q_i) == 1 for pairs [(P_i in G1, Q_i in G2)]."""
    pp = PARAMS[curve]
    K = pp.K
    f = K.one()
    for P1, Q2 in pairs:
        f = K.mul(f, miller_loop(pp, Q2, P1))
    f = K.pow(f, (pp.p ** 12 - 1) // pp.r)
    return K.eq(f, K.one())
