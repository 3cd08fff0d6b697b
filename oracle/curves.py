"""Short-Weierstrass groups y^2 = x^3 + b (a = 0) over Fp / Fp2 and MSM (oracle; test infrastructure only).

Restates what the reference gets from ``ark-ec 0.6.0`` (``Affine{x,y,infinity}``, Jacobian
``Projective``) and ``taceo_ark_algebra::msm::{msm_unchecked, msm_bigint}`` (call sites
``co-circom/co-groth16/src/groth16.rs:193-194``, ``mpc/plain.rs:66-74``, ``mpc/rep3.rs:124-132``,
``mpc/shamir.rs:111-119``, ``mpc-core/src/protocols/rep3/pointshare.rs:201-222``).
The MSM result is a group element; comparisons are made on the affine (canonical) form, as the
reference normalises before returning (``groth16.rs:333-337``).

Affine points are ``(x, y)`` tuples or ``None`` (infinity). Jacobian points are ``(X, Y, Z)``.
"""
from __future__ import annotations

import numpy as np

from . import fields as fl


class Curve:
    def __init__(self, name, F, b, gen, order, scalar_field, cofactor=1):
        self.name = name
        self.F = F
        self.b = b
        self.gen = gen
        self.order = order
        self.Fr = scalar_field
        self.cofactor = cofactor

    # ----- affine -----
    def is_on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.eq(F.sqr(y), F.add(F.mul(F.sqr(x), x), self.b))

    def neg(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]))

    def add(self, P, Q):
        F = self.F
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if F.eq(x1, x2):
            if F.eq(y1, y2):
                return self.double(P)
            return None
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    def double(self, P):
        F = self.F
        if P is None:
            return None
        x1, y1 = P
        if F.is_zero(y1):
            return None
        lam = F.mul(F.muli(F.sqr(x1), 3), F.inv(F.muli(y1, 2)))
        x3 = F.sub(F.sqr(lam), F.muli(x1, 2))
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    # ----- jacobian (for speed) -----
    def to_jac(self, P):
        F = self.F
        if P is None:
            return (F.one, F.one, F.zero)
        return (P[0], P[1], F.one)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdouble(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.muli(F.sub(F.sub(F.sqr(F.add(X, B)), A), C), 2)
        E = F.muli(A, 3)
        Fq = F.sqr(E)
        X3 = F.sub(Fq, F.muli(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.muli(C, 8))
        Z3 = F.muli(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def jadd(self, J1, J2):
        F = self.F
        X1, Y1, Z1 = J1
        X2, Y2, Z2 = J2
        if F.is_zero(Z1):
            return J2
        if F.is_zero(Z2):
            return J1
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if F.eq(U1, U2):
            if F.eq(S1, S2):
                return self.jdouble(J1)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        I = F.sqr(F.muli(H, 2))
        Jv = F.mul(H, I)
        r = F.muli(F.sub(S2, S1), 2)
        V = F.mul(U1, I)
        X3 = F.sub(F.sub(F.sqr(r), Jv), F.muli(V, 2))
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.muli(F.mul(S1, Jv), 2))
        Z3 = F.mul(F.sub(F.sub(F.sqr(F.add(Z1, Z2)), Z1Z1), Z2Z2), H)
        return (X3, Y3, Z3)

    def jadd_affine(self, J, P):
        if P is None:
            return J
        return self.jadd(J, (P[0], P[1], self.F.one))

    def mul(self, P, k: int):
        """k*P for affine P; k taken mod group order only if negative."""
        if P is None or k == 0:
            return None
        if k < 0:
            return self.mul(self.neg(P), -k)
        acc = (self.F.one, self.F.one, self.F.zero)
        base = self.to_jac(P)
        for bit in bin(k)[2:]:
            acc = self.jdouble(acc)
            if bit == "1":
                acc = self.jadd(acc, base)
        return self.to_affine(acc)

    def eq(self, P, Q):
        if P is None or Q is None:
            return P is None and Q is None
        return self.F.eq(P[0], Q[0]) and self.F.eq(P[1], Q[1])

    # ----- MSM -----
    def msm_naive(self, points, scalars):
        """sum_i scalars[i]*points[i] over the shorter of the two slices ("unchecked":
        ``co-noir/co-noir-common/src/honk_curve.rs:33-34``)."""
        n = min(len(points), len(scalars))
        acc = (self.F.one, self.F.one, self.F.zero)
        for i in range(n):
            if points[i] is None or scalars[i] % self.order == 0:
                continue
            acc = self.jadd(acc, self.to_jac(self.mul(points[i], scalars[i] % self.order)))
        return self.to_affine(acc)

    def msm(self, points, scalars, c: int | None = None):
        """Windowed Pippenger bucket method (the published arkworks ``VariableBaseMSM`` shape:
        window c ~ ln(n)+2, per-window buckets, running-sum bucket reduction, Horner over windows)."""
        n = min(len(points), len(scalars))
        if n == 0:
            return None
        if c is None:
            c = 3 if n < 32 else max(3, int(np.log(n)) + 2)
        nbits = self.Fr.p.bit_length()
        F = self.F
        inf = (F.one, F.one, F.zero)
        sc = [s % self.order for s in scalars[:n]]
        window_sums = []
        for w0 in range(0, nbits, c):
            buckets = [inf] * ((1 << c) - 1)
            for i in range(n):
                d = (sc[i] >> w0) & ((1 << c) - 1)
                if d and points[i] is not None:
                    buckets[d - 1] = self.jadd_affine(buckets[d - 1], points[i])
            running, acc = inf, inf
            for bkt in reversed(buckets):
                running = self.jadd(running, bkt)
                acc = self.jadd(acc, running)
            window_sums.append(acc)
        total = inf
        for ws in reversed(window_sums):
            for _ in range(c):
                total = self.jdouble(total)
            total = self.jadd(total, ws)
        return self.to_affine(total)


# --- the four groups ------------------------------------------------------------------------
BN254_G1 = Curve("bn254.G1", fl.BN254_FQ, 3, (1, 2), fl.BN254_R, fl.BN254_FR)
_bn_b2 = fl.BN254_FQ2.mul((3, 0), fl.BN254_FQ2.inv((9, 1)))
BN254_G2 = Curve(
    "bn254.G2", fl.BN254_FQ2, _bn_b2,
    ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
      11559732032986387107991004021392285783925812861821192530917403151452391805634),
     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
      4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    fl.BN254_R, fl.BN254_FR,
    cofactor=21888242871839275222246405745257275088844257914179612981679871602714643921549)
BLS381_G1 = Curve(
    "bls12_381.G1", fl.BLS381_FQ, 4,
    (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
     0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
    fl.BLS381_R, fl.BLS381_FR, cofactor=0x396C8C005555E1568C00AAAB0000AAAB)
BLS381_G2 = Curve(
    "bls12_381.G2", fl.BLS381_FQ2, (4, 4),
    ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
      0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
     (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
      0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)),
    fl.BLS381_R, fl.BLS381_FR, cofactor=None)

# Grumpkin, the 2-cycle partner of BN254 (HonkCurve for Projective<GrumpkinConfig>, co-noir-common/src/honk_curve.rs:163):
# y^2 = x^3 - 17 over BN254 Fr ("grumpkin::b, which is -17", honk_curve.rs:114), prime order = BN254 Fq modulus, scalars
# in BN254 Fq. Generator (1, sqrt(-16)) as in ark-grumpkin 0.6 / barretenberg.
GRUMPKIN_G1 = Curve("grumpkin.G1", fl.BN254_FR, fl.BN254_R - 17,
                    (1, 17631683881184975370165255887551781615748388533673675138860), fl.BN254_Q, fl.BN254_FQ)

# BLS12-377 (ark-bls12-377 0.6, un-vendored; the curve of the reference's LibSnarkReduction fixtures, co-circom/co-groth16/src/lib.rs:
# 231-300, whose committed circuit.vk files hold G1 / G2 points of it -- tests/test_oracle.py checks them against these equations):
# q, r from the BLS12 polynomials at x = 0x8508C00000000001; E: y^2 = x^3 + 1; the D-type twist E': y^2 = x^3 + 1/u over Fq[u]/(u^2 + 5).
_b7x = 0x8508C00000000001
assert fl.BLS377_R == _b7x**4 - _b7x**2 + 1 and fl.BLS377_Q == (_b7x - 1)**2 * fl.BLS377_R // 3 + _b7x
BLS377_G1 = Curve(
    "bls12_377.G1", fl.BLS377_FQ, 1,
    (0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
     0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6),
    fl.BLS377_R, fl.BLS377_FR, cofactor=(_b7x - 1)**2 // 3)
BLS377_G2 = Curve(
    "bls12_377.G2", fl.BLS377_FQ2, fl.BLS377_FQ2.inv((0, 1)),
    ((0x018480BE71C785FEC89630A2A3841D01C565F071203E50317EA501F557DB6B9B71889F52BB53540274E3E48F7C005196,
      0x00EA6040E700403170DC5A51B1B140D5532777EE6651CECBE7223ECE0799C9DE5CF89984BFF76FE6B26BFEFA6EA16AFE),
     (0x00690D665D446F7BD960736BCBB2EFB4DE03ED7274B49A58E458C282F832D204F2CF88886D8C7C2EF094094409FD4DDF,
      0x00F8169FD28355189E549DA3151A70AA61EF11AC3D591BF12463B01ACEE304C24279B83F5E52270BD9A1CDD185EB8F93)),
    fl.BLS377_R, fl.BLS377_FR, cofactor=None)

CURVES = {"bn254": (BN254_G1, BN254_G2), "bls12_381": (BLS381_G1, BLS381_G2), "grumpkin": (GRUMPKIN_G1,),
          "bls12_377": (BLS377_G1, BLS377_G2)}


# --- wire layout at the C ABI -----------------------------------------------------------------
# affine point = x || y, each coordinate = ncoeff Montgomery field elements (c0 || c1 for Fp2),
# little-endian u64 limbs; infinity = all-zero bytes (the zkey convention).
def pack_points(curve: Curve, pts) -> np.ndarray:
    F = curve.F
    base = F.base if isinstance(F, fl.Fp2) else F
    k = F.ncoeff()
    out = bytearray()
    nb = base.nbytes
    for P in pts:
        if P is None:
            out += bytes(2 * k * nb)
            continue
        for coord in P:
            for cpt in F.coeffs(coord):
                out += base.to_mont(cpt).to_bytes(nb, "little")
    return np.frombuffer(bytes(out), dtype="<u8").reshape(len(pts), 2 * k * base.nlimbs).copy()


def unpack_points(curve: Curve, arr, ncoords: int = 2):
    """Inverse of pack_points; ncoords=3 decodes Jacobian (X,Y,Z) triples (not normalised)."""
    F = curve.F
    base = F.base if isinstance(F, fl.Fp2) else F
    k = F.ncoeff()
    a = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, ncoords * k * base.nlimbs)
    raw = a.tobytes()
    nb = base.nbytes
    pts = []
    stride = ncoords * k * nb
    for i in range(a.shape[0]):
        coords = []
        for cidx in range(ncoords):
            cs = []
            for j in range(k):
                off = i * stride + (cidx * k + j) * nb
                cs.append(base.from_mont(int.from_bytes(raw[off:off + nb], "little")))
            coords.append(F.from_coeffs(cs))
        if ncoords == 2:
            if F.is_zero(coords[0]) and F.is_zero(coords[1]):
                pts.append(None)
            else:
                pts.append(tuple(coords))
        else:
            pts.append(tuple(coords))
    return pts
