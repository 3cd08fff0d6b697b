"""Prime fields, Fp2, and the arkworks in-memory encoding (oracle; test infrastructure only).

Encoding restated from arkworks ``Fp<MontBackend<_, N>, N>`` (ark-ff 0.6.0, the
dependency pinned at ``Cargo.toml:46``): an element is N little-endian u64 limbs
holding ``x * R mod p`` with ``R = 2^(64 N)`` (SURVEY.md section 8 notation).
"""
from __future__ import annotations

import numpy as np

# --- moduli (identical to the zkey headers of test_vectors/Groth16) -----
BN254_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BLS381_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS381_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class PrimeField:
    """Integers mod p. Elements are plain Python ints in [0, p)."""

    def __init__(self, p: int, name: str):
        self.p = p
        self.name = name
        self.nlimbs = (p.bit_length() + 63) // 64  # u64 limbs
        self.nbytes = 8 * self.nlimbs
        self.R = 1 << (64 * self.nlimbs)
        self.Rmod = self.R % p
        self.R2 = self.R * self.R % p
        self.Rinv = pow(self.R, -1, p)
        self.zero = 0
        self.one = 1
        # two-adicity / trace (ark-ff FftField::TWO_ADICITY, PrimeField::TRACE)
        t, s = p - 1, 0
        while t % 2 == 0:
            t //= 2
            s += 1
        self.two_adicity = s
        self.trace = t

    # protocol shared with Fp2 so the curve code is generic
    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def mul(self, a, b):
        return a * b % self.p

    def sqr(self, a):
        return a * a % self.p

    def neg(self, a):
        return (-a) % self.p

    def inv(self, a):
        return pow(a, -1, self.p)

    def is_zero(self, a):
        return a % self.p == 0

    def eq(self, a, b):
        return (a - b) % self.p == 0

    def from_int(self, a):
        return a % self.p

    def muli(self, a, k: int):
        return a * k % self.p

    def pow(self, a, e):
        return pow(a, e, self.p)

    def legendre(self, a):
        v = pow(a, (self.p - 1) // 2, self.p)
        return -1 if v == self.p - 1 else v

    def sqrt(self, a):
        """Any square root (Tonelli-Shanks), or None."""
        p = self.p
        a %= p
        if a == 0:
            return 0
        if self.legendre(a) != 1:
            return None
        if p % 4 == 3:
            return pow(a, (p + 1) // 4, p)
        q, s = self.trace, self.two_adicity
        z = 2
        while self.legendre(z) != -1:
            z += 1
        m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
        while t != 1:
            i, t2 = 0, t
            while t2 != 1:
                t2 = t2 * t2 % p
                i += 1
            b = pow(c, 1 << (m - i - 1), p)
            m, c = i, b * b % p
            t, r = t * c % p, r * b % p
        return r

    # --- arkworks memory encoding -------------------------------------------------------
    def to_mont(self, a: int) -> int:
        return a * self.Rmod % self.p

    def from_mont(self, a: int) -> int:
        return a * self.Rinv % self.p

    def ncoeff(self):
        return 1

    def coeffs(self, a):
        return [a]

    def from_coeffs(self, cs):
        return cs[0] % self.p


class Fp2:
    """Fp[i]/(i^2 + nr). Elements are (c0, c1) tuples. BN254 and BLS12-381 use i^2 = -1; BLS12-377 uses u^2 = -5
    (ark-bls12-377 0.6 ``Fq2Config::NONRESIDUE = -5``, un-vendored; checked below: -5 is a non-residue mod q)."""

    def __init__(self, base: PrimeField, nr: int = 1):
        self.base = base
        self.nr = nr
        assert pow((-nr) % base.p, (base.p - 1) // 2, base.p) == base.p - 1, "i^2 = -nr must have no root in the base field"
        self.p = base.p
        self.name = base.name + "^2"
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b):
        p = self.p
        return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)

    def sub(self, a, b):
        p = self.p
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - self.nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a):
        return self.mul(a, a)

    def neg(self, a):
        p = self.p
        return ((-a[0]) % p, (-a[1]) % p)

    def inv(self, a):
        p = self.p
        n = pow(a[0] * a[0] + self.nr * a[1] * a[1], -1, p)
        return (a[0] * n % p, (-a[1]) * n % p)

    def is_zero(self, a):
        return a[0] % self.p == 0 and a[1] % self.p == 0

    def eq(self, a, b):
        return (a[0] - b[0]) % self.p == 0 and (a[1] - b[1]) % self.p == 0

    def from_int(self, a):
        return (a % self.p, 0)

    def muli(self, a, k: int):
        return (a[0] * k % self.p, a[1] * k % self.p)

    def pow(self, a, e):
        r = self.one
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.sqr(a)
            e >>= 1
        return r

    def ncoeff(self):
        return 2

    def coeffs(self, a):
        return [a[0], a[1]]

    def from_coeffs(self, cs):
        return (cs[0] % self.p, cs[1] % self.p)


BN254_FQ = PrimeField(BN254_Q, "bn254.Fq")
BN254_FR = PrimeField(BN254_R, "bn254.Fr")
BLS381_FQ = PrimeField(BLS381_Q, "bls12_381.Fq")
BLS381_FR = PrimeField(BLS381_R, "bls12_381.Fr")
# BLS12-377 scalar field: only the LibSnarkReduction fixture path (Penumbra circuits, co-groth16/src/lib.rs:231-300) uses it
BLS377_R = 8444461749428370424248824938781546531375899335154063827935233455917409239041
BLS377_FR = PrimeField(BLS377_R, "bls12_377.Fr")
BLS377_Q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
BLS377_FQ = PrimeField(BLS377_Q, "bls12_377.Fq")
BLS377_FQ2 = Fp2(BLS377_FQ, 5)
BN254_FQ2 = Fp2(BN254_FQ)
BLS381_FQ2 = Fp2(BLS381_FQ)


# --- limb packing -------------------------------------------------------------------------
def int_to_limbs(x: int, nlimbs: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(8 * nlimbs, "little"), dtype="<u8").copy()


def limbs_to_int(l) -> int:
    return int.from_bytes(np.ascontiguousarray(l, dtype="<u8").tobytes(), "little")


def pack(F: PrimeField, xs, mont: bool = True) -> np.ndarray:
    """ints -> (len, nlimbs) u64 array in arkworks layout (Montgomery if ``mont``)."""
    n = F.nlimbs
    out = bytearray()
    for x in xs:
        v = F.to_mont(x % F.p) if mont else x % F.p
        out += v.to_bytes(8 * n, "little")
    return np.frombuffer(bytes(out), dtype="<u8").reshape(len(xs), n).copy()


def unpack(F: PrimeField, arr, mont: bool = True) -> list:
    a = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, F.nlimbs)
    raw = a.tobytes()
    nb = 8 * F.nlimbs
    out = []
    for i in range(a.shape[0]):
        v = int.from_bytes(raw[i * nb:(i + 1) * nb], "little")
        out.append(F.from_mont(v) if mont else v)
    return out
