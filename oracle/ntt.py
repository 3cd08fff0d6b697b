"""Radix-2 NTT conventions of the reference (oracle; test infrastructure only).

Restates ``taceo_ark_algebra::fft::{Domain, bit_reverse}`` (external crate, pinned 0.1.0) from the
reference's call sites and doc comments:

* ``Domain::with_group_gen(size, gen)``          -- ``co-circom/co-groth16/src/groth16/reduction.rs:93``
* ``Domain::new(n)`` (arkworks 2-adic root)      -- ``reduction.rs:249``
* ``ifft_in_to_out``: natural-order evaluations -> coefficients in BIT-REVERSED order, scaled by 1/n
* ``fft_out_to_in`` : bit-reversed coefficients -> natural-order evaluations
  (``reduction.rs:38-43, 70-72, 234-236, 324-326``)
* ``roots_of_unity`` / ``groth16_roots_of_unity``  -- ``co-circom/co-groth16/src/groth16.rs:60-100``
* ``bit_reversed_coset_table``                     -- ``reduction.rs:45-60``
* ``EvaluationDomain::{fft, ifft}`` natural->natural with zero padding -- ark-poly 0.6.0, call sites
  ``co-circom/co-plonk/src/mpc/plain.rs:149-161``.

Outputs are unique vectors of canonical field elements once (root of unity, ordering) are fixed, so
the internal butterfly schedule is free; here: textbook iterative DIT on bit-reversed input.
"""
from __future__ import annotations

from .fields import PrimeField


def bitrev(i: int, logn: int) -> int:
    r = 0
    for _ in range(logn):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def bit_reverse(v: list) -> list:
    """``fft::bit_reverse`` (call sites reduction.rs:58, 328): out[bitrev(i)] = in[i]."""
    n = len(v)
    logn = n.bit_length() - 1
    assert 1 << logn == n
    out = list(v)
    for i in range(n):
        j = bitrev(i, logn)
        if i < j:
            out[i], out[j] = out[j], out[i]
    return out


def roots_of_unity(F: PrimeField):
    """groth16.rs:60-73: q = smallest QNR, z = q^TRACE, repeated squaring, reversed.
    roots[k] generates the subgroup of order 2^k."""
    q = 1
    while F.legendre(q) != -1:
        q += 1
    roots = [0] * (F.two_adicity + 1)
    roots[0] = pow(q, F.trace, F.p)
    for i in range(1, len(roots)):
        roots[i] = roots[i - 1] * roots[i - 1] % F.p
    roots.reverse()
    return q, roots


def groth16_roots_of_unity(F: PrimeField, power: int):
    """groth16.rs:91-100 -> (group_gen, coset_shift)."""
    q, roots = roots_of_unity(F)
    group_gen = roots[power]
    coset_shift = q * q % F.p if F.two_adicity == power else roots[power + 1]
    return group_gen, coset_shift


def arkworks_two_adic_root(F: PrimeField, generator: int):
    """ark-ff FftField::TWO_ADIC_ROOT_OF_UNITY = GENERATOR^TRACE (generator: 5 for BN254 Fr,
    7 for BLS12-381 Fr)."""
    return pow(generator, F.trace, F.p)


def _dit_from_bitreversed(F: PrimeField, v: list, w: int) -> list:
    """In: v[bitrev(i)] = c_i. Out: X[k] = sum_i c_i w^{ik}, natural order."""
    p = F.p
    n = len(v)
    a = list(v)
    m = 1
    while m < n:
        wm = pow(w, n // (2 * m), p)
        for k in range(0, n, 2 * m):
            t = 1
            for j in range(m):
                u = a[k + j]
                x = a[k + j + m] * t % p
                a[k + j] = (u + x) % p
                a[k + j + m] = (u - x) % p
                t = t * wm % p
        m *= 2
    return a


class Domain:
    def __init__(self, F: PrimeField, size: int, group_gen: int):
        assert size & (size - 1) == 0 and size > 0
        self.F = F
        self.size = size
        self.log = size.bit_length() - 1
        self.gen = group_gen % F.p
        assert pow(self.gen, size, F.p) == 1 and (size == 1 or pow(self.gen, size // 2, F.p) != 1)
        self.gen_inv = pow(self.gen, -1, F.p)
        self.size_inv = pow(size, -1, F.p)

    @classmethod
    def snarkjs(cls, F: PrimeField, size: int):
        return cls(F, size, roots_of_unity(F)[1][size.bit_length() - 1])

    def fft_out_to_in(self, v: list) -> list:
        return _dit_from_bitreversed(self.F, v, self.gen)

    def ifft_in_to_out(self, v: list) -> list:
        # natural in -> natural coefficient vector -> bit-reversed storage
        p = self.F.p
        nat = _dit_from_bitreversed(self.F, bit_reverse(v), self.gen_inv)
        nat = [x * self.size_inv % p for x in nat]
        return bit_reverse(nat)

    def fft(self, v: list) -> list:
        """natural coefficients (zero-padded / must not exceed size) -> natural evaluations."""
        assert len(v) <= self.size
        v = list(v) + [0] * (self.size - len(v))
        return _dit_from_bitreversed(self.F, bit_reverse(v), self.gen)

    def ifft(self, v: list) -> list:
        assert len(v) <= self.size
        v = list(v) + [0] * (self.size - len(v))
        return bit_reverse(self.ifft_in_to_out(v))


def coset_powers(F: PrimeField, shift: int, size: int) -> list:
    out, cur = [], 1
    for _ in range(size):
        out.append(cur)
        cur = cur * shift % F.p
    return out


def bit_reversed_coset_table(F: PrimeField, shift: int, size: int) -> list:
    """reduction.rs:45-60."""
    return bit_reverse(coset_powers(F, shift, size))


def eval_poly_at(F: PrimeField, coeffs: list, x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % F.p
    return acc
