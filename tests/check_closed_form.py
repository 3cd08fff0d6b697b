"""Closed-form check for MSMs over the known-dlog synthetic bases (SURVEY 8d, config 2 family ii).

bases[i] = k_i * G with k_i = splitmix64(seed + i) | 1 (csh_util_generate_bases_dev), so
MSM(bases, s) = (sum_i s_i k_i mod r) * G -- O(n) integer work at any n. Checker only (uses the oracle)."""
import numpy as np

from oracle import curves as cv
from tests import helpers as H


def splitmix64_np(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def dlogs(seed: int, n: int, start: int = 0):
    with np.errstate(over="ignore"):
        idx = (np.arange(start, start + n, dtype=np.uint64) + np.uint64(seed & ((1 << 64) - 1)))
        return splitmix64_np(idx) | np.uint64(1)


def weighted_sum(limbs_u64, ks) -> int:
    """sum_i value(limbs[i]) * k_i as an exact Python integer (limbs: (n,4) u64, little-endian)."""
    limbs = np.ascontiguousarray(limbs_u64).view(np.uint64).reshape(-1, 4)
    total = 0
    # split everything in 32-bit halves so numpy u64 products/sums cannot overflow: chunks of 2^16 rows
    klo, khi = ks & np.uint64(0xFFFFFFFF), ks >> np.uint64(32)
    for l in range(4):
        col = limbs[:, l]
        for (a, sa) in ((col & np.uint64(0xFFFFFFFF), 0), (col >> np.uint64(32), 32)):
            for (b, sb) in ((klo, 0), (khi, 32)):
                prod = a * b                                 # < 2^64
                s = 0
                for c in range(0, prod.size, 1 << 16):
                    ch = prod[c:c + (1 << 16)]
                    s += int((ch & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64)) + (int((ch >> np.uint64(32)).sum(dtype=np.uint64)) << 32)
                total += s << (64 * l + sa + sb)
    return total


def closed_form_point(curve_name: str, group: int, seed: int, n: int, scalar_limbs, montgomery: bool):
    G = cv.CURVES[curve_name][group]
    F = H.FR[curve_name]
    S = weighted_sum(scalar_limbs, dlogs(seed, n)) % F.p
    if montgomery:
        S = S * F.Rinv % F.p
    return G.mul(G.gen, S)


def closed_form_ok(hip, L, seed, n, scalar_limbs_i64, jac_out) -> bool:
    G = cv.BN254_G1
    want = closed_form_point("bn254", 0, seed, n, np.asarray(scalar_limbs_i64).view(np.uint64), True)
    return G.eq(H.jac_to_affine(G, jac_out), want)


def local_dlog_sum(seed, n, scalar_limbs_i64) -> int:
    """sum_i s_i k_i mod r of one rank's share of a split MSM (scalars in Montgomery form, the factor R is removed by
    closed_form_ok_split)."""
    F = H.FR["bn254"]
    return weighted_sum(np.asarray(scalar_limbs_i64).view(np.uint64), dlogs(seed, n)) % F.p


def closed_form_ok_split(partial_sums, jac_out) -> bool:
    """Split MSM over several ranks: the folded result must equal (sum over ranks of their local sums) * G."""
    G = cv.BN254_G1
    F = H.FR["bn254"]
    S = sum(partial_sums) % F.p * F.Rinv % F.p
    return G.eq(H.jac_to_affine(G, jac_out), G.mul(G.gen, S))

