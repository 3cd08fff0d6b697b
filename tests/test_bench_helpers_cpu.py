"""Host-side helpers of bench.py that decide what the timed workload IS: the rejection sampler of the scalars (the limb
comparison mixed two encodings; wrong whenever the upper limbs tie) against plain big-integer comparison, ties forced."""
import importlib.util
import os
import random

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _tensor(vals):
    rows = []
    for v in vals:
        limbs = [(v >> (64 * i)) & (2**64 - 1) for i in range(4)]
        rows.append([x - (1 << 64) if x >= (1 << 63) else x for x in limbs])
    return torch.tensor(rows, dtype=torch.int64)


def test_ge_modulus_matches_big_integer_comparison_with_tied_upper_limbs():
    r = random.Random(5)
    for cid, m in bench.SCALAR_MODULUS.items():
        vals = [0, 1, m - 1, m, m + 1, (1 << m.bit_length()) - 1]
        for k in range(4):                       # tie the limbs above k with the modulus, vary limb k around the modulus limb
            top = (m >> (64 * (k + 1))) << (64 * (k + 1))
            ml = (m >> (64 * k)) & (2**64 - 1)
            for limb in {0, 1, ml - 1 if ml else 0, ml, min(ml + 1, 2**64 - 1), (1 << 63) - 1, 1 << 63, (1 << 63) + 1, 2**64 - 1}:
                low = r.getrandbits(64 * k) if k else 0
                vals.append(top | (limb << (64 * k)) | low)
                vals.append(top | (limb << (64 * k)) | ((m & ((1 << (64 * k)) - 1)) if k else 0))
        vals += [r.getrandbits(m.bit_length()) for _ in range(200)]
        vals = [v for v in vals if v < (1 << 255)]
        got = bench.ge_modulus(torch, _tensor(vals), m).tolist()
        assert got == [v >= m for v in vals], cid


def test_uniform_scalars_are_below_the_modulus_and_use_the_full_range():
    m = bench.SCALAR_MODULUS[0]
    sc = bench.uniform_scalars(torch, torch.device("cpu"), 4096, m, 7)
    vals = [sum(((int(x) + (1 << 64)) % (1 << 64)) << (64 * i) for i, x in enumerate(row)) for row in sc.tolist()]
    assert all(0 <= v < m for v in vals) and max(vals) > (m * 15) // 16 and len(set(vals)) == len(vals)
    assert any(((v >> 63) & 1) for v in vals) and any((v >> 252) & 1 for v in vals)        # bit 63 of the lower limbs is random too
