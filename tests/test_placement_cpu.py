"""Host-only checks of the prover's device placement plan (ProvingKey::place, co-snarks_amd/host/types.hpp): which GPU takes which
of the five query MSMs of one proof (the independent closures of rayon_join5, groth16.rs:227-294). No device needed."""
import ctypes as C

import pytest

Q_A, Q_B1, Q_B2, Q_L, Q_H = range(5)


def _plan(sizes, nslots, mode, n_range=0):
    from cosnarks_amd import groth16 as g
    L = g.glib()
    L.cog16_placement_plan.restype = C.c_int
    sz = (C.c_size_t * 5)(*sizes)
    slots = (C.c_int * 5)()
    ranges = (C.c_size_t * (2 * nslots))()
    eff = L.cog16_placement_plan(sz, nslots, mode, C.c_size_t(n_range), slots, ranges)
    return eff, list(slots), [(ranges[2 * i], ranges[2 * i + 1]) for i in range(nslots)]


def test_by_query_is_longest_processing_time_first_with_g2_weighted():
    n = 1 << 20
    sizes = [n + 3, n + 3, n + 3, n + 1, n]            # a, b_g1, b_g2, l, h
    eff, slots, _ = _plan(sizes, 2, 0)                 # AUTO with two GPUs = by query
    assert eff == 1
    g2_side = slots[Q_B2]
    assert slots[Q_H] == g2_side and {slots[Q_A], slots[Q_B1], slots[Q_L]} == {1 - g2_side}     # {b_g2, h} / {a, b_g1, l}: 3.5 vs 3 units
    assert g2_side == 0                                # the heaviest query stays on the home GPU: its scalars need no copy
    eff, slots, _ = _plan(sizes, 5, 1)
    assert eff == 1 and sorted(slots) == [0, 1, 2, 3, 4]                                         # one query per GPU
    eff, slots, _ = _plan(sizes, 8, 1)
    assert sorted(slots) == [0, 1, 2, 3, 4]                                                      # three GPUs stay idle by query
    eff, slots, _ = _plan(sizes, 3, 1)
    loads = [0.0, 0.0, 0.0]
    for q, s in enumerate(slots):
        loads[s] += (2.5 if q == Q_B2 else 1.0)
    assert max(loads) == 2.5 and sorted(loads) == [2.0, 2.0, 2.5]
    eff, slots, _ = _plan([0, 0, 0, 0, 0], 4, 1)
    assert slots == [0] * 5


@pytest.mark.parametrize("n,ns", [(1 << 20, 3), ((1 << 20) + 5, 8), (7, 8), (0, 4), (1000003, 64)])
def test_by_range_partitions_every_index_once(n, ns):
    eff, slots, ranges = _plan([n] * 5, ns, 0 if ns >= 3 else 2, n_range=n)
    assert eff == 2 and slots == [0] * 5
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    for (lo, hi), (lo2, _) in zip(ranges, ranges[1:]):
        assert lo <= hi == lo2
    lens = [hi - lo for lo, hi in ranges]
    assert sum(lens) == n and max(lens) - min(lens) <= 1


def test_one_slot_and_bad_arguments():
    eff, slots, ranges = _plan([5] * 5, 1, 0, n_range=9)
    assert eff == 1 and slots == [0] * 5 and ranges == [(0, 9)]
    from cosnarks_amd import groth16 as g
    L = g.glib()
    assert L.cog16_placement_plan(None, 2, 0, C.c_size_t(0), None, None) == -1
    assert L.cog16_placement_plan((C.c_size_t * 5)(), 2, 7, C.c_size_t(0), (C.c_int * 5)(), None) == -1
