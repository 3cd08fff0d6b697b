"""world_size-2/3/4 gloo tests of the split-MSM exchange (the N > 1 path of bench.py): each rank builds the partial
buffer of its point range (here from the oracle, standing in for csh_msm_partial_dev), the partials are
all-gathered and folded by the product's host fold; the result equals the full MSM."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n=24):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cosnarks_amd as hip
    from cosnarks_amd.distributed import allgather_and_fold
    from oracle import curves as cv
    from tests import helpers as H
    G, F = cv.BN254_G1, H.FR["bn254"]
    r = H.rng(5)                                    # same seed on every rank: identical global inputs
    c = 7
    pts = H.rand_points(G, n, r, with_inf=True)
    sc = H.rand_elems(F, n, r)
    lo, hi = rank * n // world, (rank + 1) * n // world
    # partial of this rank's range as W window sums: sum_i digit_w(s_i) * P_i (unsigned c-bit digits)
    W = (254 + c - 1) // c
    wins = []
    for w in range(W):
        acc = None
        for i in range(lo, hi):
            d = (sc[i] >> (w * c)) & ((1 << c) - 1)
            acc = G.add(acc, G.mul(pts[i], d))
        wins.append(acc)
    nbytes = hip.msm_partial_bytes(hip.BN254, hip.G1)
    buf = np.zeros(nbytes, dtype=np.uint8)
    buf[:16] = np.array([0x4D534D50, c, W, 0], dtype="<u4").view(np.uint8)
    one = H.pack(G.F, [1])
    for w, P in enumerate(wins):
        if P is None:
            continue                                  # XYZZ infinity = all zero
        xyzz = np.concatenate([cv.pack_points(G, [P]).reshape(-1), one, one]).view(np.uint8)
        buf[32 + 128 * w:32 + 128 * (w + 1)] = xyzz
    out = allgather_and_fold(torch.from_numpy(buf), hip.BN254, hip.G1, world, dist)
    got = H.jac_to_affine(G, out)
    q.put((rank, G.eq(got, G.msm(pts, sc))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 24), (3, 25), (4, 3)])
def test_split_msm_allgather_fold_gloo(world, n):
    """(2, 24): even halves; (3, 25): uneven contiguous ranges 8/8/9; (4, 3): more ranks than points -- one rank holds an
    empty range and contributes the all-infinity partial."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]
