#!/usr/bin/env python3
"""Back-to-back 2^logn BN254 transforms for rocprofv3 (kernel trace / PMC passes): WARM untimed transform pairs, then PAIRS timed pairs.
    python tools/ntt_loop.py [--logn 22] [--ncomp 1] [--pairs 10] [--warm 40] [--variant 0x0]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=22)
ap.add_argument("--ncomp", type=int, default=1)
ap.add_argument("--pairs", type=int, default=10)
ap.add_argument("--warm", type=int, default=40)
ap.add_argument("--variant", type=lambda x: int(x, 0), default=0)
a = ap.parse_args()
r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - a.logn), r) * ((1 << 256) % r) % r
gen = np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
dom = hip.Domain(hip.BN254, a.logn, gen)
rs = np.random.RandomState(1)
v = rs.randint(0, 1 << 62, size=((1 << a.logn) * a.ncomp, 4), dtype=np.uint64)
v[:, 3] >>= np.uint64(1)
d = hip.DeviceBuffer.from_host(v)
B.tune_set("ntt_variant", a.variant)
e0, e1 = B.Event(), B.Event()
for _ in range(a.warm):
    dom.ifft_in_to_out_dev(d, a.ncomp)
    dom.fft_out_to_in_dev(d, a.ncomp)
e0.record()
for _ in range(a.pairs):
    dom.ifft_in_to_out_dev(d, a.ncomp)
    dom.fft_out_to_in_dev(d, a.ncomp)
e1.record()
print(json.dumps({"op": "ntt 2^%d ncomp %d, %d pairs back to back after %d warm pairs" % (a.logn, a.ncomp, a.pairs, a.warm), "variant": hex(a.variant),
                  "avg_ms_per_transform": round(e0.elapsed_ms(e1) / (2 * a.pairs), 4)}), flush=True)
