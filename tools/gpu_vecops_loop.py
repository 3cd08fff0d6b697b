#!/usr/bin/env python3
"""Back-to-back launches of the share-vector kernels and of one NTT round trip (for rocprofv3 kernel-trace averages that can be
compared with bench.py's HIP-event averages over the same back-to-back pattern)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

ap = argparse.ArgumentParser()
ap.add_argument("--spinup-s", type=float, default=0.5, help="untimed back-to-back work before each measured loop (the clocks of an idle GPU need "
                "a few hundred ms to come up: VERDICT r5 #2b -- the round-5 kernel-trace averages were ~20 transforms on a cold GPU)")
args = ap.parse_args()
L = hip.lib()


def spin(fn, seconds):
    """fn() back to back for `seconds` (synchronising every 20 calls)"""
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        B.sync()
        k += 20
    return k


rs = np.random.RandomState(1)
n = 1 << 24
a = hip.DeviceBuffer.from_host(rs.randint(0, 1 << 62, size=(2 * n, 4), dtype=np.uint64))
b = hip.DeviceBuffer.from_host(rs.randint(0, 1 << 62, size=(2 * n, 4), dtype=np.uint64))
m = hip.DeviceBuffer.from_host(rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64))
o = hip.DeviceBuffer(n * 32)
f = lambda: B._check(L.csh_rep3_local_mul_vec_dev(0, a.ptr, b.ptr, m.ptr, o.ptr, C.c_size_t(n), None))
for _ in range(3):
    f()
spun = spin(f, args.spinup_s)
e0, e1 = B.Event(), B.Event()
e0.record()
for _ in range(20):
    f()
e1.record()
print(json.dumps({"op": "rep3_local_mul_vec 2^24 x20 back to back", "avg_ms": round(e0.elapsed_ms(e1) / 20, 4)}), flush=True)
ts = []
for _ in range(10):      # isolated launches (event sync between them), as tools/gpu_probe_ntt.py times them
    e0.record()
    f()
    e1.record()
    ts.append(e0.elapsed_ms(e1))
print(json.dumps({"op": "rep3_local_mul_vec 2^24 isolated", "avg_ms": round(sum(ts) / len(ts), 4), "min_ms": round(min(ts), 4)}), flush=True)
logn = 22
r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
gen = np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
dom = hip.Domain(hip.BN254, logn, gen)
v = rs.randint(0, 1 << 62, size=(1 << logn, 4), dtype=np.uint64)
d = hip.DeviceBuffer.from_host(v)
dom.ifft_in_to_out_dev(d, 1)


def pair():
    dom.ifft_in_to_out_dev(d, 1)
    dom.fft_out_to_in_dev(d, 1)


spun_ntt = 2 * spin(pair, args.spinup_s)
e0.record()
for _ in range(100):
    pair()
e1.record()
print(json.dumps({"op": "ntt 2^22 x200 back to back", "avg_ms": round(e0.elapsed_ms(e1) / 200, 4), "untimed_transforms_before": spun_ntt,
                  "spinup_s": args.spinup_s}), flush=True)
