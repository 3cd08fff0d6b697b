#!/usr/bin/env python3
"""Why does bench.py report the 2^22 transform at 0.53-0.56 ms when tools/ntt_ab.py times the same kernels at 0.45 ms on the same pool?
Same process, same domain: (a) cold, library buffer; (b) torch tensor as the buffer; (c) right after one second of 2^24 MSMs (clock /
power state); (d) after the multiply-add microbenchmarks bench.py runs before its NTT line."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

L = hip.lib()
logn = 22
r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
gen = np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
dom = hip.Domain(hip.BN254, logn, gen)
e0, e1 = B.Event(), B.Event()


def t_alt(ptr, reps=10):
    dom.ifft_in_to_out_dev(ptr, 1, None)
    e0.record()
    for _ in range(reps):
        dom.ifft_in_to_out_dev(ptr, 1, None)
        dom.fft_out_to_in_dev(ptr, 1, None)
    e1.record()
    return round(e0.elapsed_ms(e1) / (2 * reps), 4)


rs = np.random.RandomState(1)
host = rs.randint(0, 1 << 62, size=(1 << logn, 4), dtype=np.uint64)
host[:, 3] >>= np.uint64(1)
buf = hip.DeviceBuffer.from_host(host)
print(json.dumps({"case": "cold, hipMalloc buffer", "ms": [t_alt(buf) for _ in range(4)]}), flush=True)
dev = torch.device("cuda:0")
data = torch.randint(0, 1 << 62, (1 << logn, 4), dtype=torch.int64, device=dev)
data[:, 3] >>= 1
torch.cuda.synchronize()
print(json.dumps({"case": "torch tensor as the buffer", "ms": [t_alt(data.data_ptr()) for _ in range(4)]}), flush=True)
print(json.dumps({"case": "hipMalloc buffer again", "ms": [t_alt(buf) for _ in range(4)]}), flush=True)
# one second of heavy multiply-add load: 2^22-point MSMs
n = 1 << 22
pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
B._check(L.csh_util_generate_bases_dev(0, 0, C.c_uint64(99), C.c_size_t(n), C.c_void_p(pts.data_ptr()), None))
h = C.c_void_p()
B._check(L.csh_bases_upload_dev(0, 0, C.c_void_p(pts.data_ptr()), C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
sc = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev)
out = np.zeros(12, dtype=np.uint64)
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 1.5:
    B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), C.c_void_p(sc.data_ptr()), 1, out.ctypes.data_as(C.c_void_p), None))
    k += 1
print(json.dumps({"case": "%d MSMs of 2^22 points in 1.5 s, then immediately:" % k, "ms": [t_alt(buf) for _ in range(6)]}), flush=True)
time.sleep(1.0)
print(json.dumps({"case": "after 1 s idle", "ms": [t_alt(buf) for _ in range(3)]}), flush=True)
for kind in (0, 8, 11, 14):
    outv = C.c_double(0)
    try:
        for _ in range(3):
            B._check(L.csh_microbench(kind, 200, C.byref(outv)))
    except Exception as e:  # noqa: BLE001
        print("microbench", kind, e)
print(json.dumps({"case": "after the csh_microbench chains", "ms": [t_alt(buf) for _ in range(4)]}), flush=True)
