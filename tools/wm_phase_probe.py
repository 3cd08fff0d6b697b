#!/usr/bin/env python3
"""The host-facing Rep3 witness map (csh_groth16_witness_map_masks / csh_groth16_witness_map with seeds) straight on the C ABI, no
mirror: 2^k constraints, one entry per row, witness shares + two mask vectors as host arrays that are either kept (persistent) or
allocated and filled anew for every call (what a prover's Vec<F> does). Phase times from the library's own timers (tune host_timing)."""
import ctypes as C
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

L = hip.lib()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
nc = n - 2
r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
gen = np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
dom = hip.Domain(hip.BN254, logn, gen)
one = np.array([((1 << 256) % r >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
row_ptr = np.arange(nc + 1, dtype=np.uint64)
col = (np.arange(nc, dtype=np.uint32) % np.uint32(nc)) + np.uint32(1)
val = np.tile(one, nc)
ma, mb = C.c_void_p(), C.c_void_p()
for m in (ma, mb):
    B._check(L.csh_matrix_upload(0, row_ptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p), C.c_size_t(nc), C.c_size_t(nc), C.byref(m)))
rs = np.random.RandomState(3)
pub = rs.randint(0, 1 << 62, size=(1, 4), dtype=np.uint64)
shift = np.array([7, 0, 0, 0], dtype=np.uint64)
keys = ("stat_wm_h2d_us", "stat_wm_dev_us", "stat_wm_d2h_us")
p = lambda a: a.ctypes.data_as(C.c_void_p)


def fresh(shape):
    a = np.empty(shape, dtype=np.uint64)
    a[:] = 5
    return a


def run(mode, persistent, reps=12):
    wit = fresh((nc, 8)); mc = fresh((n, 4)); mab = fresh((n, 4)); h = np.empty((n, 4), dtype=np.uint64)
    seed = (C.c_uint8 * 32)(*range(32))
    wall, z = [], {k: B.tune_get(k) for k in keys}
    for it in range(reps + 2):
        if it == 2:
            z = {k: B.tune_get(k) for k in keys}
        if not persistent:
            wit = fresh((nc, 8)); h = np.empty((n, 4), dtype=np.uint64)
            if mode == "masks":
                mc = fresh((n, 4)); mab = fresh((n, 4))
        t0 = time.perf_counter()
        if mode == "masks":
            B._check(L.csh_groth16_witness_map_masks(dom.h, p(shift), 1, 0, ma, mb, C.c_size_t(nc), p(pub), C.c_size_t(1), p(wit), C.c_size_t(nc), p(mc), p(mab), p(h)))
        else:
            B._check(L.csh_groth16_witness_map(dom.h, p(shift), 1, 0, ma, mb, C.c_size_t(nc), p(pub), C.c_size_t(1), p(wit), C.c_size_t(nc), seed, C.c_uint64(0), seed, C.c_uint64(0), p(h)))
        if it >= 2:
            wall.append((time.perf_counter() - t0) * 1e3)
    d = {k: (B.tune_get(k) - z[k]) / reps / 1e3 for k in keys}
    print(json.dumps({"mode": mode, "persistent_buffers": persistent, "log_n": logn, "wall_ms_median": round(statistics.median(wall), 3), "wall_ms_min": round(min(wall), 3),
                      "phase_ms_avg": {k[8:-3]: round(v, 3) for k, v in d.items()}}), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "pair":      # A/B of the fused tile passes (tune ntt_pair), device phase from the library's timers
    B.tune_set("host_timing", 1)
    for rnd in range(3):
        for pair in (1, 0):
            B.tune_set("ntt_pair", pair)
            print(json.dumps({"ntt_pair": pair, "round": rnd}), end=" ")
            run("seeds", True, reps=20)
    B.tune_set("ntt_pair", 1)
    sys.exit(0)
for timing in (0, 1):
    B.tune_set("host_timing", timing)
    for mode in ("seeds", "masks"):
        for persistent in (True, False):
            run(mode, persistent)
