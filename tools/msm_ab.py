#!/usr/bin/env python3
"""Interleaved A/B of MSM tune settings in ONE process (single-shot comparisons drifted by +-7 % inside one log).
Every round runs every variant once, alternating the order; a measurement = REPS back-to-back synchronous csh_msm_dev calls (wall
clock, host fold included). Reported: median / min per variant and the median of the PAIRED differences against the first variant;
every variant's result is compared bit for bit with the first variant's.
    python tools/msm_ab.py --job 0:0:20 --rounds 12 --reps 10 c15=msm_c=15 c16=msm_c=16"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+", help="name=key=value[,key=value...]")
ap.add_argument("--job", default="0:0:20")
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
variants = []
for v in args.variants:
    name, rest = v.split("=", 1)
    variants.append((name, {kv.split("=")[0]: int(kv.split("=")[1], 0) for kv in rest.split(",")}))
L = hip.lib()
curve, group, logn = (int(x) for x in args.job.split(":"))
n = 1 << logn
pb = hip.point_bytes(curve, group)
buf = hip.DeviceBuffer(n * pb)
B._check(L.csh_util_generate_bases_dev(curve, group, C.c_uint64(1), C.c_size_t(n), buf.ptr, None))
B.sync()
h = C.c_void_p()
B._check(L.csh_bases_upload_dev(curve, group, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
buf.free()
rs = np.random.RandomState(1)
limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
limbs[:, 3] >>= np.uint64(3)
sc = hip.DeviceBuffer.from_host(limbs)
out = np.zeros(3 * pb // 16, dtype=np.uint64)
call = lambda: B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, 1, out.ctypes.data_as(C.c_void_p), None))
keys = sorted({k for _, kv in variants for k in kv})
defaults = {k: B.tune_get(k) for k in keys}


def apply(kv):
    for k in keys:
        B.tune_set(k, kv.get(k, defaults[k]))


ref = None
params = {}
for name, kv in variants:
    apply(kv)
    call()
    params[name] = B.msm_last_params()
    if ref is None:
        ref = out.copy()
    else:
        print(json.dumps({"variant": name, "equals_first_variant": bool(np.array_equal(out, ref))}), flush=True)
for _ in range(30):       # bring the clocks up
    call()
times = {name: [] for name, _ in variants}
for rnd in range(args.rounds + 1):
    for name, kv in (variants if rnd % 2 == 0 else variants[::-1]):
        apply(kv)
        call()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            call()
        dt = (time.perf_counter() - t0) / args.reps * 1e3
        if rnd:
            times[name].append(dt)
apply({})
base = variants[0][0]
for name, kv in variants:
    t = times[name]
    row = {"variant": name, "tune": kv, "job": args.job, "params_c_W_L_S": params[name], "rounds": args.rounds, "reps": args.reps,
           "ms_median": round(statistics.median(t), 4), "ms_min": round(min(t), 4), "Mpts_s_median": round(n / statistics.median(t) / 1e3, 1)}
    if name != base:
        d = [(a - b) / b for a, b in zip(t, times[base])]
        row["paired_delta_vs_first_pct_median"] = round(100 * statistics.median(d), 2)
        row["paired_delta_vs_first_pct_min_max"] = [round(100 * min(d), 2), round(100 * max(d), 2)]
    print(json.dumps(row), flush=True)
