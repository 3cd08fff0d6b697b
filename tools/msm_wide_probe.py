#!/usr/bin/env python3
"""Fixed-base tables with ONE bucket set and windows of 17 .. 22 bits (msm_sort_wide.hip) against the plain handle, same process, same
bases and scalars: every table result is compared bit for bit with the plain result; warm medians of synchronous csh_msm_dev calls
(host fold included) and the stage times (level 1, bucket histogram + scan, level 2, accumulate, tail, total) of the best of five.
    python tools/msm_wide_probe.py [--c 17 18 19 20] [--lb 8 9 10] [--reps 30] curve:group:logn ..."""
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
ap.add_argument("jobs", nargs="+")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--warm", type=int, default=20)
ap.add_argument("--c", type=int, nargs="*", default=[17, 18, 19, 20])
ap.add_argument("--lb", type=int, nargs="*", default=[8])
ap.add_argument("--chunks", type=int, nargs="*", default=[0])
ap.add_argument("--seg", type=int, nargs="*", default=[0])
ap.add_argument("--l", type=int, nargs="*", default=[0], help="forced entries per accumulate lane (tune msm_l), 0 = the plan's")
ap.add_argument("--variant", type=int, nargs="*", default=[0], help="tune msm_variant values to run (bit 7 = 128: result through a copy instead of the direct write)")
ap.add_argument("--skewed", action="store_true", help="witness-like scalars: half in {0, 1}, a quarter equal")
ap.add_argument("--profile", action="store_true", help="kernel-trace runs: ONE plain call (the reference result), the rest on the table handle")
args = ap.parse_args()
L = hip.lib()


def measure(call, reps, warm):
    for _ in range(warm):
        call()
    wall = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        wall.append((time.perf_counter() - t0) * 1e3)
    B.tune_set("msm_timing", 1)
    best = None
    for _ in range(5):
        call()
        t = B.msm_last_timing()
        if best is None or t[5] < best[5]:
            best = t
    B.tune_set("msm_timing", 0)
    return statistics.median(wall), min(wall), [round(x, 3) for x in best]


for job in args.jobs:
    curve, group, logn = (int(x) for x in job.split(":"))
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
    mont = 1
    if args.skewed:
        limbs[: n // 4] = 0
        limbs[n // 4: n // 2] = 0
        limbs[n // 4: n // 2, 0] = 1
        limbs[n // 2: 3 * n // 4] = limbs[n // 2]
        mont = 0
    sc = hip.DeviceBuffer.from_host(limbs)
    out = np.zeros(3 * pb // 16, dtype=np.uint64)
    call = lambda: B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, mont, out.ctypes.data_as(C.c_void_p), None))
    if args.profile:
        call()
        med, mn, st = 1.0, 1.0, []
    else:
        med, mn, st = measure(call, args.reps, args.warm)
    ref = out.copy()
    print(json.dumps({"job": job, "mode": "plain", "params_c_W_L_S": B.msm_last_params(), "wall_ms_median": round(med, 4), "wall_ms_min": round(mn, 4),
                      "Mpts_s": round(n / med / 1e3, 1), "stage_ms": st}), flush=True)
    for c in args.c:
        t0 = time.perf_counter()
        rc = L.csh_bases_precompute(h, c)
        if rc != 0:
            print(json.dumps({"job": job, "c": c, "error": hip.last_error() if hasattr(hip, "last_error") else rc}), flush=True)
            continue
        B.sync()
        build_s = time.perf_counter() - t0
        for lb in args.lb:
            for chunks in args.chunks:
                for seg, ll, var in [(sg, l, v) for sg in args.seg for l in args.l for v in args.variant]:
                    B.tune_set("msm_variant", var)
                    B.tune_set("msm_wide_lb", lb)
                    B.tune_set("msm_wide_chunks", chunks)
                    B.tune_set("msm_seg_buckets", seg)
                    B.tune_set("msm_l", ll)
                    out[:] = 0
                    call()
                    same = bool(np.array_equal(out, ref))
                    med, mn, st = measure(call, args.reps, args.warm)
                    print(json.dumps({"job": job, "mode": "table", "c": c, "lb": lb, "chunks": chunks, "seg": seg, "l": ll, "variant": var, "equals_plain": same, "params_c_W_L_S": B.msm_last_params(),
                                      "wall_ms_median": round(med, 4), "wall_ms_min": round(mn, 4), "Mpts_s": round(n / med / 1e3, 1), "stage_ms": st,
                                      "table_build_s": round(build_s, 3)}), flush=True)
        B.tune_set("msm_wide_lb", 0)
        B.tune_set("msm_wide_chunks", 0)
        B.tune_set("msm_seg_buckets", 0)
        B.tune_set("msm_l", 0)
        B.tune_set("msm_variant", 0)
    L.csh_bases_drop_tables(h)
    L.csh_bases_free(h)
    sc.free()
