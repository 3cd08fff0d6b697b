#!/usr/bin/env python3
"""Warm MSM timings: the clocks of an idle GPU take tens of ms of continuous work to come up (profiles/archive/r04_j_ntt_context.log), so a handful of
calls after process start under-reports small sizes by 10-20 %. Per job: WARM untimed calls, then the median wall time of REPS synchronous
csh_msm_dev calls (result on the host, host fold included) and the stage times of the best one.
    python tools/msm_warm.py [--reps 60] [--warm 60] [--c C] curve:group:logn ..."""
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
ap.add_argument("--reps", type=int, default=60)
ap.add_argument("--warm", type=int, default=60)
ap.add_argument("--c", type=int, nargs="*", default=[0])
args = ap.parse_args()
L = hip.lib()
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
    sc = hip.DeviceBuffer.from_host(limbs)
    out = np.zeros(3 * pb // 16, dtype=np.uint64)
    call = lambda: B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, 1, out.ctypes.data_as(C.c_void_p), None))
    for c in args.c:
        B.tune_set("msm_c", c)
        for _ in range(args.warm):
            call()
        wall = []
        for _ in range(args.reps):
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
        med = statistics.median(wall)
        print(json.dumps({"curve": curve, "group": group, "logn": logn, "c_forced": c, "params_c_W_L_S": B.msm_last_params(), "wall_ms_median": round(med, 4),
                          "wall_ms_min": round(min(wall), 4), "Mpts_s": round(n / med / 1e3, 1),
                          "stage_ms_digits_scan_scatter_accum_tail_total": [round(x, 3) for x in best]}), flush=True)
    B.tune_set("msm_c", 0)
    L.csh_bases_free(h)
    sc.free()
