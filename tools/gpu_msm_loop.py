#!/usr/bin/env python3
"""Lean MSM loop for profiling (no torch): known-dlog bases + uniform scalars resident in HBM, `reps` csh_msm_dev calls per
(curve, group, log n) job. Prints stage timings (HIP events) per job. Usage: gpu_msm_loop.py [--reps 5] curve:group:logn ..."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

L = hip.lib()
args = sys.argv[1:]
reps = 5
if args and args[0] == "--reps":
    reps = int(args[1])
    args = args[2:]
timing = not os.environ.get("LOOP_NO_TIMING")
for job in args:
    curve, group, logn = (int(x) for x in job.split(":"))
    n = 1 << logn
    pb = hip.point_bytes(curve, group)
    buf = hip.DeviceBuffer(n * pb)
    B._check(L.csh_util_generate_bases_dev(curve, group, C.c_uint64(1), C.c_size_t(n), buf.ptr, None))
    B.sync()
    h = C.c_void_p()
    B._check(L.csh_bases_upload_dev(curve, group, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
    buf.free()
    if os.environ.get("LOOP_PRECOMPUTE"):   # fixed-base tables on the handle: "c" or "c:rows" (c = 0 / auto: automatic width)
        spec = os.environ["LOOP_PRECOMPUTE"].replace("auto", "0").split(":")
        if len(spec) > 1 and int(spec[1]) > 1:
            B._check(L.csh_bases_precompute_grouped(h, int(spec[0]), int(spec[1])))
        else:
            B._check(L.csh_bases_precompute(h, int(spec[0])))
    rs = np.random.RandomState(1)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    if os.environ.get("LOOP_SKEW"):   # witness-like skew: a quarter zero, two repeated values a quarter each (the scalars are passed as
        # Montgomery encodings, so the limbs {1, 0, 0, 0} decode to a full-range value: one heavy bucket per window for each repeated value;
        # bench.py's witness_like line and tests/test_gpu_fullsize.py use true ones)
        kind = rs.randint(0, 4, size=n)
        limbs[kind == 0] = 0
        limbs[kind == 1] = 0
        limbs[kind == 1, 0] = 1
        limbs[kind == 2] = limbs[0]
    sc = hip.DeviceBuffer.from_host(limbs)
    out = np.zeros(3 * pb // 16, dtype=np.uint64)
    if timing:
        B.tune_set("msm_timing", 1)
    best = None
    wall = []
    import time
    for _ in range(reps):
        t0 = time.perf_counter()
        B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, 1, out.ctypes.data_as(C.c_void_p), None))
        wall.append((time.perf_counter() - t0) * 1e3)
        t = B.msm_last_timing()
        if best is None or t[5] < best[5]:
            best = t
    B.tune_set("msm_timing", 0)
    wall_ms = min(wall[1:]) if len(wall) > 1 else wall[0]      # the synchronous call incl. the host fold (first call warms up)
    print(json.dumps({"curve": curve, "group": group, "logn": logn, "params_c_W_L_S": B.msm_last_params(),
                      "ms_digits_scan_scatter_accum_reduce_total": [round(x, 3) for x in best] if timing else None,
                      "wall_ms": round(wall_ms, 3), "Mpts_s_wall": round(n / wall_ms / 1e3, 1)}), flush=True)
    L.csh_bases_free(h)
    sc.free()
