#!/usr/bin/env python3
"""Lane-length sweep of the MSM accumulate stage: for each job curve:group:logn (or curve:group:n=<points>) times csh_msm_dev with msm_l
forced to every value of --ls (default: the planner's choice first, then a grid) and prints accumulate / tail / total per L.
The accumulate kernel runs ceil(waves / SIMDs) rounds of L additions, so its time is a sawtooth in L; this is the measurement the
planner's cost model (csrc/msm_impl.hpp msm_plan) is calibrated and checked against."""
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
ls = None
reps = 6
while args and args[0].startswith("--"):
    if args[0] == "--ls":
        ls = [int(x) for x in args[1].split(",")]
    elif args[0] == "--reps":
        reps = int(args[1])
    args = args[2:]
for job in args:
    curve, group, size = job.split(":")
    curve, group = int(curve), int(group)
    n = int(size[2:]) if size.startswith("n=") else 1 << int(size)
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
    for _ in range(max(20, int(0.5 / (2e-9 * n + 1e-3)))):          # ~0.5 s of warm-up: clocks ramp over the first calls
        B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, 1, out.ctypes.data_as(C.c_void_p), None))
    B.tune_set("msm_timing", 1)
    grid = ls if ls is not None else [0] + list(range(16, 65, 2)) + list(range(68, 129, 4)) + list(range(136, 257, 8)) + [320, 384, 448, 512, 640, 768, 1024]
    acc = {}
    for order in (grid, grid[::-1]):                                 # two passes, opposite order; the minimum per L counts
        for fl in order:
            B.tune_set("msm_l", fl)
            for _ in range(reps):
                B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, 1, out.ctypes.data_as(C.c_void_p), None))
                t = B.msm_last_timing()
                p = B.msm_last_params()
                if fl not in acc or t[5] < acc[fl][0][5]:
                    acc[fl] = (t, p[2])
    rows = [{"forced": fl, "L": acc[fl][1], "accum": round(acc[fl][0][3], 3), "tail": round(acc[fl][0][4], 3), "total": round(acc[fl][0][5], 3)} for fl in grid]
    B.tune_set("msm_l", 0)
    B.tune_set("msm_timing", 0)
    auto = [r for r in rows if r["forced"] == 0]
    bestrow = min(rows, key=lambda r: r["total"])
    print(json.dumps({"curve": curve, "group": group, "n": n, "params_c_W": B.msm_last_params()[:2], "auto": auto, "best": bestrow,
                      "auto_vs_best": round(min(a["total"] for a in auto) / bestrow["total"], 4) if auto else None, "rows": rows}), flush=True)
    L.csh_bases_free(h)
    sc.free()
