#!/usr/bin/env python3
"""GPU probe: integer-pipe micro-benchmarks + MSM stage timings / window sweep. Writes JSON to stdout."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

L = hip.lib()
res = {"microbench_Gops": {}, "msm": []}
names = {0: "v_mad_u64_u32", 8: "v_mad_i64_i32", 1: "v_lshl_add_u64", 6: "v_ashrrev_i64", 2: "v_add_co+addc_u32", 7: "v_alignbit+ashr_i32",
         3: "v_mul_lo_u32", 4: "v_add_u32", 5: "v_mul_hi_u32", 13: "v_fma_f64",
         10: "modmul_bn254_fq_32x8_cios", 11: "modmul_bn254_fq_29x9_lazy", 12: "modmul_bls381_fq_32x12_cios",
         14: "modmul_bn254_fq_29x9_signed_scan"}
for kind, nm in ([] if os.environ.get("PROBE_SKIP_UB") else names.items()):
    v = C.c_double(0)
    iters = 2000 if (kind < 10 or kind == 13) else 200
    B._check(L.csh_microbench(kind, iters, C.byref(v)))
    res["microbench_Gops"][nm] = round(v.value / 1e9, 2)
print(json.dumps(res["microbench_Gops"]), flush=True)

CURVE = int(os.environ.get("PROBE_CURVE", "0"))
GROUP = int(os.environ.get("PROBE_GROUP", "0"))
PB = hip.point_bytes(CURVE, GROUP)
if os.environ.get("PROBE_SKIP_UB"):
    pass
sizes = [int(x) for x in os.environ.get("PROBE_LOGN", "20,24").split(",")]
cs = [int(x) for x in os.environ.get("PROBE_C", "0,13,14,15,16,17,18").split(",")]
B.tune_set("msm_timing", 1)
for logn in sizes:
    n = 1 << logn
    buf = hip.DeviceBuffer(n * PB)
    B._check(L.csh_util_generate_bases_dev(CURVE, GROUP, C.c_uint64(1), C.c_size_t(n), buf.ptr, None))
    B.sync()
    h = C.c_void_p()
    B._check(L.csh_bases_upload_dev(CURVE, GROUP, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h)))
    if os.environ.get("PROBE_TABLE"):
        import time as _t
        _t0 = _t.perf_counter()
        B._check(L.csh_bases_precompute(h, int(os.environ["PROBE_TABLE"]) if os.environ["PROBE_TABLE"].isdigit() else 0))
        print(json.dumps({"precompute_ms": round((_t.perf_counter() - _t0) * 1e3, 1), "logn": logn}), flush=True)
    buf.free()
    rs = np.random.RandomState(1)
    limbs = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] >>= np.uint64(3)
    sc = hip.DeviceBuffer.from_host(limbs)
    out = np.zeros(3 * PB // 16, dtype=np.uint64)
    for c in cs:
        B.tune_set("msm_c", c)
        best = None
        for rep in range(3):
            B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, 1, out.ctypes.data_as(C.c_void_p), None))
            t = B.msm_last_timing()
            if best is None or t[5] < best[5]:
                best = t
        row = {"curve": CURVE, "group": GROUP, "logn": logn, "c": c, "ms": [round(x, 3) for x in best], "Mpts_s": round(n / best[5] / 1e3, 1)}
        res["msm"].append(row)
        print(json.dumps(row), flush=True)
    L.csh_bases_free(h)
    sc.free()
