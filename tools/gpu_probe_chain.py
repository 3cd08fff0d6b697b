#!/usr/bin/env python3
"""Dependent-chain probe of the 64-bit multiply-add pipe: lane-ops/s of v_mad_i64_i32 with 1 / 2 / 4 independent chains per lane at
1..8 waves per SIMD (csh_microbench_chain). The pinned product-scanning multiplication gives a lane ONE chain."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosnarks_amd as hip
from cosnarks_amd import bindings as B

L = hip.lib()
rows = []
for chains in (1, 2, 4):
    for w in (1, 2, 3, 4, 6, 8):
        v = C.c_double(0)
        best = 0.0
        for _ in range(3):
            B._check(L.csh_microbench_chain(chains, w, 400, C.byref(v)))
            best = max(best, v.value)
        rows.append({"chains_per_lane": chains, "waves_per_simd": w, "Tmad_s": round(best / 1e12, 2)})
        print(json.dumps(rows[-1]), flush=True)
