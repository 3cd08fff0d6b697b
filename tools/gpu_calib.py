#!/usr/bin/env python3
"""Known-bytes launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (run it under `rocprofv3 --pmc ...`):
one lane = one record of 64 / 96 / 128 / 192 bytes from a 2-6 GiB table (>> the 256 MiB Infinity Cache), hashed index (the
base gather of k_msm_accum) and own index (coalesced streaming). Prints one JSON line per launch with the known traffic; the
kernel names k_gather_calib<REC, SEQ> identify the rows in the PMC database (tools/pmc_summary.py reads both)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosnarks_amd as hip
from cosnarks_amd import bindings as B

L = hip.lib()
for rec in (64, 96, 128, 192):
    for seq in (0, 1):
        ms, rd, wr = C.c_float(0), C.c_double(0), C.c_double(0)
        B._check(L.csh_microbench_gather(rec, 25, 24, seq, C.byref(ms), C.byref(rd), C.byref(wr)))
        print(json.dumps({"kernel": f"k_gather_calib<{rec}, {'true' if seq else 'false'}>", "rec_bytes": rec, "sequential": bool(seq), "ms": round(ms.value, 4),
                          "bytes_read": rd.value, "bytes_written": wr.value, "GBps": round(rd.value / ms.value / 1e6, 1)}), flush=True)
