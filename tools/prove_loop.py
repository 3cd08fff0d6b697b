#!/usr/bin/env python3
"""A few device-resident synthetic proves back to back (for rocprofv3 --kernel-trace + tools/prof_timeline.py: where one prove's ~80 launches sit).
    python tools/prove_loop.py [log_n] [proves]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g16

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = g16.glib()
h = C.c_void_p()
assert L.cog16_synth_open(hip.BN254, logn, C.byref(h)) == 0, L.cog16_last_error()
ph = (C.c_double * 3)()
for i in range(n):
    assert L.cog16_synth_prove(h, ph, None) == 0, L.cog16_last_error()
    print("prove", i, [round(x, 3) for x in ph], flush=True)
ok = C.c_int(0)
L.cog16_synth_check(h, C.byref(ok))
print("check", ok.value)
L.cog16_synth_close(h)
