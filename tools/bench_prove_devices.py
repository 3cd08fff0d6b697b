#!/usr/bin/env python3
"""One plain Groth16 prover with its five query MSMs placed on several GPUs (no torch): prints one JSON line.

    python tools/bench_prove_devices.py --devices 0,1,2,3 [--log-n 20] [--steps 5] [--warmup 2]

`--devices 0` (one entry) is the single-GPU reference; a GPU may be listed more than once (logical slots on one device)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosnarks_amd as hip
from cosnarks_amd import bindings as B
from cosnarks_amd import groth16 as g16

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="0")
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--curve", type=int, default=0)
ap.add_argument("--mode", type=int, default=0, help="0 auto, 1 whole queries per GPU, 2 ranges of every query per GPU")
a = ap.parse_args()
devs = [int(x) for x in a.devices.split(",") if x != ""]
B._check(hip.lib().csh_init(devs[0]))
g16.set_prover_devices(devs if len(devs) > 1 else None, a.mode)
c = g16.SynthCircuit(a.curve, a.log_n)
for _ in range(a.warmup):
    c.prove()
ts, ph = [], []
for _ in range(a.steps):
    t0 = time.perf_counter()
    ph.append(c.prove())
    ts.append((time.perf_counter() - t0) * 1e3)
ok = c.check()
med = lambda xs: sorted(xs)[len(xs) // 2]
print(json.dumps({"devices": devs, "mode": a.mode, "log_n": a.log_n, "prove_ms_median": med(ts), "prove_ms_min": min(ts),
                  "phases_ms_median": {k: med([p[k] for p in ph]) for k in ("witness_upload_and_map", "msm_groups", "finish")},
                  "key_setup_ms": ph[0]["key_setup_ms"], "closed_form_check": ok}))
c.close()
