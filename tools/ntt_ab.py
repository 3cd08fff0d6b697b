#!/usr/bin/env python3
"""Interleaved A/B of NTT variants in ONE process (order / clock drift of +-7 % inside one log made single-shot
comparisons unresolvable). Every round runs every variant once (ABAB...), each measurement = REPS back-to-back transform pairs
(ifft_in_to_out + fft_out_to_in) between two HIP events; reported: median and min per variant over the rounds and the median of the
PAIRED differences against the first variant. Every variant's output is compared bit for bit with the first variant's on the same input.

    python tools/ntt_ab.py --logn 22 --ncomp 1 --rounds 12 --reps 10 0 0x100 0x200 ...     (values of tune "ntt_variant")
    a variant may be written  name=value[,key=value...]  to set several tune keys, e.g.  t10=ntt_variant=0x101
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B


def snarkjs_root(logn):
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
    return np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def parse_variant(txt):
    if "=" not in txt:
        return txt, {"ntt_variant": int(txt, 0)}
    name, rest = txt.split("=", 1)
    if "=" not in rest:
        return name, {"ntt_variant": int(rest, 0)}
    kv = {}
    for item in rest.split(","):
        k, v = item.split("=")
        kv[k] = int(v, 0)
    return name, kv


ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--logn", type=int, default=22)
ap.add_argument("--ncomp", type=int, default=1)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
variants = [parse_variant(v) for v in args.variants]
n = 1 << args.logn
rs = np.random.RandomState(7)
host = rs.randint(0, 1 << 62, size=(n * args.ncomp, 4), dtype=np.uint64)
host[:, 3] >>= np.uint64(1)   # canonical (< r)
dom = hip.Domain(hip.BN254, args.logn, snarkjs_root(args.logn))
buf = hip.DeviceBuffer.from_host(host)
e0, e1 = B.Event(), B.Event()


def apply(kv):
    for k, v in kv.items():
        B.tune_set(k, v)


def reset(kv):
    for k in kv:
        B.tune_set(k, 0 if k == "ntt_variant" else B.tune_get(k))


# correctness: every variant against the first, both directions
ref = None
for name, kv in variants:
    apply(kv)
    b2 = hip.DeviceBuffer.from_host(host)
    dom.ifft_in_to_out_dev(b2, args.ncomp)
    B.sync()
    inv = b2.to_host().copy()
    dom.fft_out_to_in_dev(b2, args.ncomp)
    B.sync()
    fwd = b2.to_host().copy()
    b2.free()
    B.tune_set("ntt_variant", 0)
    if ref is None:
        ref = (inv, fwd)
        ok = bool(np.array_equal(fwd.reshape(-1)[:host.size], host.reshape(-1)))      # round trip returns the input
        print(json.dumps({"variant": name, "round_trip_is_identity": ok}), flush=True)
    else:
        print(json.dumps({"variant": name, "equals_first_variant": bool(np.array_equal(inv, ref[0]) and np.array_equal(fwd, ref[1]))}), flush=True)

times = {name: {"pair": [], "ifft": [], "fft": [], "alt": []} for name, _ in variants}
for rnd in range(args.rounds + 1):
    order = variants if rnd % 2 == 0 else variants[::-1]          # alternate the order inside a round as well
    for name, kv in order:
        apply(kv)
        dom.ifft_in_to_out_dev(buf, args.ncomp)
        e0.record()
        for _ in range(args.reps):
            dom.ifft_in_to_out_dev(buf, args.ncomp)
        e1.record()
        ti = e0.elapsed_ms(e1) / args.reps
        e0.record()
        for _ in range(args.reps):
            dom.fft_out_to_in_dev(buf, args.ncomp)
        e1.record()
        tf = e0.elapsed_ms(e1) / args.reps
        e0.record()
        for _ in range(args.reps):                                 # alternating directions (a witness map's pattern; bench.py's too):
            dom.ifft_in_to_out_dev(buf, args.ncomp)                # both twiddle tables are live at once
            dom.fft_out_to_in_dev(buf, args.ncomp)
        e1.record()
        ta = e0.elapsed_ms(e1) / (2 * args.reps)
        B.tune_set("ntt_variant", 0)
        if rnd:                                                    # round 0 warms up
            times[name]["ifft"].append(ti)
            times[name]["fft"].append(tf)
            times[name]["alt"].append(ta)
            times[name]["pair"].append(ti + tf)
base = variants[0][0]
for name, kv in variants:
    t = times[name]
    row = {"variant": name, "tune": {k: hex(v) for k, v in kv.items()}, "logn": args.logn, "ncomp": args.ncomp, "rounds": args.rounds, "reps": args.reps,
           "ifft_ms_median": round(statistics.median(t["ifft"]), 4), "fft_ms_median": round(statistics.median(t["fft"]), 4),
           "ifft_ms_min": round(min(t["ifft"]), 4), "fft_ms_min": round(min(t["fft"]), 4),
           "alternating_ms_per_transform_median": round(statistics.median(t["alt"]), 4)}
    if name != base:
        d = [(a - b) / b for a, b in zip(t["pair"], times[base]["pair"])]
        row["paired_delta_vs_first_pct_median"] = round(100 * statistics.median(d), 2)
        row["paired_delta_vs_first_pct_min_max"] = [round(100 * min(d), 2), round(100 * max(d), 2)]
    print(json.dumps(row), flush=True)
