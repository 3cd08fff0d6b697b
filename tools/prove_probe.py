#!/usr/bin/env python3
"""Synthetic 2^k Groth16 prove (device-resident mirror + trait path) under different fixed-base table settings of the key, one process:
COG16_TABLES unset = the library's policy (round 6: one bucket set, 17-bit windows), 4 = the grouped tables of rounds 2-5 (c = 16, 4 rows),
0 = plain handles. Medians of `iters` proves, phases beside them.
    python tools/prove_probe.py [--log-n 20] [--iters 11] [--modes policy 4 0]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g16

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--iters", type=int, default=11)
ap.add_argument("--modes", nargs="*", default=["policy", "4", "0"])
ap.add_argument("--rep3", action="store_true")
args = ap.parse_args()
for rnd in range(2):
    for m in args.modes:
        if m == "policy":
            os.environ.pop("COG16_TABLES", None)
        else:
            os.environ["COG16_TABLES"] = m
        r = g16.bench_synthetic(hip.BN254, args.log_n, args.iters, with_rep3=args.rep3)
        keep = {k: r[k] for k in ("prove_ms", "prove_ms_min", "prove_phases_ms", "trait_path_ms", "trait_path_phases_ms", "key_setup_ms", "closed_form_check",
                                  "trait_path_closed_form_check")}
        if args.rep3:
            keep["rep3_three_parties_prove_ms"] = r["rep3_three_parties_prove_ms"]
            keep["rep3_trait_path"] = {k: {"one_party_alone_ms": v["one_party_alone_ms"], "phases": v["one_party_alone_phases_ms"], "three_parties_one_gpu_ms": v["three_parties_one_gpu_ms"],
                                           "proofs_equal_plain": v["proofs_equal_plain"]} for k, v in r["rep3_trait_path"].items()}
        print(json.dumps({"round": rnd, "tables": m, "log_n": args.log_n, **keep}), flush=True)
