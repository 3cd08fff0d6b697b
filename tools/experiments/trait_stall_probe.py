#!/usr/bin/env python3
"""Which setting makes the host-facing witness map of the trait path take 20+ ms instead of 2.4 on some boxes / in some process states?
Runs the plain trait-path prove of the synthetic 2^20 circuit repeatedly under a sequence of tune settings and prints the phases and
the host-copy counters after every step."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cosnarks_amd as hip
from cosnarks_amd import bindings as B, groth16 as g

B._check(hip.lib().csh_init(0))
B.tune_set("host_timing", int(os.environ.get("PROBE_HOST_TIMING", "0")))
B.tune_set("host_h2d", int(os.environ.get("PROBE_HOST_H2D", "2")))
B.tune_set("host_d2h", int(os.environ.get("PROBE_HOST_D2H", "2")))
B.tune_set("host_copier_pool", int(os.environ.get("PROBE_POOL", "1")))
B.tune_set("host_populate", int(os.environ.get("PROBE_POPULATE", "257")))
KEYS = ("stat_stage_all_switches", "stat_d2h_slow", "stat_h2d_slow", "stat_h2d_staged", "stat_wm_h2d_us", "stat_wm_dev_us", "stat_wm_d2h_us", "stat_populate_us", "stat_join_wait_us", "stat_finish_us", "stat_d2h_slow", "stat_d2h_staged", "stat_uploads_shared", "stat_arena_grows", "stat_lanes")
if len(sys.argv) > 1 and sys.argv[1] == "short":
    seq = [dict(msm_share_uploads=1)] * int(os.environ.get("PROBE_STEPS", "4"))
else:
  seq = [dict(msm_share_uploads=0)] * 3 + [dict(msm_share_uploads=1)] * 3 + [dict(msm_share_uploads=0)] * 2 + \
        [dict(msm_share_uploads=0, host_d2h=1)] * 2 + [dict(msm_share_uploads=0, host_d2h=0)] * 2 + [dict(msm_share_uploads=0, host_d2h=2, host_populate=0)] * 2 + \
        [dict(msm_share_uploads=1, host_d2h=2, host_populate=0x101)] * 2
defaults = {k: B.tune_get(k) for k in ("msm_share_uploads", "host_d2h", "host_populate", "host_h2d", "host_copier_pool")}
for i, kv in enumerate(seq):
    for k, v in {**defaults, **kv}.items():
        B.tune_set(k, v)
    before = {k: B.tune_get(k) for k in KEYS}
    r = g.bench_synthetic(hip.BN254, 20, 7, with_rep3=False)
    d = {k[5:]: B.tune_get(k) - before[k] for k in KEYS}
    print(json.dumps({"step": i, "tune": kv, "prove_ms": round(r["prove_ms"], 2), "witness_map_alone_ms": round(r["witness_map_ms"], 2), "trait_ms": round(r["trait_path_ms"], 2),
                      "trait_min": round(r["trait_path_ms_min"], 2), "trait_phases": {k: round(v, 2) for k, v in r["trait_path_phases_ms"].items()}, "counters": d}), flush=True)
