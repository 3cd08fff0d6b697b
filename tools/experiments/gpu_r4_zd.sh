#!/bin/bash
# Round 4, run ZD: why the trait path slowed down after / in the no-table modes of run ZB: arena and lane counters across mode switches.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 420 python - > $O/r04_zd_trait_modes.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
from cosnarks_amd import bindings as B
seq = [("tables", {})] * 3 + [("no_tables", {"msm_no_table": 1})] * 4 + [("tables", {})] * 4 + [("no_tables", {"msm_no_table": 1})] * 2
for i, (name, kv) in enumerate(seq):
    a0, l0 = B.tune_get("stat_arena_grows"), B.tune_get("stat_lanes")
    with hip.tuned(**kv):
        r = g.bench_synthetic(hip.BN254, 20, 2, with_rep3=False)
    ph = r["trait_path_phases_ms"]
    print(json.dumps({"i": i, "mode": name, "trait_path_ms": round(r["trait_path_ms"], 3), "wm": round(ph["witness_map_host_slices"], 3), "msm": round(ph["msm_groups_host_scalars"], 3),
                      "finish": round(ph["finish"], 3), "prove_ms": round(r["prove_ms"], 3), "arena_grows": B.tune_get("stat_arena_grows") - a0, "lanes_new": B.tune_get("stat_lanes") - l0,
                      "lanes": B.tune_get("stat_lanes")}), flush=True)
PY
grep -v amdgpu.ids $O/r04_zd_trait_modes.log | tail -20
