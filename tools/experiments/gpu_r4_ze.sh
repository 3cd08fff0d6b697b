#!/bin/bash
# Round 4, run ZE: where a slow host-facing witness map spends its time (population worker, the caller's wait for it, the final stream wait).
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 420 python - > $O/r04_ze_populate_trace.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
from cosnarks_amd import bindings as B
keys = ("stat_populate_us", "stat_join_wait_us", "stat_finish_us", "stat_arena_grows")
seq = [0x101] * 4 + [0x0, 0x101, 0x101, 0x0, 0x101, 0x1, 0x101, 0x101, 0x104, 0x101, 0x101]
for i, pop in enumerate(seq):
    before = {k: B.tune_get(k) for k in keys}
    with hip.tuned(host_populate=pop):
        r = g.bench_synthetic(hip.BN254, 20, 2, with_rep3=False)
    ph = r["trait_path_phases_ms"]
    row = {"i": i, "host_populate": hex(pop), "trait_path_ms": round(r["trait_path_ms"], 3), "wm": round(ph["witness_map_host_slices"], 3), "msm": round(ph["msm_groups_host_scalars"], 3),
           "witness_map_ms_zero_filled": round(r["witness_map_ms"], 3)}
    row.update({k[5:]: B.tune_get(k) - before[k] for k in keys})
    print(json.dumps(row), flush=True)
PY
grep -v amdgpu.ids $O/r04_ze_populate_trace.log | tail -20
