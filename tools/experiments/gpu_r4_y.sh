#!/bin/bash
# Round 4, run Y: robustness of the D2H page population (retrying, 2 MiB-granular, huge-page hint): six alternating trials per setting.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 500 python - > $O/r04_y_populate.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
for rnd in range(6):
    for pop in (0x102, 0x104, 0x2, 0x4, 0x101):
        with hip.tuned(host_populate=pop):
            r = g.bench_synthetic(hip.BN254, 20, 2, with_rep3=False)
            print(json.dumps({"round": rnd, "host_populate": hex(pop), "trait_path_ms": round(r["trait_path_ms"], 3), "witness_map_host_slices": round(r["trait_path_phases_ms"]["witness_map_host_slices"], 3),
                              "msm": round(r["trait_path_phases_ms"]["msm_groups_host_scalars"], 3), "witness_map_ms_zero_filled_vector": round(r["witness_map_ms"], 3)}), flush=True)
PY
grep -v amdgpu.ids $O/r04_y_populate.log | python -c "
import sys, json, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); d[r['host_populate']].append((r['witness_map_host_slices'], r['trait_path_ms']))
for k, v in d.items(): print(k, 'witness map', sorted(x[0] for x in v), 'prove', sorted(x[1] for x in v))
"
