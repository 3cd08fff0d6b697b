#!/bin/bash
# Round 4, run ZB: what the fixed-base tables and the two-stream bucket stages are worth inside a device-resident prove (2^20, 2^18).
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 420 python - > $O/r04_zb_prove_modes.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
modes = {"tables": {}, "no_tables_overlap": {"msm_no_table": 1}, "no_tables_serial": {"msm_no_table": 1, "msm_multi_overlap": 0}}
for logn in (20, 18):
    for rnd in range(4):
        for name, kv in modes.items():
            with hip.tuned(**kv):
                r = g.bench_synthetic(hip.BN254, logn, 2, with_rep3=False)
            print(json.dumps({"log_n": logn, "round": rnd, "mode": name, "prove_ms": round(r["prove_ms"], 3), "msm_groups": round(r["prove_phases_ms"]["msm_groups"], 3),
                              "trait_path_ms": round(r["trait_path_ms"], 3), "check": r["closed_form_check"]}), flush=True)
PY
grep -v amdgpu.ids $O/r04_zb_prove_modes.log | python -c "
import sys, json, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); d[(r['log_n'], r['mode'])].append((r['prove_ms'], r['msm_groups'], r['trait_path_ms']))
    else: print(l.rstrip()[:300])
for k, v in sorted(d.items()): print(k, 'prove', sorted(x[0] for x in v), 'msm', sorted(x[1] for x in v), 'trait', sorted(x[2] for x in v))
"
