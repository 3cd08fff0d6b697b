#!/bin/bash
# Round 4, run X: (1) c = 15 vs 16 at 2^19..2^21, interleaved; (2) page population of large D2H results with and without
# a transparent-huge-page hint, trait-path prove at 2^20.
mkdir -p gpurun_out; O=$PWD/gpurun_out
for J in 0:0:20 0:0:19 0:0:21; do timeout -s KILL 120 python tools/msm_ab.py --job $J --rounds 12 --reps 10 auto=msm_c=0 c15=msm_c=15 c16=msm_c=16; done > $O/r04_x_msm_c_ab.log 2>&1
grep "ms_median" $O/r04_x_msm_c_ab.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['job'], d['variant'], d['params_c_W_L_S'], d['ms_median'], d['Mpts_s_median'], d.get('paired_delta_vs_first_pct_median'), d.get('paired_delta_vs_first_pct_min_max'))
"
timeout -s KILL 300 python - > $O/r04_x_populate.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
for rnd in range(2):
    for pop in (4, 0x104, 2, 0x102, 0x108):
        with hip.tuned(host_populate=pop):
            r = g.bench_synthetic(hip.BN254, 20, 3, with_rep3=False)
            print(json.dumps({"host_populate": hex(pop), "trait_path_ms": round(r["trait_path_ms"], 3), "phases": {k: round(v, 3) for k, v in r["trait_path_phases_ms"].items()}}), flush=True)
PY
cat $O/r04_x_populate.log | grep -v amdgpu.ids
