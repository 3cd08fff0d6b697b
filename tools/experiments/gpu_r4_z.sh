#!/bin/bash
# Round 4, run Z: coalesced host-scalar MSMs (csh_msm): parity tests, then the trait-path prove with msm_coalesce = 0 / 1, alternating.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 420 python -m pytest tests/test_gpu_msm_coalesce.py tests/test_gpu_trait_path.py tests/test_gpu_groth16.py -x -q -m gpu -p no:cacheprovider > $O/r04_z_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r04_z_pytest.log
timeout -s KILL 400 python - > $O/r04_z_coalesce.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
for logn in (20, 18, 16):
    for rnd in range(5):
        for co in (0, 1):
            with hip.tuned(msm_coalesce=co):
                r = g.bench_synthetic(hip.BN254, logn, 2, with_rep3=False)
                print(json.dumps({"log_n": logn, "round": rnd, "msm_coalesce": co, "trait_path_ms": round(r["trait_path_ms"], 3),
                                  "msm": round(r["trait_path_phases_ms"]["msm_groups_host_scalars"], 3), "device_resident_ms": round(r["prove_ms"], 3) if "prove_ms" in r else None,
                                  "check": r.get("trait_path_closed_form_check")}), flush=True)
PY
grep -v amdgpu.ids $O/r04_z_coalesce.log | python -c "
import sys, json, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); d[(r['log_n'], r['msm_coalesce'])].append((r['trait_path_ms'], r['msm']))
for k, v in sorted(d.items()): print(k, 'prove', sorted(x[0] for x in v), 'msm', sorted(x[1] for x in v))
" | tee $O/r04_z_summary.txt
tail -5 $O/r04_z_pytest.log
