#!/bin/bash
# Round 4, run ZN: occupancy-aware lane length in the merged (fixed-base table) and shared (multi) plans: parity suites + prove / trait timings at 2^16, 2^18, 2^20.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 800 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py tests/test_gpu_plonk_honk.py tests/test_gpu_plonk_vectors.py tests/test_gpu_trait_path.py tests/test_gpu_msm_split.py -q -m gpu -p no:cacheprovider --maxfail 5 > $O/r04_zn_pytest.log 2>&1
echo "pytest exit $?" >> $O/r04_zn_pytest.log
timeout -s KILL 300 python - > $O/r04_zn_prove_sizes.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
for logn in (14, 16, 17, 18, 20):
    for rnd in range(3):
        r = g.bench_synthetic(hip.BN254, logn, 2, with_rep3=False)
        print(json.dumps({"log_n": logn, "round": rnd, "prove_ms": round(r["prove_ms"], 3), "msm_groups": round(r["prove_phases_ms"]["msm_groups"], 3), "trait_path_ms": round(r["trait_path_ms"], 3),
                          "trait_msm": round(r["trait_path_phases_ms"]["msm_groups_host_scalars"], 3), "check": r["closed_form_check"] and r["trait_path_closed_form_check"]}), flush=True)
PY
grep -E "passed|failed|exit" $O/r04_zn_pytest.log | tail -3; grep -v amdgpu.ids $O/r04_zn_prove_sizes.log | tail -15
