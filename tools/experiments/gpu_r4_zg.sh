#!/bin/bash
# Round 4, run ZG: result-copy modes (host_d2h 2 = auto default / 0 direct / 1 staged): parity test, then alternating timing with counters.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 300 python -m pytest tests/test_gpu_trait_path.py -x -q -m gpu -p no:cacheprovider > $O/r04_zg_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r04_zg_pytest.log
timeout -s KILL 300 python - > $O/r04_zg_modes.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
from cosnarks_amd import bindings as B
keys = ("stat_populate_us", "stat_join_wait_us", "stat_finish_us", "stat_d2h_slow", "stat_d2h_staged")
for rnd in range(5):
    for name, kv in (("auto", {}), ("direct", {"host_d2h": 0}), ("staged", {"host_d2h": 1})):
        before = {k: B.tune_get(k) for k in keys}
        with hip.tuned(**kv):
            r = g.bench_synthetic(hip.BN254, 20, 2, with_rep3=False)
        ph = r["trait_path_phases_ms"]
        row = {"round": rnd, "mode": name, "trait_path_ms": round(r["trait_path_ms"], 3), "wm": round(ph["witness_map_host_slices"], 3), "msm": round(ph["msm_groups_host_scalars"], 3),
               "witness_map_ms_zero_filled": round(r["witness_map_ms"], 3), "check": r["trait_path_closed_form_check"]}
        row.update({k[5:]: B.tune_get(k) - before[k] for k in keys})
        print(json.dumps(row), flush=True)
PY
tail -3 $O/r04_zg_pytest.log; grep -v amdgpu.ids $O/r04_zg_modes.log | tail -15
