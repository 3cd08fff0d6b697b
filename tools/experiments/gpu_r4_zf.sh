#!/bin/bash
# Round 4, run ZF: staged result copies (tune host_d2h = 1) against direct ones: parity, then alternating timing with the transfer counters.
mkdir -p gpurun_out; O=$PWD/gpurun_out
(echo -n "numa_balancing: "; cat /proc/sys/kernel/numa_balancing; for f in enabled defrag; do echo -n "thp $f: "; cat /sys/kernel/mm/transparent_hugepage/$f; done; uname -r; lscpu | grep -E "NUMA node|Socket"; grep -E "thp_fault_alloc|thp_fault_fallback |thp_collapse_alloc |compact_stall|numa_pte_updates|numa_hint_faults " /proc/vmstat) > $O/r04_zf_host.log 2>&1
CSH_HOST_D2H=1 timeout -s KILL 300 python -m pytest tests/test_gpu_trait_path.py tests/test_gpu_groth16.py tests/test_gpu_vec_ntt.py -x -q -m gpu -p no:cacheprovider > $O/r04_zf_pytest_staged.log 2>&1
echo "pytest rc=$?" >> $O/r04_zf_pytest_staged.log
timeout -s KILL 300 python - > $O/r04_zf_staged_ab.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
from cosnarks_amd import bindings as B
keys = ("stat_populate_us", "stat_join_wait_us", "stat_finish_us")
for rnd in range(5):
    for name, kv in (("direct", {"host_d2h": 0}), ("staged", {"host_d2h": 1}), ("staged_nopop", {"host_d2h": 1, "host_populate": 0}), ("staged_4pop", {"host_d2h": 1, "host_populate": 0x104})):
        before = {k: B.tune_get(k) for k in keys}
        with hip.tuned(**kv):
            r = g.bench_synthetic(hip.BN254, 20, 2, with_rep3=False)
        ph = r["trait_path_phases_ms"]
        row = {"round": rnd, "mode": name, "trait_path_ms": round(r["trait_path_ms"], 3), "wm": round(ph["witness_map_host_slices"], 3), "msm": round(ph["msm_groups_host_scalars"], 3),
               "witness_map_ms_zero_filled": round(r["witness_map_ms"], 3), "check": r["trait_path_closed_form_check"]}
        row.update({k[5:]: B.tune_get(k) - before[k] for k in keys})
        print(json.dumps(row), flush=True)
PY
grep -E "thp_fault_alloc|thp_fault_fallback |thp_collapse_alloc |compact_stall|numa_pte_updates|numa_hint_faults " /proc/vmstat >> $O/r04_zf_host.log
tail -3 $O/r04_zf_pytest_staged.log; grep -v amdgpu.ids $O/r04_zf_staged_ab.log | tail -20; cat $O/r04_zf_host.log
