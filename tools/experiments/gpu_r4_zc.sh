#!/bin/bash
# Round 4, run ZC: the trait path's host side on THIS box: THP mode, page population settings, phases.
mkdir -p gpurun_out; O=$PWD/gpurun_out
(for f in enabled defrag shmem_enabled khugepaged/defrag; do echo -n "thp $f: "; cat /sys/kernel/mm/transparent_hugepage/$f; done; uname -r; nproc; cat /sys/fs/cgroup/cpu.max; free -g | head -2; grep -E "thp|compact_stall" /proc/vmstat | head -20) > $O/r04_zc_host.log 2>&1
timeout -s KILL 420 python - > $O/r04_zc_trait.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
for rnd in range(4):
    for pop in (0x101, 0x1, 0x104, 0x4, 0x0):
        with hip.tuned(host_populate=pop):
            r = g.bench_synthetic(hip.BN254, 20, 2, with_rep3=False)
        ph = r["trait_path_phases_ms"]
        print(json.dumps({"round": rnd, "host_populate": hex(pop), "trait_path_ms": round(r["trait_path_ms"], 3), "wm": round(ph["witness_map_host_slices"], 3),
                          "msm": round(ph["msm_groups_host_scalars"], 3), "finish": round(ph["finish"], 3), "prove_ms": round(r["prove_ms"], 3),
                          "witness_map_ms_zero_filled": round(r["witness_map_ms"], 3)}), flush=True)
PY
grep -E "thp|compact_stall" /proc/vmstat | head -20 >> $O/r04_zc_host.log
grep -v amdgpu.ids $O/r04_zc_trait.log | tail -22; cat $O/r04_zc_host.log
