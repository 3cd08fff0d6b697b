#!/bin/bash
# round 5, call t: staged transfers as the default -- 8 processes of the stall probe with the defaults, 4 with direct copies for contrast; parity
O=gpurun_out/r05_t; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do echo "== process $i defaults (host_d2h 1, host_h2d 1)" >> $O/probe_default.log; PROBE_HOST_D2H=1 PROBE_HOST_H2D=1 PROBE_STEPS=5 python tools/experiments/trait_stall_probe.py short 2>&1 | cut -c1-600 >> $O/probe_default.log; done
for i in 1 2 3 4; do echo "== process $i direct (host_d2h 0, host_h2d 0)" >> $O/probe_default.log; PROBE_HOST_D2H=0 PROBE_HOST_H2D=0 PROBE_STEPS=5 python tools/experiments/trait_stall_probe.py short 2>&1 | cut -c1-600 >> $O/probe_default.log; done
python -m pytest tests/test_gpu_trait_path.py tests/test_gpu_vec_ntt.py tests/test_gpu_groth16.py tests/test_gpu_plonk_honk.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -3 $O/pytest.log
python - <<'PY'
import re
for l in open("gpurun_out/r05_t/probe_default.log"):
    if l.startswith("=="): print(l.strip()); continue
    m=re.search(r'trait_ms": ([0-9.]+), "trait_min": ([0-9.]+).*witness_map_host_slices": ([0-9.]+), "msm_groups_host_scalars": ([0-9.]+)', l)
    print("  ", m.groups() if m else l[:160])
PY
