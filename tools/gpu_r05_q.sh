#!/bin/bash
# round 5, call q: staged uploads -- parity, then the stall probe per upload mode (direct / staged / auto), several processes each
O=gpurun_out/r05_q; mkdir -p $O
python -m pytest tests/test_gpu_trait_path.py tests/test_gpu_msm.py tests/test_gpu_vec_ntt.py tests/test_gpu_groth16.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
for i in 1 2 3; do for m in 0 1 2; do echo "== process $i host_h2d $m" >> $O/probe_h2d.log; PROBE_HOST_H2D=$m python tools/experiments/trait_stall_probe.py short 2>&1 | cut -c1-520 >> $O/probe_h2d.log; done; done
tail -3 $O/pytest.log
python - <<'PY'
import re
for l in open("gpurun_out/r05_q/probe_h2d.log"):
    if l.startswith("=="): print(l.strip()); continue
    m=re.search(r'trait_ms": ([0-9.]+).*witness_map_host_slices": ([0-9.]+), "msm_groups_host_scalars": ([0-9.]+).*h2d_slow": (\d+), "h2d_staged": (\d+)', l)
    print("  ", m.groups() if m else l[:120])
PY
