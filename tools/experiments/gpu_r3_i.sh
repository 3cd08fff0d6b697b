#!/bin/bash
# fused merge (the window reduction folds a bucket's partial slots itself) vs the separate merge launch (msm_variant 16); MSM parity suites
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_msm_split.py -m gpu -q --timeout 900 -p no:cacheprovider -k "not ntt" > $O/pytest_msm.log 2>&1
echo "pytest exit $?" >> $O/pytest_msm.log; grep -E "passed|failed" $O/pytest_msm.log | tail -2
for round in 1 2; do for v in 0 16; do echo "== CSH_MSM_VARIANT=$v (0 fused merge, 16 separate merge launch)"; CSH_MSM_VARIANT=$v timeout 300 python tools/gpu_msm_loop.py --reps 10 0:0:18 0:0:20 0:0:22 0:0:24 0:1:20 1:0:20 1:1:20; done; done > $O/fused_merge.log 2>&1
python - <<'PY'
import json
cur=None
for l in open("gpurun_out/fused_merge.log"):
    if l.startswith("=="): cur=l.split()[1]; continue
    if l.startswith("{"):
        d=json.loads(l); t=d["ms_digits_scan_scatter_accum_reduce_total"]; print(cur, d["curve"], d["group"], d["logn"], "accum", t[3], "tail", t[4], "total", t[5], "wall", d["wall_ms"])
PY
