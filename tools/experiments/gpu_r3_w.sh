#!/bin/bash
# merge with one lane per bucket (msm_variant 32, G1 groups) against the four-lane merge; parity of the variant
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
JOBS="0:0:16 0:0:18 0:0:20 0:0:22 0:0:24 1:0:20 1:0:24"
for rep in 1 2; do
  for v in 0 32; do
    CSH_MSM_VARIANT=$v timeout 600 python tools/gpu_msm_loop.py --reps 6 $JOBS > $O/w_v${v}_$rep.log 2>&1
  done
done
python - <<'PY'
import json
for v in (0, 32):
    for rep in (1, 2):
        for ln in open("gpurun_out/w_v%d_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print("variant", v, rep, d["curve"], d["group"], d["logn"], "accum", t[3], "tail", t[4], "total", t[5], "wall", d["wall_ms"])
PY
timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "variants" > $O/pytest_w.log 2>&1
echo "pytest exit $?" >> $O/pytest_w.log; tail -3 $O/pytest_w.log
