#!/bin/bash
# bit-sliced window reduction (msm_variant 64) against the segment walk + fold tree: stage times and parity
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for rep in 1 2; do
  for v in 0 64; do
    CSH_MSM_VARIANT=$v timeout 300 python tools/gpu_msm_loop.py --reps 5 0:0:18 0:0:20 0:0:22 0:0:24 1:0:20 0:1:20 1:1:20 > $O/ac_v${v}_$rep.log 2>&1
  done
done
python - <<'PY'
import json
for v in (0, 64):
    for rep in (1, 2):
        for ln in open("gpurun_out/ac_v%d_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print("variant", v, rep, d["curve"], d["group"], d["logn"], "tail", t[4], "total", t[5], "wall", d["wall_ms"])
            elif ln.strip() and not ln.startswith(("Hostname", "Librccl")):
                print("variant", v, rep, "LOG", ln.strip()[:200])
PY
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "variants or fuzz or plan" > $O/pytest_ac.log 2>&1
echo "pytest exit $?" >> $O/pytest_ac.log; tail -4 $O/pytest_ac.log
