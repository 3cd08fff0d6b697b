#!/bin/bash
# level-1 scatter at two blocks per CU (amdgpu_waves_per_eu(8): 64 VGPRs) against one (75 VGPRs, -DCSH_L1_WPE=0): sort-stage times
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for rep in 1 2 3; do
  for v in wpe8 base; do
    unset COSNARKS_HIP_LIB
    [ $v = base ] && export COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_l1base.so
    timeout 600 python tools/gpu_msm_loop.py --reps 6 0:0:20 0:0:22 0:0:24 1:1:20 > $O/y_${v}_$rep.log 2>&1
  done
done
python - <<'PY'
import json
for v in ("wpe8", "base"):
    for rep in (1, 2, 3):
        for ln in open("gpurun_out/y_%s_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print(v, rep, d["curve"], d["group"], d["logn"], "digits", t[0], "scatter", t[2], "accum", t[3], "total", t[5])
PY
