#!/bin/bash
# G2 accumulate forms on the current tree: whole point per lane (default), the same without the register prefetch of the next point (cache-line
# touch only, -DCSH_ACC_TOUCH, BLS12-381 G2), two lanes per point (msm_variant 2)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
JOBS="0:1:20 1:1:20 1:1:22"
for rep in 1 2; do
  for v in base touch pair; do
    unset COSNARKS_HIP_LIB CSH_MSM_VARIANT
    [ $v = touch ] && export COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_touch.so
    [ $v = pair ] && export CSH_MSM_VARIANT=2
    timeout 600 python tools/gpu_msm_loop.py --reps 5 $JOBS > $O/v_${v}_$rep.log 2>&1
  done
done
python - <<'PY'
import json
for v in ("base", "touch", "pair"):
    for rep in (1, 2):
        for ln in open("gpurun_out/v_%s_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print(v, rep, d["curve"], d["group"], d["logn"], "accum", t[3], "reduce", t[4], "total", t[5], "wall", d["wall_ms"])
PY
