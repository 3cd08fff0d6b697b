#!/bin/bash
# accumulate index prefetch: two entries ahead (default) against the former one-ahead form (-DCSH_ACC_PREFETCH1), all groups, one box
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
JOBS="0:0:20 0:0:22 0:0:24 1:0:20 0:1:20 1:1:20"
for rep in 1 2 3; do
  for v in base pf1; do
    unset COSNARKS_HIP_LIB
    [ $v = pf1 ] && export COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_pf1.so
    timeout 600 python tools/gpu_msm_loop.py --reps 6 $JOBS > $O/u_${v}_$rep.log 2>&1
  done
done
python - <<'PY'
import json
for v in ("base", "pf1"):
    for rep in (1, 2, 3):
        for ln in open("gpurun_out/u_%s_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print(v, rep, d["curve"], d["group"], d["logn"], "accum", t[3], "reduce", t[4], "total", t[5], "wall", d["wall_ms"])
PY
