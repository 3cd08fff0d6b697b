#!/bin/bash
# window reduction: buckets per segment (tune msm_seg_buckets; 0 = one round of waves at one wave per SIMD). The chain probe (r03_o) says one
# wave of dependent multiply-adds per SIMD issues at a third of the pipe's rate, two at two thirds: half-length segments = two waves per SIMD.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for rep in 1 2; do
  for sb in 0 12 9 6 4; do
    CSH_MSM_SEG_BUCKETS=$sb timeout 600 python tools/gpu_msm_loop.py --reps 6 0:0:18 0:0:20 0:0:22 0:0:24 1:0:20 0:1:20 1:1:20 > $O/z_sb${sb}_$rep.log 2>&1
  done
done
python - <<'PY'
import json
for sb in (0, 12, 9, 6, 4):
    for rep in (1, 2):
        for ln in open("gpurun_out/z_sb%d_%d.log" % (sb, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print("seg_buckets", sb, rep, d["curve"], d["group"], d["logn"], d["params_c_W_L_S"], "tail", t[4], "total", t[5])
PY
