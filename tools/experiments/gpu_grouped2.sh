#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/grp2.log
for pc in "" 16:2 16:4 ""; do
  echo "== LOOP_PRECOMPUTE=$pc" >> gpurun_out/grp2.log
  LOOP_PRECOMPUTE=$pc python tools/gpu_msm_loop.py --reps 8 1:0:20 1:1:20 2:0:20 0:0:19 0:0:21 >> gpurun_out/grp2.log 2>&1
done
grep -E "==|curve|rror" gpurun_out/grp2.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //' | cut -c1-170
