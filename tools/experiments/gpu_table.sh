#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/table.log
for pc in "" auto 15 16 13; do
  echo "== LOOP_PRECOMPUTE=$pc" >> gpurun_out/table.log
  LOOP_PRECOMPUTE=$pc python tools/gpu_msm_loop.py --reps 10 0:0:20 0:0:18 0:1:20 0:0:22 >> gpurun_out/table.log 2>&1
done
grep -E "==|curve|rror" gpurun_out/table.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //' | cut -c1-170
