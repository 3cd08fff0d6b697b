#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/csweep.log
for c in 0 13 14 15 16 0; do
  echo "== CSH_MSM_C=$c" >> gpurun_out/csweep.log
  CSH_MSM_C=$c python tools/gpu_msm_loop.py --reps 8 0:0:17 0:0:18 0:0:19 0:0:20 0:0:21 0:0:22 0:1:20 1:0:20 1:1:20 >> gpurun_out/csweep.log 2>&1
done
grep -E "==|curve" gpurun_out/csweep.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //; s/"Mpts_s_wall".*//' | cut -c1-150
