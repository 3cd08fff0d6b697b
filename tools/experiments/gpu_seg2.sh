#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/seg3.log
for v in 0 1 0 1; do
  echo "== CSH_MSM_VARIANT=$v" >> gpurun_out/seg3.log
  CSH_MSM_VARIANT=$v python tools/gpu_msm_loop.py --reps 10 0:0:20 1:0:20 0:0:18 0:0:22 0:0:24 >> gpurun_out/seg3.log 2>&1
done
grep -E "==|curve" gpurun_out/seg3.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //; s/"Mpts_s_wall".*//' | cut -c1-150
