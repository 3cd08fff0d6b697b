#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/lsweep.log
for L in 32 16 20 24 28 30 34 36 40 48 64 32; do
  echo "== CSH_MSM_L=$L" >> gpurun_out/lsweep.log
  CSH_MSM_L=$L python tools/gpu_msm_loop.py --reps 8 0:0:20 >> gpurun_out/lsweep.log 2>&1
done
for L in 128 96 112 136 144 160 192 256 128; do
  echo "== CSH_MSM_L=$L" >> gpurun_out/lsweep.log
  CSH_MSM_L=$L python tools/gpu_msm_loop.py --reps 5 0:0:22 >> gpurun_out/lsweep.log 2>&1
done
grep -E "==|curve" gpurun_out/lsweep.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/' | cut -c1-200
