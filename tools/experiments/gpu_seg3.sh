#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "not config5" > gpurun_out/seg4_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/seg4_tests.log | tail -3
: > gpurun_out/seg4.log
for v in 0 1 4 0 1; do
  echo "== CSH_MSM_VARIANT=$v" >> gpurun_out/seg4.log
  CSH_MSM_VARIANT=$v python tools/gpu_msm_loop.py --reps 10 0:1:20 1:1:20 0:1:22 0:0:20 >> gpurun_out/seg4.log 2>&1
done
grep -E "==|curve" gpurun_out/seg4.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //; s/"Mpts_s_wall".*//' | cut -c1-150
