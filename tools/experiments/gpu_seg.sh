#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py -m gpu -q -x -p no:cacheprovider -k "not config5" > gpurun_out/seg_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/seg_tests.log | tail -3
: > gpurun_out/seg_stages.log
for cfg in "0 0" "8 0" "0 4" "8 4" "0 0" "8 0"; do
  set -- $cfg
  echo "== CSH_MSM_SEG_BUCKETS=$1 CSH_MSM_VARIANT=$2" >> gpurun_out/seg_stages.log
  CSH_MSM_SEG_BUCKETS=$1 CSH_MSM_VARIANT=$2 python tools/gpu_msm_loop.py --reps 10 0:0:20 0:1:20 1:0:20 1:1:20 0:0:18 0:0:22 >> gpurun_out/seg_stages.log 2>&1
done
grep -E "==|curve" gpurun_out/seg_stages.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/' | cut -c1-170
