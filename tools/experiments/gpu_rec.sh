#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -q -x -p no:cacheprovider -k "closed_form or window_sizes" > gpurun_out/rec_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/rec_tests.log | tail -3
: > gpurun_out/rec_stages.log
for v in 0 8 0 8; do
  echo "== CSH_MSM_VARIANT=$v" >> gpurun_out/rec_stages.log
  CSH_MSM_VARIANT=$v python tools/gpu_msm_loop.py --reps 6 0:0:24 1:0:24 >> gpurun_out/rec_stages.log 2>&1
done
cat gpurun_out/rec_stages.log
