#!/bin/bash
# A/B of the lane-pair G2 kernels: parity first, then stage timings with variant bits (2: serial accumulate, 4: serial reduce)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py -m gpu -q -x -p no:cacheprovider -k "not config5" > gpurun_out/pair_tests.log 2>&1
tail -3 gpurun_out/pair_tests.log
for v in 0 6 2 4; do
  echo "== CSH_MSM_VARIANT=$v" >> gpurun_out/pair_stages.log
  CSH_MSM_VARIANT=$v python tools/gpu_msm_loop.py 0:1:20 1:1:20 0:1:22 >> gpurun_out/pair_stages.log 2>&1
done
cat gpurun_out/pair_stages.log
