#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "not config5" > gpurun_out/trim_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/trim_tests.log | tail -3
python tools/gpu_msm_loop.py --reps 8 0:0:20 0:1:20 1:0:20 1:1:20 2:0:20 0:0:24 0:0:22 > gpurun_out/trim_stages.log 2>&1
cat gpurun_out/trim_stages.log
