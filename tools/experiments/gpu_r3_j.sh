#!/bin/bash
# MSM suites on the final default (separate merge launch; fused merge as variant 16 / 17)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_msm.log 2>&1
echo "pytest exit $?" >> $O/pytest_msm.log; grep -E "passed|failed" $O/pytest_msm.log | tail -2
timeout 300 python tools/gpu_msm_loop.py --reps 10 0:0:20 0:0:24 > $O/msm_final_check.log 2>&1; cat $O/msm_final_check.log
