#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py -m gpu -q -x -p no:cacheprovider -k "not config5" > gpurun_out/exp_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/exp_tests.log | tail -3
python tools/gpu_msm_loop.py --reps 10 0:0:20 0:0:18 0:1:20 1:1:20 0:0:20 > gpurun_out/exp_stages.log 2>&1
cat gpurun_out/exp_stages.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //' | cut -c1-170
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; done
