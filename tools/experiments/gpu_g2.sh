#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_groth16.py -m gpu -q -x -p no:cacheprovider -k "not config5 and not closed_form_full" > gpurun_out/g2_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/g2_tests.log | tail -3
python tools/gpu_msm_loop.py --reps 10 0:1:20 0:1:22 0:1:18 0:1:20 > gpurun_out/g2_stages.log 2>&1
cat gpurun_out/g2_stages.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //' | cut -c1-170
timeout 600 python -c "
import cosnarks_amd.groth16 as g
for _ in range(3): print(g.bench_synthetic(0, 20, 4, False))" 2>&1 | tail -3
