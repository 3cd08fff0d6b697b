#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -q -x -p no:cacheprovider -k "fixed_base or small_matches or edge_cases or window_sizes" > gpurun_out/grp_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/grp_tests.log | tail -5
: > gpurun_out/grp.log
for pc in "" 15:2 16:2 15:4 16:4 15:3 ""; do
  echo "== LOOP_PRECOMPUTE=$pc" >> gpurun_out/grp.log
  LOOP_PRECOMPUTE=$pc python tools/gpu_msm_loop.py --reps 10 0:0:20 0:1:20 0:0:18 0:0:22 >> gpurun_out/grp.log 2>&1
done
grep -E "==|curve|rror" gpurun_out/grp.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //' | cut -c1-170
