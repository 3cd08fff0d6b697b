#!/bin/bash
# straight-line digits kernel: stage times (digits + histogram is stage 0) and the MSM / split / prover parity suites
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 600 python tools/gpu_msm_loop.py --reps 6 0:0:16 0:0:20 0:0:22 0:0:24 1:0:20 1:1:20 2:0:20 > $O/x_stages.log 2>&1; cat $O/x_stages.log | cut -c1-230
LOOP_PRECOMPUTE=16:2 timeout 600 python tools/gpu_msm_loop.py --reps 6 0:0:20 > $O/x_stages_tab.log 2>&1; cat $O/x_stages_tab.log | cut -c1-230
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py tests/test_gpu_plonk_honk.py -m gpu -q --timeout 900 -p no:cacheprovider -x > $O/pytest_x.log 2>&1
echo "pytest exit $?" >> $O/pytest_x.log; tail -3 $O/pytest_x.log
