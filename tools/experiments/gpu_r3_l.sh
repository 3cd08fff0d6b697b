#!/bin/bash
# radix-4 DIF with two folds per round; default = radix-4 for forward transforms >= 2^20
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_vec_ntt.py tests/test_gpu_plonk_honk.py tests/test_gpu_groth16.py -m gpu -q --timeout 900 -p no:cacheprovider -k "ntt or fft or golden or libsnark or closed_form" > $O/pytest_ntt.log 2>&1
echo "pytest exit $?" >> $O/pytest_ntt.log; grep -E "passed|failed" $O/pytest_ntt.log | tail -2
for v in 0 1 2; do echo "== CSH_NTT_VARIANT=$v (0 default mix, 1 radix-4 everywhere, 2 radix-2 everywhere)"; CSH_NTT_VARIANT=$v NTT_LOGN=18,20,22,24 timeout 300 python tools/gpu_probe_ntt.py; done > $O/ntt_mix.log 2>&1; grep -E "==|\"ntt\"" $O/ntt_mix.log
