#!/bin/bash
# staged twiddle tables (one contiguous run per stage): parity on every NTT size / h pipelines, timing
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_vec_ntt.py tests/test_gpu_plonk_honk.py tests/test_gpu_groth16.py -m gpu -q --timeout 900 -p no:cacheprovider -k "ntt or fft or golden or libsnark or closed_form" > $O/pytest_ntt.log 2>&1
echo "pytest exit $?" >> $O/pytest_ntt.log; grep -E "passed|failed" $O/pytest_ntt.log | tail -2
for v in 0 1; do echo "== CSH_NTT_VARIANT=$v (0 radix-2, 1 radix-4), staged twiddle tables"; CSH_NTT_VARIANT=$v NTT_LOGN=16,20,22,24 timeout 300 python tools/gpu_probe_ntt.py; done > $O/ntt_staged.log 2>&1; grep -E "==|\"ntt\"" $O/ntt_staged.log
