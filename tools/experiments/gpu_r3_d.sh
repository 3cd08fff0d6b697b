#!/bin/bash
# persistent radix-4 NTT (next tile prefetched into registers) vs one workgroup per tile vs radix-2
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_vec_ntt.py -m gpu -q --timeout 900 -p no:cacheprovider -k "ntt or fft" > $O/pytest_ntt.log 2>&1
echo "pytest exit $?" >> $O/pytest_ntt.log; grep -E "passed|failed" $O/pytest_ntt.log | tail -2
for v in 0 2 1; do echo "== CSH_NTT_VARIANT=$v (0 radix-4 persistent, 2 radix-4 one workgroup per tile, 1 radix-2)"; CSH_NTT_VARIANT=$v NTT_LOGN=18,20,22,24 timeout 300 python tools/gpu_probe_ntt.py; done > $O/ntt_r4p.log 2>&1; grep -E "==|\"ntt\"" $O/ntt_r4p.log
