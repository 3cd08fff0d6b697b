#!/bin/bash
# staged tables: packed 32-byte entries (re-sliced per butterfly) vs pre-sliced 9-limb entries, A/B on one box
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_vec_ntt.py -m gpu -q --timeout 900 -p no:cacheprovider -k "ntt or fft" > $O/pytest_ntt.log 2>&1
echo "pytest exit $?" >> $O/pytest_ntt.log; grep -E "passed|failed" $O/pytest_ntt.log | tail -2
for round in 1 2; do for v in packed sliced; do echo "== $v"; COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_$v.so NTT_LOGN=18,20,22,24 timeout 300 python tools/gpu_probe_ntt.py; done; done > $O/ntt_sliced2.log 2>&1; grep -E "==|\"ntt\"" $O/ntt_sliced2.log | grep -E "==|ncomp\": 1"
