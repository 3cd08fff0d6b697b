#!/bin/bash
# NTT ablations (wrong results on purpose): no twiddle gather / no multiplication / neither -- where does the pass spend its time?
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for v in sliced abl_tw abl_mul abl_both; do echo "== $v"; COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_$v.so NTT_LOGN=20,22,24 timeout 300 python tools/gpu_probe_ntt.py; done > $O/ntt_ablate.log 2>&1; grep -E "==|\"ntt\"" $O/ntt_ablate.log
