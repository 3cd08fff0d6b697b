#!/bin/bash
# Round 3, third GPU call: radix-4 NTT pass vs radix-2 (parity on every size, timing), BLS12-381 G2 accumulate forms, new G2 parity tests
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_vec_ntt.py tests/test_gpu_plonk_honk.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=6 -k "ntt or fft or g2_full_range" > $O/pytest_ntt.log 2>&1
echo "pytest exit $?" >> $O/pytest_ntt.log; grep -E "passed|failed" $O/pytest_ntt.log | tail -2
for v in 0 1; do echo "== CSH_NTT_VARIANT=$v (0 radix-4, 1 radix-2)"; CSH_NTT_VARIANT=$v NTT_LOGN=16,18,20,22,24 timeout 300 python tools/gpu_probe_ntt.py; done > $O/ntt_r4.log 2>&1; grep -E "==|\"ntt\"" $O/ntt_r4.log
echo "== BLS12-381 G2 accumulate: whole point per lane (0) vs lane pair (2)" > $O/g2_acc_forms.log
for v in 0 2; do echo "-- CSH_MSM_VARIANT=$v" >> $O/g2_acc_forms.log; CSH_MSM_VARIANT=$v timeout 300 python tools/gpu_msm_loop.py --reps 10 1:1:20 0:1:20 1:1:22 >> $O/g2_acc_forms.log 2>&1; done; cat $O/g2_acc_forms.log
