#!/bin/bash
# Round 4, run W: unit-twiddle rounds (global stages 0 / 1 without their trivial multiplications): parity on every size and plan, then A/B.
mkdir -p gpurun_out; R=$PWD; O=$R/gpurun_out
timeout -s KILL 900 python -m pytest tests/test_gpu_vec_ntt.py tests/test_gpu_plonk_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py tests/test_gpu_trait_path.py -m gpu -q -x -p no:cacheprovider --timeout 300 -k "not msm" > $O/r04_w_pytest.log 2>&1; tail -3 $O/r04_w_pytest.log
for LOGN in 22 20 24 16 18; do timeout -s KILL 200 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 10 --reps 10 unit=0x0 general=0x100000; done > $O/r04_w_ntt_unit_ab.log 2>&1
timeout -s KILL 200 python tools/ntt_ab.py --logn 22 --ncomp 2 --rounds 6 --reps 6 unit=0x0 general=0x100000 >> $O/r04_w_ntt_unit_ab.log 2>&1
grep "false" $O/r04_w_ntt_unit_ab.log
grep "tune\"" $O/r04_w_ntt_unit_ab.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['ncomp'], d['variant'], 'ifft', d['ifft_ms_median'], 'fft', d['fft_ms_median'], 'alt', d['alternating_ms_per_transform_median'], d.get('paired_delta_vs_first_pct_median'), d.get('paired_delta_vs_first_pct_min_max'))
"
