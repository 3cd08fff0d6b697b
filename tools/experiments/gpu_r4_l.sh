#!/bin/bash
# Round 4, run L: branch-free canonicalisation (canonical_narrow) in every pass tail and share-vector kernel: parity, then timings.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_vec_ntt.py tests/test_gpu_plonk_vectors.py tests/test_gpu_plonk_honk.py "tests/test_gpu_fullsize.py" -m gpu -q -x -p no:cacheprovider -k "not msm" > $O/r04_l_pytest.log 2>&1; tail -3 $O/r04_l_pytest.log
for LOGN in 22 20 24; do timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 10 --reps 10 base=0x0 t10=0x100; done > $O/r04_l_ntt.log 2>&1
timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 2 --rounds 8 --reps 6 base=0x0 t10=0x100 >> $O/r04_l_ntt.log 2>&1
timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 1 --rounds 6 --reps 10 base=0x0 p1=0x1000 p2=0x2000 p3=0x3000 >> $O/r04_l_ntt.log 2>&1
grep "tune\"" $O/r04_l_ntt.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['ncomp'], d['variant'], 'ifft', d['ifft_ms_median'], 'fft', d['fft_ms_median'], 'alt', d['alternating_ms_per_transform_median'], d.get('paired_delta_vs_first_pct_median'))
"
timeout 300 python tools/gpu_vecops_loop.py > $O/r04_l_vecops.log 2>&1; cat $O/r04_l_vecops.log
