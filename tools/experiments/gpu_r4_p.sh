#!/bin/bash
# Round 4, run P: the tile-size rule as the default: full NTT / vector / Groth16 parity, then the size sweep against the forced 2^11 tiles.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 2400 python -m pytest tests/test_gpu_vec_ntt.py tests/test_gpu_plonk_vectors.py tests/test_gpu_plonk_honk.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py tests/test_gpu_trait_path.py -m gpu -q -x -p no:cacheprovider -k "not msm" > $O/r04_p_pytest.log 2>&1; tail -3 $O/r04_p_pytest.log
for LOGN in 10 12 13 14 15 16 17 18 19 20 22 24; do
  timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 6 --reps 20 auto=0x0 t11=0x100
done > $O/r04_p_ntt_rule.log 2>&1
for LOGN in 14 16 17 18 20 22; do
  timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 2 --rounds 6 --reps 10 auto=0x0 t11=0x100
done >> $O/r04_p_ntt_rule.log 2>&1
grep "false" $O/r04_p_ntt_rule.log
grep "tune\"" $O/r04_p_ntt_rule.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['ncomp'], d['variant'], 'ifft', d['ifft_ms_median'], 'fft', d['fft_ms_median'], 'alt', d['alternating_ms_per_transform_median'], d.get('paired_delta_vs_first_pct_median'))
"
