#!/bin/bash
# round 5, call g: lane length 2 L in the narrow windows (msm_variant bit 6 = one length) -- parity, then A/B with forced L around the plan's
O=gpurun_out/r05_g; mkdir -p $O
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_msm_split.py -x -q -m gpu -k "balanced or fuzz or window or edge or small or variants or split_matches or multi" > $O/pytest_msm.log 2>&1; echo "pytest exit $?" >> $O/pytest_msm.log
for job in 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 1:0:16 1:0:17 0:1:16; do
  python tools/msm_ab.py --job $job --rounds 6 --reps 10 one=msm_variant=64 two=msm_variant=0 two_l8=msm_l=8 two_l10=msm_l=10 two_l12=msm_l=12 two_l14=msm_l=14 two_l16=msm_l=16 one_l12=msm_variant=64,msm_l=12 one_l14=msm_variant=64,msm_l=14 >> $O/ab_ln.log 2>&1
done
tail -4 $O/pytest_msm.log
grep -h '"tune"' $O/ab_ln.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['job'],d['variant'],d['params_c_W_L_S'],d['ms_median'],d.get('paired_delta_vs_first_pct_median'))"
