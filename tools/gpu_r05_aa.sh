#!/bin/bash
# round 5, call aa: load-ahead in the merge / window-reduction kernels (msm_variant bit 7 = off) -- parity (plan fuzz), interleaved A/B per size and group
O=gpurun_out/r05_aa; mkdir -p $O
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fuzz or variants or window or balanced or edge or small" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
for job in 0:0:14 0:0:16 0:0:18 0:0:20 0:0:22 1:0:16 1:0:20 0:1:16 0:1:20; do
  python tools/msm_ab.py --job $job --rounds 8 --reps 10 plain=msm_variant=128 ahead=msm_variant=0 >> $O/ab_load_ahead.log 2>&1
done
tail -3 $O/pytest.log
grep -h '"tune"' $O/ab_load_ahead.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['job'],d['variant'],d['params_c_W_L_S'],d['ms_median'],d.get('paired_delta_vs_first_pct_median'),d.get('paired_delta_vs_first_pct_min_max'))"
