#!/bin/bash
# round 5, call f: lane lengths at 2^15 .. 2^17 after balanced windows (the occupancy rule's floor of 8 entries), G1 groups
O=gpurun_out/r05_f; mkdir -p $O
for job in 0:0:15 0:0:16 0:0:17 1:0:15 1:0:16 1:0:17 2:0:16; do
  python tools/msm_ab.py --job $job --rounds 6 --reps 10 default=msm_l=0 l8=msm_l=8 l10=msm_l=10 l12=msm_l=12 l14=msm_l=14 l16=msm_l=16 l20=msm_l=20 >> $O/ab_lane_floor.log 2>&1
done
grep -h '"tune"' $O/ab_lane_floor.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['job'],d['variant'],d['params_c_W_L_S'],d['ms_median'],d.get('paired_delta_vs_first_pct_median'))"
