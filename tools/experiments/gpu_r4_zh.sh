#!/bin/bash
# Round 4, run ZH: buckets per window-reduction segment (msm_seg_buckets; 0 = one round of waves): more, shorter chains in two or three rounds?
mkdir -p gpurun_out; O=$PWD/gpurun_out
for J in 0:0:20 0:0:18 0:0:22 0:1:20 1:0:20; do
  timeout -s KILL 120 python tools/msm_ab.py --job $J --rounds 8 --reps 10 auto=msm_seg_buckets=0 p12=msm_seg_buckets=12 p9=msm_seg_buckets=9 p7=msm_seg_buckets=7 p5=msm_seg_buckets=5 p24=msm_seg_buckets=24
done > $O/r04_zh_seg_buckets.log 2>&1
grep -v amdgpu.ids $O/r04_zh_seg_buckets.log | grep ms_median | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['job'], r['variant'], r['params_c_W_L_S'], r['ms_median'], r['Mpts_s_median'], r.get('paired_delta_vs_first_pct_median'))"
grep -c "equals_first_variant\": true" $O/r04_zh_seg_buckets.log; grep -c "equals_first_variant\": false" $O/r04_zh_seg_buckets.log
