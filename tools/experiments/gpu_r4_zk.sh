#!/bin/bash
# Round 4, run ZK: short lanes at small sizes (the lane-length model starts at 16 and assumes one wave saturates a SIMD).
mkdir -p gpurun_out; O=$PWD/gpurun_out
V="auto=msm_c=0"
for c in 10 11 12 13; do for l in 8 12 16 20; do V="$V c${c}L${l}=msm_c=$c,msm_l=$l"; done; done
for J in 0:0:14 0:0:15 0:0:16 0:0:17; do timeout -s KILL 200 python tools/msm_ab.py --job $J --rounds 6 --reps 10 $V; done > $O/r04_zk_short_lanes.log 2>&1
V="auto=msm_c=0 c13L16=msm_c=13,msm_l=16 c13L20=msm_c=13,msm_l=20 c13L24=msm_c=13,msm_l=24 c13L27=msm_c=13,msm_l=27 c12L16=msm_c=12,msm_l=16 c12L20=msm_c=12,msm_l=20 c14L16=msm_c=14,msm_l=16 c14L20=msm_c=14,msm_l=20"
timeout -s KILL 200 python tools/msm_ab.py --job 0:0:18 --rounds 6 --reps 10 $V >> $O/r04_zk_short_lanes.log 2>&1
grep -v amdgpu.ids $O/r04_zk_short_lanes.log | grep ms_median | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['job'], r['variant'], r['params_c_W_L_S'], r['ms_median'], r['Mpts_s_median'], r.get('paired_delta_vs_first_pct_median'))"
grep -c "equals_first_variant\": false" $O/r04_zk_short_lanes.log
