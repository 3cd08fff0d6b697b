#!/bin/bash
# Round 4, run ZL: the occupancy-aware lane length (new auto) against the former choice (forced msm_l) and neighbours, G1 groups.
mkdir -p gpurun_out; O=$PWD/gpurun_out
run() { timeout -s KILL 200 python tools/msm_ab.py --job $1 --rounds 8 --reps 10 "${@:2}"; }
{
run 0:0:15 auto=msm_l=0 old16=msm_l=16 L8=msm_l=8 L12=msm_l=12
run 0:0:16 auto=msm_l=0 old23=msm_l=23 L12=msm_l=12 L16=msm_l=16
run 0:0:17 auto=msm_l=0 old21=msm_l=21 L12=msm_l=12 L16=msm_l=16
run 0:0:18 auto=msm_l=0 old27=msm_l=27 L16=msm_l=16
run 0:0:19 auto=msm_l=0 old46=msm_l=46 L32=msm_l=32
run 0:0:20 auto=msm_l=0 old55=msm_l=55
run 1:0:15 auto=msm_l=0 old=msm_l=16 L8=msm_l=8 L12=msm_l=12 L24=msm_l=24
run 1:0:16 auto=msm_l=0 L8=msm_l=8 L12=msm_l=12 L16=msm_l=16 L24=msm_l=24 L32=msm_l=32
run 1:0:17 auto=msm_l=0 L12=msm_l=12 L16=msm_l=16 L24=msm_l=24 L32=msm_l=32
run 1:0:18 auto=msm_l=0 L16=msm_l=16 L24=msm_l=24 L32=msm_l=32 L40=msm_l=40
run 2:0:16 auto=msm_l=0 old23=msm_l=23 L12=msm_l=12
run 0:1:16 auto=msm_l=0 L12=msm_l=12 L16=msm_l=16
} > $O/r04_zl_lane_rule.log 2>&1
grep -v amdgpu.ids $O/r04_zl_lane_rule.log | grep ms_median | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['job'], r['variant'], r['params_c_W_L_S'], r['ms_median'], r['Mpts_s_median'], r.get('paired_delta_vs_first_pct_median'))"
grep -c "equals_first_variant\": false" $O/r04_zl_lane_rule.log; grep -i "error\|Traceback" $O/r04_zl_lane_rule.log | head -3
