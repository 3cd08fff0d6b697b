#!/bin/bash
# Round 4, run ZO: lane lengths that fill WHOLE occupancy rounds (3 waves per SIMD together on BN254 G1) at the large sizes.
mkdir -p gpurun_out; O=$PWD/gpurun_out
run() { timeout -s KILL 250 python tools/msm_ab.py --job $1 --rounds $2 --reps $3 "${@:4}"; }
{
run 0:0:20 10 10 auto=msm_l=0 L46=msm_l=46 L50=msm_l=50 L61=msm_l=61 L69=msm_l=69 L76=msm_l=76 L91=msm_l=91 L100=msm_l=100
run 0:0:22 6 5 auto=msm_l=0 L114=msm_l=114 L103=msm_l=103 L137=msm_l=137 L171=msm_l=171 L86=msm_l=86
run 0:0:24 4 3 auto=msm_l=0 L228=msm_l=228 L273=msm_l=273 L205=msm_l=205 L342=msm_l=342
run 0:0:21 6 8 auto=msm_l=0 L57=msm_l=57 L86=msm_l=86 L114=msm_l=114 L171=msm_l=171
} > $O/r04_zo_occupancy_rounds.log 2>&1
grep -v amdgpu.ids $O/r04_zo_occupancy_rounds.log | grep ms_median | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['job'], r['variant'], r['params_c_W_L_S'], r['ms_median'], r['Mpts_s_median'], r.get('paired_delta_vs_first_pct_median'))"
grep -c "equals_first_variant\": false" $O/r04_zo_occupancy_rounds.log
