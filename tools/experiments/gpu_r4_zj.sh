#!/bin/bash
# Round 4, run ZJ: plan sweep (window width c, lane length L) at the sizes of real proving keys, interleaved against the cost model's choice.
mkdir -p gpurun_out; O=$PWD/gpurun_out
V18="auto=msm_c=0 c13L20=msm_c=13,msm_l=20 c13L32=msm_c=13,msm_l=32 c13L40=msm_c=13,msm_l=40 c12L27=msm_c=12,msm_l=27 c12L36=msm_c=12,msm_l=36 c14L24=msm_c=14,msm_l=24 c14L32=msm_c=14,msm_l=32 c14L48=msm_c=14,msm_l=48 c15L32=msm_c=15,msm_l=32"
V16="auto=msm_c=0 c11L16=msm_c=11,msm_l=16 c11L24=msm_c=11,msm_l=24 c12L16=msm_c=12,msm_l=16 c12L32=msm_c=12,msm_l=32 c13L16=msm_c=13,msm_l=16 c13L24=msm_c=13,msm_l=24 c10L24=msm_c=10,msm_l=24"
for J in 0:0:18 0:0:19 0:0:17; do timeout -s KILL 150 python tools/msm_ab.py --job $J --rounds 8 --reps 10 $V18; done > $O/r04_zj_plan_sweep.log 2>&1
timeout -s KILL 150 python tools/msm_ab.py --job 0:0:16 --rounds 8 --reps 10 $V16 >> $O/r04_zj_plan_sweep.log 2>&1
grep -v amdgpu.ids $O/r04_zj_plan_sweep.log | grep ms_median | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['job'], r['variant'], r['params_c_W_L_S'], r['ms_median'], r['Mpts_s_median'], r.get('paired_delta_vs_first_pct_median'))"
grep -c "equals_first_variant\": false" $O/r04_zj_plan_sweep.log
