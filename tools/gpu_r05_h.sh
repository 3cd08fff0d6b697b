#!/bin/bash
# round 5, call h: lane lengths of BLS12-381 G1 / Grumpkin at 2^16 .. 2^19 (does the occupancy rule want 3 waves per SIMD there too?)
O=gpurun_out/r05_h; mkdir -p $O
python tools/msm_ab.py --job 1:0:18 --rounds 6 --reps 8 default=msm_l=0 l20=msm_l=20 l24=msm_l=24 l27=msm_l=27 l32=msm_l=32 l36=msm_l=36 l44=msm_l=44 >> $O/ab_lane_bls.log 2>&1
python tools/msm_ab.py --job 1:0:19 --rounds 6 --reps 8 default=msm_l=0 l36=msm_l=36 l44=msm_l=44 l48=msm_l=48 l56=msm_l=56 l64=msm_l=64 l72=msm_l=72 >> $O/ab_lane_bls.log 2>&1
python tools/msm_ab.py --job 1:0:20 --rounds 6 --reps 6 default=msm_l=0 l48=msm_l=48 l56=msm_l=56 l64=msm_l=64 l80=msm_l=80 l96=msm_l=96 >> $O/ab_lane_bls.log 2>&1
python tools/msm_ab.py --job 1:0:15 --rounds 6 --reps 10 default=msm_l=0 l6=msm_l=6 l8=msm_l=8 l10=msm_l=10 l12=msm_l=12 >> $O/ab_lane_bls.log 2>&1
python tools/msm_ab.py --job 2:0:17 --rounds 6 --reps 10 default=msm_l=0 l12=msm_l=12 l14=msm_l=14 l16=msm_l=16 >> $O/ab_lane_bls.log 2>&1
python tools/msm_ab.py --job 0:0:19 --rounds 6 --reps 8 default=msm_l=0 l32=msm_l=32 l40=msm_l=40 l46=msm_l=46 l52=msm_l=52 l64=msm_l=64 >> $O/ab_lane_bls.log 2>&1
grep -h '"tune"' $O/ab_lane_bls.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['job'],d['variant'],d['params_c_W_L_S'],d['ms_median'],d.get('paired_delta_vs_first_pct_median'))"
