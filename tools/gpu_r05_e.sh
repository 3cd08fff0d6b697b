#!/bin/bash
# round 5, call e: small MSMs -- existing tail variants by interleaved A/B (fused merge, segment lengths, lane lengths), per-kernel stats after balanced windows
O=gpurun_out/r05_e; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for ln in 14 16 18; do
  python tools/msm_ab.py --job 0:0:$ln --rounds 6 --reps 10 default=msm_variant=0 fused=msm_variant=16 seg2=msm_seg_buckets=2 seg3=msm_seg_buckets=3 seg4=msm_seg_buckets=4 seg6=msm_seg_buckets=6 seg10=msm_seg_buckets=10 >> $O/ab_tail_small.log 2>&1
  python tools/msm_ab.py --job 0:0:$ln --rounds 6 --reps 10 default=msm_l=0 l6=msm_l=6 l8=msm_l=8 l12=msm_l=12 l16=msm_l=16 l24=msm_l=24 l32=msm_l=32 >> $O/ab_lane_small.log 2>&1
done
cd /tmp
for ln in 16 18; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$ln -o m -- python $R/tools/gpu_msm_loop.py --reps 20 0:0:$ln > $R/$O/prof_$ln.log 2>&1
  python $R/tools/prof_summary.py $(find $R/$O/prof_$ln -name "*.db" | head -1) $R/$O/msm_2p${ln}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_msm_loop.py --reps 20 0:0:$ln (round 5: balanced windows)"
  rm -rf $R/$O/prof_$ln
done
cd $R
grep -h '"tune"' $O/ab_tail_small.log $O/ab_lane_small.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['job'],d['variant'],d['params_c_W_L_S'],d['ms_median'],d.get('paired_delta_vs_first_pct_median'))"
cat $O/msm_2p16_kernel_stats.csv | cut -c1-120
