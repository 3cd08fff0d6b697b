#!/bin/bash
# Round 4, run ZP: per-kernel times of the BN254 G1 MSM at 2^16 and 2^18 (where the tail and the fixed costs dominate).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
cd /tmp
for LG in 16 18; do
  rm -rf $O/prof_small_$LG
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/prof_small_$LG -o s -- python $R/tools/gpu_msm_loop.py --reps 20 0:0:$LG > $O/prof_small_$LG.log 2>&1
  python $R/tools/prof_summary.py $(find $O/prof_small_$LG -name "*.db" | head -1) $O/r04_zp_msm_2p${LG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_msm_loop.py --reps 20 0:0:$LG"
  rm -rf $O/prof_small_$LG
done
cd $R; head -24 $O/r04_zp_msm_2p16_kernel_stats.csv | cut -c1-150; head -24 $O/r04_zp_msm_2p18_kernel_stats.csv | cut -c1-150
