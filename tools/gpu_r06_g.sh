#!/bin/bash
# round 6 g: timeline (kernel starts / durations / gaps) of one table MSM at 2^16 and 2^20, c = 17
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
db() { find $1 -name "*.db" | head -1; }
for n in 16 20; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_w -o msm -- python $R/tools/msm_wide_probe.py --profile --c 17 --reps 20 --warm 20 0:0:$n > $R/gpurun_out/r06_g_prof_2p$n.log 2>&1)
  python tools/prof_timeline.py $(db gpurun_out/prof_w) gpurun_out/r06_g_wide_c17_2p${n}_timeline.csv 30
  python tools/prof_summary.py $(db gpurun_out/prof_w) gpurun_out/r06_g_wide_c17_2p${n}_kernel_stats.csv "msm_wide_probe.py --profile --c 17 0:0:$n"
  rm -rf gpurun_out/prof_w
done
