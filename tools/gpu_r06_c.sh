#!/bin/bash
# round 6 c: kernel trace of the wide-window table MSM (2^20 c = 17 / 20, 2^24 c = 20), lb = 11 at 2^22 / 2^24
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
db() { find $1 -name "*.db" | head -1; }
for j in "17 9 20" "20 10 20" "20 10 24"; do
  set -- $j
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_w -o msm -- python $R/tools/msm_wide_probe.py --profile --c $1 --lb $2 --reps 20 --warm 5 0:0:$3 > $R/gpurun_out/r06_c_prof_c$1_2p$3.log 2>&1)
  python tools/prof_summary.py $(db gpurun_out/prof_w) gpurun_out/r06_c_wide_c$1_lb$2_2p$3_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/msm_wide_probe.py --profile --c $1 --lb $2 --reps 20 --warm 5 0:0:$3 (one plain call, 31 table calls)"
  rm -rf gpurun_out/prof_w
done
(timeout 400 python tools/msm_wide_probe.py --c 19 20 --lb 10 11 --reps 8 --warm 5 0:0:22 2>&1 | tail -30) > gpurun_out/r06_c_wide_probe_2p22.log
(timeout 600 python tools/msm_wide_probe.py --c 20 --lb 10 11 --chunks 0 512 2048 --reps 5 --warm 3 0:0:24 2>&1 | tail -30) > gpurun_out/r06_c_wide_probe_2p24.log
