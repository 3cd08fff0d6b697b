#!/bin/bash
# round 6 b: kernel trace of the wide-window table MSM at 2^20 (c = 17, 20), segment sweep, probes at 2^22 / 2^24
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for c in 17 20; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_wide_c$c -o msm -- python $R/tools/msm_wide_probe.py --c $c --lb 9 --reps 40 --warm 10 0:0:20 > $R/gpurun_out/r06_b_prof_c$c.log 2>&1)
  db=$(ls gpurun_out/prof_wide_c$c/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db gpurun_out/r06_b_wide_c${c}_2p20_kernel_stats.csv "msm_wide_probe.py --c $c --lb 9 0:0:20 (plain + table calls in one process)"
done
(timeout 300 python tools/msm_wide_probe.py --c 17 19 20 --lb 9 --seg 0 2 4 8 16 --reps 20 0:0:20 2>&1 | tail -30) > gpurun_out/r06_b_seg_sweep_2p20.log
(timeout 400 python tools/msm_wide_probe.py --c 18 20 21 22 --lb 8 9 10 --reps 8 --warm 5 0:0:22 2>&1 | tail -30) > gpurun_out/r06_b_wide_probe_2p22.log
(timeout 600 python tools/msm_wide_probe.py --c 20 21 22 --lb 8 9 10 --reps 5 --warm 3 0:0:24 2>&1 | tail -30) > gpurun_out/r06_b_wide_probe_2p24.log
rm -rf gpurun_out/prof_wide_c17 gpurun_out/prof_wide_c20
