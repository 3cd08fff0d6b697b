#!/bin/bash
# round 6 e: which window width for which key size (policy), all groups at 2^20
mkdir -p gpurun_out
(timeout 300 python tools/msm_wide_probe.py --c 17 18 --reps 30 0:0:17 0:0:18 2>&1 | tail -20) > gpurun_out/r06_e_policy_small.log
(timeout 300 python tools/msm_wide_probe.py --c 17 18 19 20 --reps 20 0:0:19 0:0:21 2>&1 | tail -20) > gpurun_out/r06_e_policy_mid.log
(timeout 300 python tools/msm_wide_probe.py --c 19 20 21 22 --reps 6 --warm 4 0:0:23 2>&1 | tail -20) > gpurun_out/r06_e_policy_2p23.log
(timeout 600 python tools/msm_wide_probe.py --c 17 18 19 20 --reps 10 --warm 5 0:1:20 1:0:20 1:1:20 2:0:20 2>&1 | tail -40) > gpurun_out/r06_e_groups_2p20.log
(timeout 300 python tools/msm_wide_probe.py --skewed --c 17 20 --reps 10 --warm 5 0:0:20 0:0:22 2>&1 | tail -20) > gpurun_out/r06_e_skewed.log
