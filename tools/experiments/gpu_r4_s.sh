#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out
for A in 1; do echo "== helper-thread deadline"; timeout -s KILL 45 python tools/experiments/rccl_deadline_probe.py $A 2>&1 | grep -v amdgpu.ids; echo "exit $?"; done > $O/r04_s2_rccl_deadline.log 2>&1
cat $O/r04_s2_rccl_deadline.log
