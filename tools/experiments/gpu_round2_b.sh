#!/bin/bash
# Round 2, GPU call B: full parity suite (incl. full-size direct compares and the split MSM), bench with the CPU baseline suite.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())" > $O/info.log 2>&1
(echo -n "cgroup cpu.max: "; cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us; uptime) >> $O/info.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 --maxfail 20 -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; tail -30 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -c 3000 $O/bench.log
