#!/bin/bash
# Profiles only: kernel trace at 2^20 and 2^24, PMC passes at 2^20 (clean runs: no secondary metrics).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof20 -o msm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/prof20.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof24 -o msm -- python $R/bench.py --log-n 24 --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/prof24.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o msm -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o msm -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/pmc_write.log 2>&1)
tail -1 gpurun_out/prof24.log | cut -c1-300
