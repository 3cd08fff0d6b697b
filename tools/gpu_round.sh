#!/bin/bash
# One GPU-box session: parity tests, probes, bench, rocprof (kernel trace + PMC). Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import cosnarks_amd as h; print('devices', h.device_count(), h.lib().csh_version())" > gpurun_out/info.log 2>&1
nproc >> gpurun_out/info.log; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/info.log
timeout ${T_TESTS:-1500} python -m pytest tests -m gpu -q --timeout 900 --maxfail 10 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 600 python tools/gpu_probe.py > gpurun_out/probe.log 2>&1; tail -4 gpurun_out/probe.log
timeout 600 python tools/gpu_probe_ntt.py > gpurun_out/probe_ntt.log 2>&1; tail -3 gpurun_out/probe_ntt.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log
R=$PWD
if [ "${DO_PROF:-1}" = "1" ]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o msm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/prof.log 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o msm -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o msm -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/pmc_write.log 2>&1)
  ls gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
fi
