#!/bin/bash
# One GPU-box session: parity tests, probe, bench, rocprof. Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import cosnarks_amd as h; print('devices', h.device_count(), h.lib().csh_version())" > gpurun_out/info.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 >> gpurun_out/info.log
nproc >> gpurun_out/info.log; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/info.log
timeout ${T_TESTS:-1500} python -m pytest tests -m gpu -q --timeout 900 --maxfail 10 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu_probe.py > gpurun_out/probe.log 2>&1
tail -30 gpurun_out/probe.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1
tail -5 gpurun_out/bench.log
if [ "${DO_PROF:-1}" = "1" ]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o msm -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-check > $OLDPWD/gpurun_out/prof.log 2>&1)
  find gpurun_out/prof -name "*stats*" | head; 
fi
