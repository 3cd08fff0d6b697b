#!/bin/bash
# fused h pipeline (coset table on the last inverse-transform pass, a b - c in one kernel) vs the unfused sequence; full suite
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail 20 -p no:cacheprovider --durations=5 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
for u in 0 1 0 1; do echo "== CSH_H_UNFUSED=$u"; CSH_H_UNFUSED=$u timeout 300 python tools/bench_prove_devices.py --devices 0 --steps 9 --warmup 3; CSH_H_UNFUSED=$u timeout 300 python tools/bench_prove_devices.py --devices 0 --steps 9 --warmup 3 --log-n 16; done > $O/h_fused.log 2>&1; grep -E "==|prove_ms" $O/h_fused.log | cut -c1-330
