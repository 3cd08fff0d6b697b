#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_msm.py -m gpu -q -x -p no:cacheprovider -k "not closed_form_full" > gpurun_out/prove_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/prove_tests.log | tail -2
timeout 600 python -c "
import cosnarks_amd.groth16 as g
for _ in range(3): print(g.bench_synthetic(0, 20, 4, False))
print(g.bench_synthetic(0, 20, 3, True))" > gpurun_out/prove.log 2>&1
cat gpurun_out/prove.log
