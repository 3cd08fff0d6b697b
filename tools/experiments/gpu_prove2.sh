#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_plonk_honk.py -m gpu -q -x -p no:cacheprovider > gpurun_out/prove2_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/prove2_tests.log | tail -3
for t in 4 0 2 4 0; do
echo "== COG16_TABLES=$t" 
COG16_TABLES=$t timeout 600 python -c "
import cosnarks_amd.groth16 as g
for _ in range(2): print(g.bench_synthetic(0, 20, 4, False))
print(g.bench_synthetic(0, 18, 4, False))" 2>&1 | tail -3
done
COG16_TABLES=4 timeout 600 python -c "
import cosnarks_amd.groth16 as g
print(g.bench_synthetic(0, 20, 3, True))" 2>&1 | tail -1
