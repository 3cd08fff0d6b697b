#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_groth16.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -2
timeout 600 python -c "
import cosnarks_amd.groth16 as g
for ld in (20, 18, 16, 14): print(g.bench_synthetic(0, ld, 4, False))
print(g.bench_synthetic(0, 16, 3, True))
print(g.bench_synthetic(1, 16, 3, False))" 2>&1 | tail -6
