#!/bin/bash
for t in 4 8 16 4; do
echo "== COG16_TABLES=$t"
COG16_TABLES=$t timeout 600 python -c "
import cosnarks_amd.groth16 as g
for _ in range(2): print(g.bench_synthetic(0, 20, 4, False))
print(g.bench_synthetic(0, 18, 4, False))
print(g.bench_synthetic(0, 16, 4, False))" 2>&1 | tail -4
done
