#!/bin/bash
# Round 4, run C: the caller-mask witness map + the trait path: parity, then the synthetic 2^20 / 2^18 prove both ways.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_trait_path.py tests/test_gpu_groth16.py -m gpu -q -x -p no:cacheprovider > $O/r04_c_pytest.log 2>&1; tail -5 $O/r04_c_pytest.log
timeout 600 python - > $O/r04_c_trait_path.log 2>&1 <<'PY'
import json
import cosnarks_amd as hip
from cosnarks_amd import groth16 as g
for logn in (20, 18):
    print(json.dumps(g.bench_synthetic(hip.BN254, logn, 4, with_rep3=False)))
for pop in (0, 1, 2, 4, 8):
    with hip.tuned(host_populate=pop):
        r = g.bench_synthetic(hip.BN254, 20, 3, with_rep3=False)
        print(json.dumps({"host_populate": pop, "witness_map_ms": r["witness_map_ms"], "trait_path_ms": r["trait_path_ms"], "phases": r["trait_path_phases_ms"]}))
PY
cat $O/r04_c_trait_path.log
