#!/bin/bash
# final tree: bench line with the live probes, the new CPU-tested placement path through the GPU tests
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_groth16.py -m gpu -q --timeout 900 -p no:cacheprovider -k "placed or clone or prover_devices" > $O/pytest_placed.log 2>&1
echo "pytest exit $?" >> $O/pytest_placed.log; grep -E "passed|failed" $O/pytest_placed.log | tail -2
timeout 900 python bench.py > $O/bench.log 2>&1; tail -c 300 $O/bench.log
python - <<'PY'
import json
d=json.loads([x for x in open("gpurun_out/bench.log") if x.startswith("{")][-1])
print("headline", round(d["value"]/1e6,1), round(d["ms_per_step"],4), d["roofline"]["alu"])
print("2^24", d["secondary"]["msm_bn254_g1_2p24"]["ms"], d["secondary"]["msm_bn254_g1_2p24"]["roofline"]["alu"]["frac"])
print("ntt", d["secondary"]["ntt_bn254_2p22"]["ms"], d["secondary"]["ntt_bn254_2p22"]["roofline"]["alu"])
print("prove", d["secondary"]["groth16_prove_synthetic_2p20"]["prove_ms"])
PY
