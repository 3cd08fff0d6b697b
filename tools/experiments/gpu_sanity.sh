#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py -m gpu -q -x -p no:cacheprovider -k "not closed_form_full and not large_tiled" 2>&1 | grep -E "passed|failed|rror" | tail -2
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
python -c "
import cosnarks_amd as h
for n in (1<<18,1<<20,1200000,1<<22,1<<24): print(n, h.msm_plan(0,n))"
