#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/c1516.log
for c in 0 15 0 15; do
  echo "== CSH_MSM_C=$c" >> gpurun_out/c1516.log
  CSH_MSM_C=$c python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['msm_params'], d['roofline']['stage_ms'])" >> gpurun_out/c1516.log
  CSH_MSM_C=$c python tools/gpu_msm_loop.py --reps 10 0:0:20 0:1:20 2:0:20 >> gpurun_out/c1516.log 2>&1
done
timeout 600 python -c "
import cosnarks_amd.groth16 as g
for _ in range(2): print(g.bench_synthetic(0, 20, 3, False))" >> gpurun_out/c1516.log 2>&1
CSH_MSM_C=15 timeout 600 python -c "
import cosnarks_amd.groth16 as g
for _ in range(2): print('c15', g.bench_synthetic(0, 20, 3, False))" >> gpurun_out/c1516.log 2>&1
cat gpurun_out/c1516.log | sed 's/"params_c_W_L_S"/p/; s/"ms_digits_scan_scatter_accum_reduce_total"/ms/; s/"curve": //; s/"group": //; s/"logn": //' | cut -c1-330
