#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/spin.log
for s in 0 250 0 250 0 1000; do
  sleep 3
  echo -n "spinup $s: " >> gpurun_out/spin.log
  python bench.py --steps 20 --warmup 5 --spinup $s --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), d['roofline']['stage_ms']['accum'])" >> gpurun_out/spin.log
done
cat gpurun_out/spin.log
