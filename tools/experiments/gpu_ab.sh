#!/bin/bash
# A/B of library variants built by tools/experiments/build_variant.sh on ONE box: usage gpu_ab.sh "<jobs>" name1 name2 ...
mkdir -p gpurun_out
jobs=$1; shift
: > gpurun_out/ab.log
for round in 1 2 3; do
  for v in "$@"; do
    echo "== $v (round $round)" >> gpurun_out/ab.log
    COSNARKS_HIP_LIB=$PWD/gpurun_ab/libcosnarks_hip_$v.so python tools/gpu_msm_loop.py --reps 10 $jobs >> gpurun_out/ab.log 2>&1
  done
done
python - <<'PY'
import json, re, collections
cur=None; acc=collections.defaultdict(list)
for line in open("gpurun_out/ab.log"):
    if line.startswith("=="): cur=line.split()[1]; continue
    if line.startswith("{"):
        d=json.loads(line); acc[(cur,d["curve"],d["group"],d["logn"])].append((d["ms_digits_scan_scatter_accum_reduce_total"][3], d["ms_digits_scan_scatter_accum_reduce_total"][4], d["wall_ms"]))
for k,v in sorted(acc.items(), key=lambda kv:(kv[0][1:],kv[0][0])):
    print(k, "accum", [x[0] for x in v], "tail", [x[1] for x in v], "wall", [x[2] for x in v])
PY
