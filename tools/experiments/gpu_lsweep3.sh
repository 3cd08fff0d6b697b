#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py -m gpu -q -x -p no:cacheprovider > gpurun_out/l3_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/l3_tests.log | tail -3
python tools/gpu_lsweep.py 0:0:20 0:0:n=1200000 0:0:19 0:0:18 0:0:n=3000000 0:0:22 1:0:20 0:1:20 1:1:20 0:1:n=1200000 > gpurun_out/lsweep3.log 2>&1
python tools/gpu_lsweep.py --reps 4 --ls 0,128,192,240,256,320,384,448,512,640,768,1024,0 0:0:24 1:0:24 >> gpurun_out/lsweep3.log 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/lsweep3.log"):
    if not line.startswith("{"): print(line.strip()); continue
    d=json.loads(line)
    print(d["curve"],d["group"],d["n"],d["params_c_W"],"auto",[(a["L"],a["accum"],a["tail"],a["total"]) for a in d["auto"]],"best",(d["best"]["L"],d["best"]["accum"],d["best"]["tail"],d["best"]["total"]),"auto/best",d["auto_vs_best"])
PY
