#!/bin/bash
mkdir -p gpurun_out
python tools/gpu_lsweep.py --ls 0,16,20,24,26,28,30,32,33,34,35,36,37,38,40,42,44,45,46,47,48,52,54,55,56,60,64,68,69,70,80,90,91,92,93,96,128,136,138,140,0 0:0:20 > gpurun_out/lsweep2.log 2>&1
python tools/gpu_lsweep.py 0:0:n=1200000 0:0:19 0:0:21 0:0:18 1:0:20 0:1:20 2:0:20 >> gpurun_out/lsweep2.log 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/lsweep2.log"):
    if not line.startswith("{"): print(line.strip()); continue
    d=json.loads(line)
    print(d["curve"],d["group"],d["n"],d["params_c_W"],"auto",d["auto"],"best",d["best"],"auto/best",d["auto_vs_best"])
    print("   ", " ".join(f'{r["L"]}:{r["accum"]}+{r["tail"]}={r["total"]}' for r in d["rows"]))
PY
