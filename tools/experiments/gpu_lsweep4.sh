#!/bin/bash
mkdir -p gpurun_out
python tools/gpu_lsweep.py --reps 5 --ls 0,16,20,24,28,32,35,36,40,46,48,55,56,64,69,72,92,100,138,144 0:0:20 0:1:20 2:0:20 > gpurun_out/lsweep4.log 2>&1
python tools/gpu_lsweep.py --reps 5 --ls 0,16,18,20,24,28,32,36,40,46,50,64,72 0:0:19 >> gpurun_out/lsweep4.log 2>&1
python tools/gpu_lsweep.py --reps 5 --ls 0,16,20,22,27,28,32,42,44,48 0:0:18 >> gpurun_out/lsweep4.log 2>&1
python tools/gpu_lsweep.py --reps 5 --ls 0,20,26,32,36,40,54,63,64,80,108,160 0:0:n=1200000 >> gpurun_out/lsweep4.log 2>&1
python tools/gpu_lsweep.py --reps 5 --ls 0,32,38,44,52,54,64,128 1:0:20 >> gpurun_out/lsweep4.log 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/lsweep4.log"):
    if not line.startswith("{"): print(line.strip()); continue
    d=json.loads(line)
    print(d["curve"],d["group"],d["n"],d["params_c_W"],"auto",[(a["L"],a["accum"],a["tail"],a["total"]) for a in d["auto"]],"best",(d["best"]["L"],d["best"]["total"]),"auto/best",d["auto_vs_best"])
    print("   ", " ".join(f'{r["L"]}:{r["accum"]}+{r["tail"]}={r["total"]}' for r in d["rows"]))
PY
