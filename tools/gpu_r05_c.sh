#!/bin/bash
# round 5, call c: balanced windows with the tuned window count -- parity, then a sweep of W around it per size, NTT beyond 2^23
O=gpurun_out/r05_c; mkdir -p $O
cat /sys/fs/cgroup/memory.max > $O/host.log 2>&1; free -g >> $O/host.log 2>&1; nproc >> $O/host.log
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_msm_split.py -x -q -m gpu -k "balanced or fuzz or window or closed_form or edge or small or split or beyond" --durations=8 > $O/pytest_msm.log 2>&1; echo "pytest exit $?" >> $O/pytest_msm.log
for ln in 13 14 15 16 17 18 19 20; do
  WU=$(python - <<PY
import cosnarks_amd as hip
from cosnarks_amd import bindings as B
import ctypes as C
out=(C.c_uint32*6)()
B.tune_set("msm_balanced",0)
B._check(hip.lib().csh_msm_plan(0, C.c_size_t(1<<$ln), out))
print(out[1])
PY
)
  python tools/msm_ab.py --job 0:0:$ln --rounds 6 --reps 10 uniform=msm_balanced=0 wu=msm_w=$WU wm1=msm_w=$((WU-1)) wm2=msm_w=$((WU-2)) wp1=msm_w=$((WU+1)) >> $O/ab_w_sweep.log 2>&1
done
tail -5 $O/pytest_msm.log; grep -h '"tune"' $O/ab_w_sweep.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['job'],d['variant'],d['params_c_W_L_S'],d['ms_median'],d.get('paired_delta_vs_first_pct_median'))"
