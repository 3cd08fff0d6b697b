#!/bin/bash
# accumulate: end of the following bucket loaded on entering a bucket (-DCSH_ACC_NEXTPF) against the default (which now prefetches the sorted
# index two entries ahead); MSM parity on the default build
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
JOBS="0:0:20 0:0:22 0:0:24 1:0:20 0:1:20"
for rep in 1 2 3; do
  for v in base nextpf; do
    unset COSNARKS_HIP_LIB
    [ $v = nextpf ] && export COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_nextpf.so
    timeout 600 python tools/gpu_msm_loop.py --reps 6 $JOBS > $O/t_${v}_$rep.log 2>&1
  done
done
unset COSNARKS_HIP_LIB
python - <<'PY'
import json
for v in ("base", "nextpf"):
    for rep in (1, 2, 3):
        for ln in open("gpurun_out/t_%s_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print(v, rep, d["curve"], d["group"], d["logn"], "accum", t[3], "reduce", t[4], "total", t[5], "wall", d["wall_ms"])
PY
COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_nextpf.so timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -q --timeout 900 -p no:cacheprovider -x > $O/pytest_t_nextpf.log 2>&1
echo "pytest exit $?" >> $O/pytest_t_nextpf.log; tail -3 $O/pytest_t_nextpf.log
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -p no:cacheprovider -x > $O/pytest_t.log 2>&1
echo "pytest exit $?" >> $O/pytest_t.log; tail -3 $O/pytest_t.log
