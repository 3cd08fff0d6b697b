#!/bin/bash
# level 2 of the two-level sort, one block per tile-sized slice of a partition (default) against one block per partition (msm_variant 32):
# stage times on uniform scalars and on witness-like ones (LOOP_SKEW: a quarter 0, a quarter 1, a quarter one repeated value), then parity
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for rep in 1 2; do
  for v in 0 32; do
    CSH_MSM_VARIANT=$v timeout 600 python tools/gpu_msm_loop.py --reps 6 0:0:20 0:0:22 0:0:24 1:1:20 > $O/aa_v${v}_$rep.log 2>&1
    CSH_MSM_VARIANT=$v LOOP_SKEW=1 timeout 600 python tools/gpu_msm_loop.py --reps 4 0:0:20 0:0:22 0:0:24 > $O/aa_skew_v${v}_$rep.log 2>&1
  done
done
python - <<'PY'
import json
for kind in ("", "skew_"):
    for v in (0, 32):
        for rep in (1, 2):
            for ln in open("gpurun_out/aa_%sv%d_%d.log" % (kind, v, rep)):
                if ln.startswith("{"):
                    d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                    print(kind + "variant", v, rep, d["curve"], d["group"], d["logn"], "digits+hist", t[0], "scatter", t[2], "accum", t[3], "tail", t[4], "total", t[5])
PY
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_msm_split.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "not ntt" > $O/pytest_aa.log 2>&1
echo "pytest exit $?" >> $O/pytest_aa.log; tail -3 $O/pytest_aa.log
