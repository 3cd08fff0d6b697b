#!/bin/bash
# dual-chain scan (two independent multiplications in lockstep, FpS::reduce_scan2) vs the single-chain build (-DCSH_DUAL_CHAIN=0,
# gpurun_ab/libcosnarks_hip_single.so): accumulate stage times per group, NTT, interleaved A/B/A/B on one box; MSM/NTT parity on the dual build
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
JOBS="0:0:20 0:0:24 0:1:20 1:0:20 1:1:20"
for rep in 1 2; do
  for v in dual single; do
    if [ $v = single ]; then export COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_single.so; else unset COSNARKS_HIP_LIB; fi
    timeout 600 python tools/gpu_msm_loop.py --reps 6 $JOBS > $O/ab_${v}_$rep.log 2>&1
    NTT_LOGN=16,20,22 timeout 300 python tools/gpu_probe_ntt.py 2>&1 | grep '"op": "ntt"' > $O/ab_ntt_${v}_$rep.log
  done
done
unset COSNARKS_HIP_LIB
python - <<'PY'
import json, glob
for v in ("dual", "single"):
    for rep in (1, 2):
        for ln in open("gpurun_out/ab_%s_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print(v, rep, d["curve"], d["group"], d["logn"], "accum", t[3], "reduce", t[4], "total", t[5], "wall", d["wall_ms"])
        for ln in open("gpurun_out/ab_ntt_%s_%d.log" % (v, rep)):
            d = json.loads(ln); print(v, rep, "ntt", d["logn"], d["ncomp"], d["ifft_ms"], d["fft_ms"])
PY
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_vec_ntt.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -p no:cacheprovider -x > $O/pytest_dual.log 2>&1
echo "pytest exit $?" >> $O/pytest_dual.log; tail -3 $O/pytest_dual.log
