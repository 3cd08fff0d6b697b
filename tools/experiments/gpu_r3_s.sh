#!/bin/bash
# accumulate: sorted index prefetched two entries ahead (-DCSH_ACC_PREFETCH2, gpurun_ab/libcosnarks_hip_pf2.so) against the default, workgroup
# size 64 against 128; NTT 2^21 / 2^23 with 64-byte runs (ntt_variant 32)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
JOBS="0:0:20 0:0:22 0:0:24 1:0:20"
for rep in 1 2; do
  for v in base pf2 blk64; do
    unset COSNARKS_HIP_LIB CSH_ACC_BLK
    [ $v = pf2 ] && export COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_pf2.so
    [ $v = blk64 ] && export CSH_ACC_BLK=64
    timeout 600 python tools/gpu_msm_loop.py --reps 6 $JOBS > $O/s_${v}_$rep.log 2>&1
  done
done
unset COSNARKS_HIP_LIB CSH_ACC_BLK
for v in 0 32; do CSH_NTT_VARIANT=$v NTT_LOGN=19,21,23 timeout 300 python tools/gpu_probe_ntt.py 2>&1 | grep '"op": "ntt"' > $O/s_ntt_v$v.log; done
python - <<'PY'
import json
for v in ("base", "pf2", "blk64"):
    for rep in (1, 2):
        for ln in open("gpurun_out/s_%s_%d.log" % (v, rep)):
            if ln.startswith("{"):
                d = json.loads(ln); t = d["ms_digits_scan_scatter_accum_reduce_total"]
                print(v, rep, d["curve"], d["group"], d["logn"], "accum", t[3], "reduce", t[4], "total", t[5], "wall", d["wall_ms"])
for v in (0, 32):
    for ln in open("gpurun_out/s_ntt_v%d.log" % v):
        d = json.loads(ln); print("ntt variant", v, d["logn"], d["ncomp"], d["ifft_ms"], d["fft_ms"])
PY
