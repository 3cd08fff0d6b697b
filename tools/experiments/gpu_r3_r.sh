#!/bin/bash
# NTT: 2^12-element tiles (144 KiB of LDS, one tile per CU) so that a 2^22 transform is two sweeps (12 + 10 stages, 128-byte runs);
# tune "ntt_variant": bits 4-6 run_log + 1, bit 7 tile 2^12, bit 9 1024 lanes per radix-4 tile. A/B on one box + parity with the variants forced.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for rep in 1 2; do
  for v in 0 48 176 688; do
    CSH_NTT_VARIANT=$v NTT_LOGN=20,21,22,23,24 timeout 300 python tools/gpu_probe_ntt.py 2>&1 | grep '"op": "ntt"' > $O/r_ntt_v${v}_$rep.log
  done
done
python - <<'PY'
import json
for v in (0, 48, 176, 688):
    for rep in (1, 2):
        for ln in open("gpurun_out/r_ntt_v%d_%d.log" % (v, rep)):
            d = json.loads(ln); print("variant", v, "rep", rep, "logn", d["logn"], "ncomp", d["ncomp"], d["ifft_ms"], d["fft_ms"])
PY
for v in 176 688; do
CSH_NTT_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_vec_ntt.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "ntt or fft or domain or witness or h_ or golden" > $O/pytest_r_$v.log 2>&1
echo "pytest exit $?" >> $O/pytest_r_$v.log; tail -3 $O/pytest_r_$v.log
done
