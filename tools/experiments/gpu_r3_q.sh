#!/bin/bash
# NTT: shorter contiguous runs in the strided passes -> fewer sweeps (tune "ntt_variant" bits 4-6 = run_log + 1; 2^22: 11 + 11 stages
# with 32-byte runs instead of 11 + 6 + 5 with 256-byte runs). A/B on one box, then the NTT parity suites with the variant forced.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for rep in 1 2; do
  for v in 0 16 32 48; do
    CSH_NTT_VARIANT=$v NTT_LOGN=18,20,22,24 timeout 300 python tools/gpu_probe_ntt.py 2>&1 | grep '"op": "ntt"' > $O/q_ntt_v${v}_$rep.log
  done
done
python - <<'PY'
import json
for v in (0, 16, 32, 48):
    for rep in (1, 2):
        for ln in open("gpurun_out/q_ntt_v%d_%d.log" % (v, rep)):
            d = json.loads(ln); print("variant", v, "rep", rep, "logn", d["logn"], "ncomp", d["ncomp"], d["ifft_ms"], d["fft_ms"])
PY
CSH_NTT_VARIANT=16 timeout 900 python -m pytest tests/test_gpu_vec_ntt.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "ntt or fft or domain or witness or h_ or golden" > $O/pytest_q.log 2>&1
echo "pytest exit $?" >> $O/pytest_q.log; tail -3 $O/pytest_q.log
