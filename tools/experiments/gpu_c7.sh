export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_vec_ntt.py tests/test_gpu_fullsize.py tests/test_gpu_plonk_honk.py tests/test_gpu_groth16.py -m gpu -q -x -p no:cacheprovider -k "ntt or fft or groth16 or plonk or honk or prove" > $O/pytest_c7.log 2>&1; grep -E "passed|failed|error" $O/pytest_c7.log | tail -3
NTT_LOGN=16,20,22,24 timeout 300 python tools/gpu_probe_ntt.py 2>&1 | grep '"op": "ntt"' > $O/ntt_c7.log; cat $O/ntt_c7.log
