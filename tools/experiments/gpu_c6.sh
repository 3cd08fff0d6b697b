export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
PROBE_LOGN=20 PROBE_C=0 timeout 300 python tools/gpu_probe.py > $O/probe_c6.log 2>&1; head -1 $O/probe_c6.log
timeout 300 python tools/bench_single_process_split.py --devices 1 --log-n 20 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_groth16.py tests/test_gpu_vec_ntt.py -m gpu -q -x -p no:cacheprovider > $O/pytest_c6.log 2>&1; grep -E "passed|failed|error" $O/pytest_c6.log | tail -3
