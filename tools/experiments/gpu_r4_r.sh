#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 600 python -m pytest tests/test_gpu_msm_split.py tests/test_gpu_bench_schema.py tests/test_gpu_plonk_honk.py tests/test_gpu_plonk_vectors.py -m gpu -q -x -p no:cacheprovider --timeout 300 > $O/r04_r_pytest.log 2>&1; tail -15 $O/r04_r_pytest.log
