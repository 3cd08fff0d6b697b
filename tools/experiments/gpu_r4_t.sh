#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_vec_ntt.py tests/test_gpu_trait_path.py -m gpu -q -x -p no:cacheprovider --timeout 300 > $O/r04_t_pytest.log 2>&1; tail -15 $O/r04_t_pytest.log
