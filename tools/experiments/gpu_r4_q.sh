#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_schema.py -m gpu -q -x -p no:cacheprovider > $O/r04_q_pytest_bench.log 2>&1; tail -30 $O/r04_q_pytest_bench.log
timeout 900 python bench.py > $O/r04_q_bench.log 2>&1; tail -c 400 $O/r04_q_bench.log
