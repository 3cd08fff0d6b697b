#!/bin/bash
# Round 3, first GPU call: parity suite on the new tree (product-scanning reduction, placed prover), A/B of the reduction forms
# (base = row-wise reduce of round 2, scan = product scanning with pinned multiply-adds, pin1 = the same with the empty-asm pin that
# makes the compiler insert s_nop), NTT base vs scan, lane-length sweep with the smaller register footprint, bench, placed prove.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
python -c "import os, cosnarks_amd as h; print('devices', h.device_count(), h.lib().csh_version())" > $O/info.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail 20 -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
bash tools/experiments/gpu_ab.sh "0:0:20 0:0:22 0:0:24" base scan pin1 > $O/ab_g1.txt 2>&1; cp $O/ab.log $O/ab_g1.log; tail -9 $O/ab_g1.txt
bash tools/experiments/gpu_ab.sh "0:1:20 1:0:20 1:1:20" base scan > $O/ab_other.txt 2>&1; cp $O/ab.log $O/ab_other.log; tail -6 $O/ab_other.txt
for v in base scan; do echo "== $v"; COSNARKS_HIP_LIB=$R/gpurun_ab/libcosnarks_hip_$v.so NTT_LOGN=20,22,24 timeout 300 python tools/gpu_probe_ntt.py; done > $O/ntt_ab.log 2>&1; grep -E "==|ntt|NTT|ifft|fft" $O/ntt_ab.log | head -20
timeout 300 python tools/gpu_lsweep.py --ls 40,44,48,52,55,60,64,72,80,96 0:0:20 > $O/lsweep20.log 2>&1; tail -12 $O/lsweep20.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -c 1500 $O/bench.log
timeout 300 python tools/bench_prove_devices.py --devices 0 > $O/prove_devices.log 2>&1
timeout 300 python tools/bench_prove_devices.py --devices 0,0 >> $O/prove_devices.log 2>&1
timeout 300 python tools/bench_prove_devices.py --devices 0,0,0,0,0 >> $O/prove_devices.log 2>&1; cat $O/prove_devices.log
BENCH_FOLD_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --workload groth16_prove --log-n 18 > $O/bench_prove_fold2.log 2>&1; tail -c 900 $O/bench_prove_fold2.log
timeout 300 python bench.py --workload groth16_prove --log-n 20 --steps 5 --warmup 2 > $O/bench_prove_n1.log 2>&1; tail -c 900 $O/bench_prove_n1.log
