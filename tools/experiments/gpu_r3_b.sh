#!/bin/bash
# Round 3, second GPU call: parity suite (pinned accumulate / NTT units, unpinned tails, placement by query and by range, stream-
# ordered peer copies), A/B base vs scan2 on every group, the four-lane reduce on G2 now that it no longer spills, bench, prove.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail 20 -p no:cacheprovider --durations=5 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
bash tools/experiments/gpu_ab.sh "0:0:20 0:0:22 0:0:24 0:1:20 1:0:20 1:1:20" base scan2 > $O/ab2.txt 2>&1; cp $O/ab.log $O/ab2.log; tail -12 $O/ab2.txt
echo "== G2 reduce forms (scan2): default pair vs four-lane (msm_variant 1)" > $O/g2_forms.log
for v in 0 1; do echo "-- CSH_MSM_VARIANT=$v" >> $O/g2_forms.log; CSH_MSM_VARIANT=$v timeout 300 python tools/gpu_msm_loop.py --reps 10 0:1:20 1:1:20 0:1:22 >> $O/g2_forms.log 2>&1; done; cat $O/g2_forms.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -c 400 $O/bench.log
for m in "0 --mode 0" "0,0 --mode 1" "0,0,0,0 --mode 1" "0,0,0,0 --mode 2" "0,0,0,0,0,0,0,0 --mode 2"; do timeout 300 python tools/bench_prove_devices.py --devices $m; done > $O/prove_devices.log 2>&1; cat $O/prove_devices.log
BENCH_FOLD_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --workload groth16_prove --log-n 18 > $O/bench_prove_fold2.log 2>&1; tail -c 600 $O/bench_prove_fold2.log
