#!/bin/bash
# Round 5, final tree: GPU parity suite, smoke, the bench line under the driver's own command, the N = 2 line folded onto one GPU
O=gpurun_out/r05_zz; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 600 --maxfail 20 -p no:cacheprovider --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; echo "bench exit $?" >> $O/bench.err
timeout -s KILL 300 python tools/msm_warm.py 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 0:0:19 0:0:20 1:0:16 1:0:17 1:0:18 > $O/msm_warm.log 2>&1
tail -4 $O/pytest_gpu.log; tail -1 $O/smoke.log; tail -c 400 $O/bench.log
