#!/bin/bash
# Round 4, run ZI: the parity suite, smoke and the bench line on the final tree.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 1200 python -m pytest tests -m gpu -q --timeout 600 --maxfail 20 -p no:cacheprovider --durations=5 > $O/r04_i_pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/r04_i_pytest_gpu.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_i_smoke.log 2>&1
timeout -s KILL 600 python bench.py > $O/r04_i_bench.log 2>&1
grep -E "passed|failed" $O/r04_i_pytest_gpu.log | tail -2; tail -1 $O/r04_i_smoke.log; tail -c 400 $O/r04_i_bench.log
