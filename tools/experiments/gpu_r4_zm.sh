#!/bin/bash
# Round 4, run ZM: parity suites touched by the occupancy-aware lane length + the warm small-size MSM figures.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 800 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py tests/test_gpu_plonk_honk.py tests/test_gpu_plonk_vectors.py tests/test_gpu_trait_path.py tests/test_gpu_msm_split.py -q -m gpu -p no:cacheprovider --maxfail 5 > $O/r04_zm_pytest.log 2>&1
echo "pytest exit $?" >> $O/r04_zm_pytest.log
timeout -s KILL 200 python tools/msm_warm.py 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 0:0:20 1:0:16 2:0:16 > $O/r04_zm_msm_warm.log 2>&1
grep -E "passed|failed|exit" $O/r04_zm_pytest.log | tail -3; grep -v amdgpu.ids $O/r04_zm_msm_warm.log | cut -c1-170
