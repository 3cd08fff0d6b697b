#!/bin/bash
# sliced merge of oversized buckets: stage times on witness-like scalars (LOOP_SKEW) and uniform ones, kernel stats of the skewed run, parity
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for rep in 1 2; do
  LOOP_SKEW=1 timeout 600 python tools/gpu_msm_loop.py --reps 5 0:0:20 0:0:22 0:0:24 0:1:20 1:0:20 > $O/ab_skew_$rep.log 2>&1
  timeout 600 python tools/gpu_msm_loop.py --reps 5 0:0:20 0:0:24 1:1:20 > $O/ab_uni_$rep.log 2>&1
done
cat $O/ab_skew_1.log $O/ab_skew_2.log $O/ab_uni_1.log $O/ab_uni_2.log | cut -c1-230
cd /tmp; LOOP_SKEW=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_skew2 -o s -- python $R/tools/gpu_msm_loop.py --reps 5 0:0:20 > $O/prof_skew2.log 2>&1; cd $R
python tools/prof_summary.py $(find $O/prof_skew2 -name "*.db" | head -1) $O/skew2_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- LOOP_SKEW=1 python tools/gpu_msm_loop.py --reps 5 0:0:20 (sliced giant merge)"; head -12 $O/skew2_kernel_stats.csv | cut -c1-110
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py tests/test_gpu_msm_split.py tests/test_gpu_groth16.py tests/test_gpu_plonk_honk.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "not ntt" > $O/pytest_ab.log 2>&1
echo "pytest exit $?" >> $O/pytest_ab.log; tail -3 $O/pytest_ab.log
