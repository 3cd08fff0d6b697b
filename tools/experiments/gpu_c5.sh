export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for G in 1 2 3 4 6; do
  echo "pipeline groups $G"
  LOOP_NO_TIMING=1 CSH_MSM_PIPELINE=$G python tools/gpu_msm_loop.py --reps 8 0:0:18 0:0:20 0:0:22 0:0:24 0:1:20 1:0:20 1:1:20 2>&1 | cut -c1-200
done > $O/c5_pipeline.log 2>&1
for C in 15 16; do echo "c=$C groups 3"; LOOP_NO_TIMING=1 CSH_MSM_PIPELINE=3 CSH_MSM_C=$C python tools/gpu_msm_loop.py --reps 8 0:0:20 2>&1 | cut -c1-200; done >> $O/c5_pipeline.log 2>&1
cat $O/c5_pipeline.log
CSH_MSM_PIPELINE=3 timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py -m gpu -q -x -p no:cacheprovider > $O/pytest_c5.log 2>&1; grep -E "passed|failed|error" $O/pytest_c5.log | tail -3
