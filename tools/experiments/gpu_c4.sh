export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
for V in 0 1; do for SB in 8 4 2; do
  echo "variant $V seg_buckets $SB"
  CSH_MSM_VARIANT=$V CSH_MSM_SEG_BUCKETS=$SB python tools/gpu_msm_loop.py 0:0:20 0:1:20 1:0:20 1:1:20 0:0:24 1:1:22 2>&1 | cut -c1-175
done; done > $O/c4_sweep.log 2>&1
cat $O/c4_sweep.log
