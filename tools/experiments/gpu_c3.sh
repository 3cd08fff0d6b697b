export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py -m gpu -q -x -p no:cacheprovider > $O/pytest_c3.log 2>&1; grep -E "passed|failed|error" $O/pytest_c3.log | tail -3
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o g -- python $R/tools/gpu_msm_loop.py 0:0:20 1:1:20 > $O/prof_c3.log 2>&1
cd $R; python tools/prof_summary.py $(find $O/prof_c3 -name "*.db" | head -1) $O/c3_kernel_stats.csv "quad tail"; cat $O/c3_kernel_stats.csv | grep -v gen_bases
