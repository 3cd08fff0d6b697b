export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_split.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "msm" > $O/pytest_c11.log 2>&1; grep -E "passed|failed|error" $O/pytest_c11.log | tail -3
python tools/gpu_msm_loop.py 0:0:16 0:0:18 0:0:20 0:0:22 0:0:24 1:0:20 2>&1 | cut -c1-200
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c11 -o g -- python $R/tools/gpu_msm_loop.py 0:0:20 > $O/prof_c11.log 2>&1
cd $R; python tools/prof_summary.py $(find $O/prof_c11 -name "*.db" | head -1) $O/c11_kernel_stats.csv "scan rewrite"; grep -E "scan|digits|hist|scatter|part_|colscan" $O/c11_kernel_stats.csv
