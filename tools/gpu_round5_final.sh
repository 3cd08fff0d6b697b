#!/bin/bash
# Round 5 final evidence run on the committed tree: parity suite, smoke, bench (N = 1, and the N = 2 code path folded onto one
# GPU), rocprofv3 kernel stats of the bench headline command, per-group kernel stats + SQ counters, FETCH_SIZE / WRITE_SIZE
# calibration and HBM counters of the MSM / NTT / share kernels. Summaries (CSV) are written next to the raw databases.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; S=$O/summaries; rm -rf $S; mkdir -p $S
python -c "import os, cosnarks_amd as h; print('devices', h.device_count(), h.lib().csh_version()); print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())" > $O/info.log 2>&1
(echo -n "cgroup cpu.max: "; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|^CPU\(s\)"; uptime) >> $O/info.log 2>&1
timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 600 --maxfail 20 -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -c 600 $O/bench.log   # the driver's own command
timeout -s KILL 900 python bench.py --no-extras --no-cpu-baseline > $O/bench_default_flags.log 2>&1   # 100 steps after 20
timeout -s KILL 300 python bench.py --workload groth16_prove --log-n 20 --steps 10 --warmup 3 > $O/bench_prove_n1.log 2>&1; tail -c 300 $O/bench_prove_n1.log
for m in "0 --mode 0" "0,0 --mode 1" "0,0,0,0 --mode 1" "0,0,0,0 --mode 2"; do timeout -s KILL 300 python tools/bench_prove_devices.py --devices $m; done > $O/prove_devices.log 2>&1
timeout -s KILL 300 python tools/gpu_msm_loop.py 0:0:16 0:0:18 0:0:20 0:0:22 0:0:24 0:1:20 0:1:22 1:0:20 1:0:22 1:0:24 1:1:20 1:1:22 2:0:20 > $O/msm_stages.log 2>&1
timeout -s KILL 300 python tools/gpu_probe.py > $O/probe.log 2>&1; head -1 $O/probe.log
NTT_LOGN=16,20,22,24 timeout -s KILL 300 python tools/gpu_probe_ntt.py > $O/probe_ntt.log 2>&1
timeout -s KILL 300 python tools/gpu_vecops_loop.py > $O/vecops_loop.log 2>&1
timeout -s KILL 300 python tools/msm_warm.py 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 0:0:19 0:0:20 0:0:22 0:1:20 1:0:20 1:1:20 > $O/msm_warm.log 2>&1
for LOGN in 12 16 18 20 22 24; do timeout -s KILL 200 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 6 --reps 10 default=0x0 plain=0x100900; done > $O/ntt_sizes.log 2>&1
timeout -s KILL 200 python tools/ntt_ab.py --logn 22 --ncomp 2 --rounds 6 --reps 6 default=0x0 plain=0x100900 >> $O/ntt_sizes.log 2>&1
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/prof_bench.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/prof_groups -o g -- python $R/tools/gpu_msm_loop.py 0:0:20 0:1:20 1:0:20 1:1:20 > $O/prof_groups.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc $SQ --kernel-trace -d $O/pmc_sq_groups -o g -- python $R/tools/gpu_msm_loop.py --reps 2 0:0:20 0:1:20 1:0:20 1:1:20 > $O/pmc_sq_groups.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/prof_g1_24 -o g -- python $R/tools/gpu_msm_loop.py --reps 3 0:0:24 > $O/prof_g1_24.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/prof_vecops -o v -- python $R/tools/gpu_vecops_loop.py > $O/prof_vecops.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc $SQ --kernel-trace -d $O/pmc_sq_vecops -o v -- python $R/tools/gpu_vecops_loop.py > $O/pmc_sq_vecops.log 2>&1
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $CNT --kernel-trace -d $O/pmc_${CNT}_calib -o c -- python $R/tools/gpu_calib.py > $O/pmc_${CNT}_calib.log 2>&1
  for J in 0:0:20 0:0:24 0:1:20 1:0:20 1:1:20; do
    timeout -s KILL 300 rocprofv3 --pmc $CNT --kernel-trace -d $O/pmc_${CNT}_msm_${J//:/_} -o m -- python $R/tools/gpu_msm_loop.py --reps 2 $J > $O/pmc_${CNT}_msm_${J//:/_}.log 2>&1
  done
  timeout -s KILL 300 rocprofv3 --pmc $CNT --kernel-trace -d $O/pmc_${CNT}_vec -o v -- python $R/tools/gpu_vecops_loop.py > $O/pmc_${CNT}_vec.log 2>&1
done
cd $R
db() { find $1 -name "*.db" | head -1; }
python tools/prof_summary.py $(db $O/prof_bench) $S/bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline (the headline command without its untimed extras)"
python tools/prof_summary.py $(db $O/prof_groups) $S/msm_groups_2p20_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_msm_loop.py 0:0:20 0:1:20 1:0:20 1:1:20 (5 MSMs each)"
python tools/prof_summary.py $(db $O/prof_g1_24) $S/msm_bn254g1_2p24_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_msm_loop.py --reps 3 0:0:24"
python tools/prof_summary.py $(db $O/prof_vecops) $S/vecops_ntt_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_vecops_loop.py (back-to-back launches)"
python tools/pmc_sq_summary.py $(db $O/pmc_sq_groups) $S/msm_groups_2p20_pmc_sq.csv "rocprofv3 --pmc SQ_* --kernel-trace -- python tools/gpu_msm_loop.py --reps 2 0:0:20 0:1:20 1:0:20 1:1:20"
python tools/pmc_sq_summary.py $(db $O/pmc_sq_vecops) $S/vecops_ntt_pmc_sq.csv "rocprofv3 --pmc SQ_* --kernel-trace -- python tools/gpu_vecops_loop.py"
python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_calib) $(db $O/pmc_WRITE_SIZE_calib) $S/calib_gather_pmc_hbm_bytes.csv "known-bytes launches of tools/gpu_calib.py: 2^24 lanes x REC bytes read, 2^24 x 4 written"
grep "^{" $O/pmc_FETCH_SIZE_calib.log > $S/calib_gather_known_bytes.jsonl
for J in 0_0_20 0_0_24 0_1_20 1_0_20 1_1_20; do
  python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_msm_$J) $(db $O/pmc_WRITE_SIZE_msm_$J) $S/msm_${J}_pmc_hbm_bytes.csv "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python tools/gpu_msm_loop.py --reps 2 ${J//_/:}; RAW counter bytes"
done
python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_vec) $(db $O/pmc_WRITE_SIZE_vec) $S/vecops_ntt_pmc_hbm_bytes.csv "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python tools/gpu_vecops_loop.py; RAW counter bytes"
for f in info.log pytest_gpu.log smoke.log bench.log bench_default_flags.log bench_prove_n1.log prove_devices.log msm_stages.log msm_warm.log ntt_sizes.log probe.log probe_ntt.log vecops_loop.log; do cp $O/$f $S/$f; done
ls $S | wc -l; du -sh $O
find $O -maxdepth 1 -type d \( -name "prof_*" -o -name "pmc_*" \) -exec rm -rf {} + 2>/dev/null   # the raw rocprof databases are not merged back (64 MiB cap)
