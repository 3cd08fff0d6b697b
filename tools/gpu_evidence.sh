#!/bin/bash
# The evidence set of a round, one parametrised runner (replaces tools/gpu_round*_final.sh and the per-experiment scripts):
#   tools/gpu_evidence.sh <tag> [parts...]        (through gpurun, from the repo root)
#   parts (default: all): tests long bench prof vecops pmc sq warm prove
# Writes gpurun_out/<tag>_*; copy what DESIGN.md cites into profiles/ and run tools/make_roofline_inputs.py <tag>.
TAG=${1:?tag}; shift
PARTS=${*:-tests long bench prof vecops pmc sq warm prove}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; S=$O/$TAG
has() { [[ " $PARTS " == *" $1 "* ]]; }
db() { find $1 -name "*.db" | head -1; }
prof() {  # prof <dir> <stats csv> <note> -- <command...>
  local d=$1 csv=$2 note=$3; shift 4
  (cd /tmp && timeout -s KILL 900 rocprofv3 --kernel-trace --stats -d $d -o p -- "$@" > $d.log 2>&1)
  python tools/prof_summary.py $(db $d) $csv "$note"
}
pmc() {   # pmc <dir> <counters...> -- <command...>   (counters in their own run with --kernel-trace only: gpurun refuses anything else)
  local d=$1; shift; local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  (cd /tmp && timeout -s KILL 600 rocprofv3 --pmc "${ctr[@]}" --kernel-trace -d $d -o p -- "$@" > $d.log 2>&1)
}
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
python -c "import os, cosnarks_amd as h; print('devices', h.device_count(), h.lib().csh_version()); print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())" > ${S}_info.log 2>&1
(echo -n "cgroup cpu.max: "; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|^CPU\(s\)"; uptime) >> ${S}_info.log 2>&1
if has tests; then (timeout -s KILL 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 2>&1 | tail -30) > ${S}_pytest_gpu.log; fi
if has long; then (timeout -s KILL 1500 python -m pytest tests -m gpu_long -q -p no:cacheprovider --durations=10 2>&1 | tail -25) > ${S}_pytest_gpu_long.log; fi
if has bench; then
  (timeout -s KILL 1200 python bench.py --steps 20 --warmup 5 2>${S}_bench.err | tail -1) > ${S}_bench.log          # the driver's own command
  (timeout -s KILL 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > ${S}_bench_default_flags.log
  (timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > ${S}_smoke.log
  (timeout -s KILL 300 python bench.py --workload groth16_prove --log-n 20 --steps 10 --warmup 3 2>/dev/null | tail -1) > ${S}_bench_prove_n1.log
fi
if has prof; then
  prof $O/prof_bench ${S}_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline (the headline command without its untimed extras)" -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline
  prof $O/prof_tbl20 ${S}_tables_c17_2p20_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/msm_wide_probe.py --profile --c 17 --reps 40 --warm 20 0:0:20 (one plain call, the rest on the table handle)" -- python $R/tools/msm_wide_probe.py --profile --c 17 --reps 40 --warm 20 0:0:20
  python tools/prof_timeline.py $(db $O/prof_tbl20) ${S}_tables_c17_2p20_timeline.csv 16
  prof $O/prof_tbl16 ${S}_tables_c17_2p16_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/msm_wide_probe.py --profile --c 17 --reps 40 --warm 20 0:0:16" -- python $R/tools/msm_wide_probe.py --profile --c 17 --reps 40 --warm 20 0:0:16
  python tools/prof_timeline.py $(db $O/prof_tbl16) ${S}_tables_c17_2p16_timeline.csv 16
  prof $O/prof_tbl24 ${S}_tables_c20_2p24_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/msm_wide_probe.py --profile --c 20 --reps 6 --warm 3 0:0:24" -- python $R/tools/msm_wide_probe.py --profile --c 20 --reps 6 --warm 3 0:0:24
  prof $O/prof_groups ${S}_msm_groups_2p20_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_msm_loop.py 0:0:20 0:1:20 1:0:20 1:1:20 (5 MSMs each)" -- python $R/tools/gpu_msm_loop.py 0:0:20 0:1:20 1:0:20 1:1:20
  prof $O/prof_g1_24 ${S}_msm_bn254g1_2p24_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_msm_loop.py --reps 3 0:0:24" -- python $R/tools/gpu_msm_loop.py --reps 3 0:0:24
fi
if has vecops; then   # NTT / share-vector kernels AFTER the same kind of spin-up the bench uses (VERDICT r5 #2b): 3 x avg pass must reproduce ntt_bn254_2p22.ms
  (timeout -s KILL 300 python tools/gpu_vecops_loop.py --spinup-s 0.5 2>&1 | tail -12) > ${S}_vecops_loop.log
  prof $O/prof_vecops ${S}_vecops_ntt_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_vecops_loop.py --spinup-s 0.5 (>= 0.5 s of untimed back-to-back work before each measured loop)" -- python $R/tools/gpu_vecops_loop.py --spinup-s 0.5
fi
if has sq; then
  pmc $O/pmc_sq_groups $SQ -- python $R/tools/gpu_msm_loop.py --reps 2 0:0:20 0:1:20 1:0:20 1:1:20
  python tools/pmc_sq_summary.py $(db $O/pmc_sq_groups) ${S}_msm_groups_2p20_pmc_sq.csv "rocprofv3 --pmc SQ_* --kernel-trace -- python tools/gpu_msm_loop.py --reps 2 0:0:20 0:1:20 1:0:20 1:1:20"
  pmc $O/pmc_sq_vecops $SQ -- python $R/tools/gpu_vecops_loop.py --spinup-s 0
  python tools/pmc_sq_summary.py $(db $O/pmc_sq_vecops) ${S}_vecops_ntt_pmc_sq.csv "rocprofv3 --pmc SQ_* --kernel-trace -- python tools/gpu_vecops_loop.py --spinup-s 0"
  pmc $O/pmc_sq_tbl $SQ -- python $R/tools/msm_wide_probe.py --profile --c 17 --reps 3 --warm 2 0:0:20
  python tools/pmc_sq_summary.py $(db $O/pmc_sq_tbl) ${S}_tables_c17_2p20_pmc_sq.csv "rocprofv3 --pmc SQ_* --kernel-trace -- python tools/msm_wide_probe.py --profile --c 17 --reps 3 --warm 2 0:0:20"
fi
if has pmc; then
  for CNT in FETCH_SIZE WRITE_SIZE; do
    pmc $O/pmc_${CNT}_calib $CNT -- python $R/tools/gpu_calib.py
    for J in 0:0:20 0:0:24 0:1:20 1:0:20 1:1:20; do pmc $O/pmc_${CNT}_msm_${J//:/_} $CNT -- python $R/tools/gpu_msm_loop.py --reps 2 $J; done
    pmc $O/pmc_${CNT}_vec $CNT -- python $R/tools/gpu_vecops_loop.py --spinup-s 0
    pmc $O/pmc_${CNT}_tbl20 $CNT -- python $R/tools/msm_wide_probe.py --profile --c 17 --reps 3 --warm 2 0:0:20
    pmc $O/pmc_${CNT}_tbl24 $CNT -- python $R/tools/msm_wide_probe.py --profile --c 20 --reps 2 --warm 1 0:0:24
  done
  python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_calib) $(db $O/pmc_WRITE_SIZE_calib) ${S}_calib_gather_pmc_hbm_bytes.csv "known-bytes launches of tools/gpu_calib.py: 2^24 lanes x REC bytes read, 2^24 x 4 written"
  grep "^{" $O/pmc_FETCH_SIZE_calib.log > ${S}_calib_gather_known_bytes.jsonl
  for J in 0_0_20 0_0_24 0_1_20 1_0_20 1_1_20; do
    python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_msm_$J) $(db $O/pmc_WRITE_SIZE_msm_$J) ${S}_msm_${J}_pmc_hbm_bytes.csv "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python tools/gpu_msm_loop.py --reps 2 ${J//_/:}; RAW counter bytes"
  done
  python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_vec) $(db $O/pmc_WRITE_SIZE_vec) ${S}_vecops_ntt_pmc_hbm_bytes.csv "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python tools/gpu_vecops_loop.py --spinup-s 0; RAW counter bytes"
  python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_tbl20) $(db $O/pmc_WRITE_SIZE_tbl20) ${S}_tables_c17_2p20_pmc_hbm_bytes.csv "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python tools/msm_wide_probe.py --profile --c 17 --reps 3 --warm 2 0:0:20; RAW counter bytes"
  python tools/pmc_summary.py $(db $O/pmc_FETCH_SIZE_tbl24) $(db $O/pmc_WRITE_SIZE_tbl24) ${S}_tables_c20_2p24_pmc_hbm_bytes.csv "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python tools/msm_wide_probe.py --profile --c 20 --reps 2 --warm 1 0:0:24; RAW counter bytes"
fi
if has warm; then
  (timeout -s KILL 600 python tools/msm_warm.py 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 0:0:19 0:0:20 0:0:22 0:1:20 1:0:20 1:1:20 2>&1 | tail -12) > ${S}_msm_warm.log
  (timeout -s KILL 600 python tools/msm_wide_probe.py --c 17 --reps 40 --warm 30 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 0:0:19 0:0:20 2>&1 | tail -16) > ${S}_msm_warm_tables.log
  timeout -s KILL 300 python tools/gpu_probe.py > ${S}_probe.log 2>&1
  NTT_LOGN=16,20,22,24 timeout -s KILL 300 python tools/gpu_probe_ntt.py > ${S}_probe_ntt.log 2>&1
fi
if has prove; then (timeout -s KILL 900 python tools/prove_probe.py --iters 11 --rep3 2>&1 | tail -8) > ${S}_prove_probe.log; fi
find $O -maxdepth 1 -type d \( -name "prof_*" -o -name "pmc_*" \) -exec rm -rf {} + 2>/dev/null   # raw rocprof databases are not merged back (64 MiB cap)
rm -f $O/prof_*.log $O/pmc_*.log
ls $O | grep "^$TAG" | wc -l
