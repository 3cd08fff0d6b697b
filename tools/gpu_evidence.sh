#!/bin/bash
# The evidence set of a round, one runner: tools/gpu_evidence.sh <tag> [parts...]   (run through gpurun from the repo root)
#   parts (default: all): tests long bench prof pmc sq vecops warm prove
# Writes gpurun_out/<tag>_*; copy what DESIGN.md cites into profiles/.
TAG=${1:?tag}; shift
PARTS=${*:-tests long bench prof pmc sq vecops warm prove}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; S=$O/$TAG
has() { [[ " $PARTS " == *" $1 "* ]]; }
db() { find $1 -name "*.db" | head -1; }
prof() {  # prof <dir> <stats csv> <note> -- <command...>
  local d=$1 csv=$2 note=$3; shift 4
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $d -o p -- "$@" > $d.log 2>&1)
  python tools/prof_summary.py $(db $d) $csv "$note"
}
pmc() {   # pmc <counter list> <dir> -- <command...>   (counters in their own run, kernel trace only: gpurun refuses anything else)
  local ctr=$1 d=$2; shift 3
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d $d -o p -- "$@" > $d.log 2>&1)
}
if has tests; then (timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -30) > ${S}_pytest_gpu.log; fi
if has long; then (timeout 1500 python -m pytest tests -m gpu_long -x -q --durations=10 2>&1 | tail -20) > ${S}_pytest_gpu_long.log; fi
if has bench; then
  (timeout 1200 python bench.py --steps 20 --warmup 5 2>${S}_bench.err | tail -1) > ${S}_bench.log
  (timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > ${S}_bench_default_flags.log
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > ${S}_smoke.log
fi
if has prof; then
  prof $O/prof_bench ${S}_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline (the headline command without its untimed extras)" -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline
  prof $O/prof_tbl20 ${S}_tables_c17_2p20_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/msm_wide_probe.py --profile --c 17 --reps 40 --warm 20 0:0:20 (one plain call, the rest on the table handle)" -- python $R/tools/msm_wide_probe.py --profile --c 17 --reps 40 --warm 20 0:0:20
  python tools/prof_timeline.py $(db $O/prof_tbl20) ${S}_tables_c17_2p20_timeline.csv 16
  prof $O/prof_tbl24 ${S}_tables_c20_2p24_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/msm_wide_probe.py --profile --c 20 --reps 6 --warm 3 0:0:24" -- python $R/tools/msm_wide_probe.py --profile --c 20 --reps 6 --warm 3 0:0:24
  prof $O/prof_groups ${S}_msm_groups_2p20_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_msm_loop.py 0:0:20 0:1:20 1:0:20 1:1:20" -- python $R/tools/gpu_msm_loop.py 0:0:20 0:1:20 1:0:20 1:1:20
fi
if has vecops; then   # NTT / share-vector kernels AFTER the same spin-up the bench uses (VERDICT r5 #2b): 3 x avg pass must reproduce ntt_bn254_2p22.ms
  prof $O/prof_vecops ${S}_vecops_ntt_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/gpu_vecops_loop.py --spinup-s 0.5 (>= 0.5 s of untimed transforms first, then the loops)" -- python $R/tools/gpu_vecops_loop.py --spinup-s 0.5
  (timeout 300 python tools/gpu_vecops_loop.py --spinup-s 0.5 2>&1 | tail -12) > ${S}_vecops_loop.log
fi
if has pmc; then
  for job in 0:0:20 0:0:24 0:1:20 1:0:20; do
    j=${job//:/_}
    pmc FETCH_SIZE $O/pmc_f_$j -- python $R/tools/gpu_msm_loop.py --reps 3 $job
    pmc WRITE_SIZE $O/pmc_w_$j -- python $R/tools/gpu_msm_loop.py --reps 3 $job
    python tools/pmc_summary.py $(db $O/pmc_f_$j) $(db $O/pmc_w_$j) ${S}_msm_${j}_pmc_hbm_bytes.csv 2>/dev/null || true
  done
  pmc FETCH_SIZE $O/pmc_f_vec -- python $R/tools/gpu_vecops_loop.py --spinup-s 0
  pmc WRITE_SIZE $O/pmc_w_vec -- python $R/tools/gpu_vecops_loop.py --spinup-s 0
  python tools/pmc_summary.py $(db $O/pmc_f_vec) $(db $O/pmc_w_vec) ${S}_vecops_ntt_pmc_hbm_bytes.csv 2>/dev/null || true
  pmc FETCH_SIZE $O/pmc_f_cal -- python $R/tools/gpu_calib.py
  pmc WRITE_SIZE $O/pmc_w_cal -- python $R/tools/gpu_calib.py
  python tools/pmc_summary.py $(db $O/pmc_f_cal) $(db $O/pmc_w_cal) ${S}_calib_pmc_hbm_bytes.csv 2>/dev/null || true
fi
if has sq; then
  pmc "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" $O/sq_a -- python $R/tools/gpu_msm_loop.py --reps 3 0:0:20 0:1:20 1:0:20
  pmc "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VALU" $O/sq_b -- python $R/tools/gpu_msm_loop.py --reps 3 0:0:20 0:1:20 1:0:20
  python tools/pmc_sq_summary.py $(db $O/sq_a) $(db $O/sq_b) ${S}_msm_groups_2p20_pmc_sq.csv 2>/dev/null || true
fi
if has warm; then
  (timeout 600 python tools/msm_warm.py 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 0:0:19 0:0:20 2>&1 | tail -10) > ${S}_msm_warm.log
  (timeout 600 python tools/msm_wide_probe.py --c 17 --reps 40 --warm 30 0:0:14 0:0:15 0:0:16 0:0:17 0:0:18 0:0:19 0:0:20 2>&1 | tail -20) > ${S}_msm_warm_tables.log
  (timeout 300 python tools/gpu_probe.py 2>&1 | tail -12) > ${S}_probe.log
fi
if has prove; then (timeout 900 python tools/prove_probe.py --iters 11 --rep3 2>&1 | tail -8) > ${S}_prove_probe.log; fi
rm -rf $O/prof_* $O/pmc_* $O/sq_*
ls $O | grep "^$TAG" | head -60
