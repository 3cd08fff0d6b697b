#!/bin/bash
# SQ issue / stall counters (one PMC pass, kernel trace only) for the MSM step at 2^20 and the NTT probe at 2^22.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
(cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_sq_msm -o msm -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras > $R/gpurun_out/pmc_sq_msm.log 2>&1)
(cd /tmp && NTT_LOGN=22 timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_sq_ntt -o ntt -- python $R/tools/gpu_probe_ntt.py > $R/gpurun_out/pmc_sq_ntt.log 2>&1)
ls gpurun_out/pmc_sq_msm gpurun_out/pmc_sq_ntt; tail -2 gpurun_out/pmc_sq_msm.log | cut -c1-200; tail -2 gpurun_out/pmc_sq_ntt.log | cut -c1-200
