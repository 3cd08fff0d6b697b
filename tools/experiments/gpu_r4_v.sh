#!/bin/bash
# Round 4, run V: LDS bank swizzle of the NTT tiles: parity, interleaved A/B against the unswizzled layout, SQ counters of both.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
timeout -s KILL 600 python -m pytest tests/test_gpu_vec_ntt.py tests/test_gpu_plonk_vectors.py "tests/test_gpu_fullsize.py::test_ntt_every_size_up_to_2p19_vs_cpu_restatement" -m gpu -q -x -p no:cacheprovider --timeout 300 > $O/r04_v_pytest.log 2>&1; tail -3 $O/r04_v_pytest.log
for LOGN in 22 20 16; do timeout -s KILL 200 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 10 --reps 10 swz=0x0 noswz=0x800; done > $O/r04_v_ntt_swizzle_ab.log 2>&1
timeout -s KILL 200 python tools/ntt_ab.py --logn 22 --ncomp 2 --rounds 6 --reps 6 swz=0x0 noswz=0x800 >> $O/r04_v_ntt_swizzle_ab.log 2>&1
grep "tune\"" $O/r04_v_ntt_swizzle_ab.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['ncomp'], d['variant'], 'ifft', d['ifft_ms_median'], 'fft', d['fft_ms_median'], 'alt', d['alternating_ms_per_transform_median'], d.get('paired_delta_vs_first_pct_median'), d.get('paired_delta_vs_first_pct_min_max'))
"
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for V in 0x0 0x800; do
  timeout -s KILL 240 rocprofv3 --pmc $SQ --kernel-trace -d $O/pmc_sq_ntt_$V -o n -- python $R/tools/ntt_loop.py --pairs 3 --warm 2 --variant $V > $O/r04_v_pmc_sq_ntt_$V.log 2>&1
done
cd $R
db() { find $1 -name "*.db" | head -1; }
python tools/pmc_sq_summary.py $(db $O/pmc_sq_ntt_0x0) $O/r04_v_ntt_2p22_pmc_sq_swizzled.csv "rocprofv3 --pmc SQ_* --kernel-trace -- python tools/ntt_loop.py --pairs 3 --warm 2 (default: swizzled LDS tile, 2^10-element tiles at 2^22)"
python tools/pmc_sq_summary.py $(db $O/pmc_sq_ntt_0x800) $O/r04_v_ntt_2p22_pmc_sq_unswizzled.csv "rocprofv3 --pmc SQ_* --kernel-trace -- python tools/ntt_loop.py --pairs 3 --warm 2 --variant 0x800 (LDS swizzle off)"
cat $O/r04_v_ntt_2p22_pmc_sq_swizzled.csv $O/r04_v_ntt_2p22_pmc_sq_unswizzled.csv | cut -c1-400
rm -rf $O/pmc_sq_ntt_0x0 $O/pmc_sq_ntt_0x800
