#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out
for LOGN in 22 20 24; do timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 8 --reps 10 base=0x0 t10=0x100; done > $O/r04_h_ntt_alternating.log 2>&1
grep "tune\"" $O/r04_h_ntt_alternating.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['variant'], 'ifft', d['ifft_ms_median'], 'fft', d['fft_ms_median'], 'alternating', d['alternating_ms_per_transform_median'])
"
