#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out
for LOGN in 22 23 24; do timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 10 --reps 10 base=0x0 t10=0x100 t9=0x200; done > $O/r04_k_ntt_tiles.log 2>&1
timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 2 --rounds 8 --reps 6 base=0x0 t10=0x100 t9=0x200 >> $O/r04_k_ntt_tiles.log 2>&1
grep "false" $O/r04_k_ntt_tiles.log; grep "tune\"" $O/r04_k_ntt_tiles.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['ncomp'], d['variant'], 'ifft', d['ifft_ms_median'], 'fft', d['fft_ms_median'], 'alt', d['alternating_ms_per_transform_median'], d.get('paired_delta_vs_first_pct_median'), d.get('paired_delta_vs_first_pct_min_max'))
"
