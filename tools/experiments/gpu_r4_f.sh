#!/bin/bash
# Round 4, run F: persistent radix-4 pass with the next tile's loads under the store tail (ntt_variant bit 10) against the default, interleaved.
mkdir -p gpurun_out; O=$PWD/gpurun_out
for LOGN in 22 24 20 21 23; do
  timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 10 --reps 10 base=0x0 pers=0x400 pers_t10=0x500 t10=0x100
done > $O/r04_f_ntt_persistent_ab.log 2>&1
timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 2 --rounds 8 --reps 6 base=0x0 pers=0x400 pers_t10=0x500 >> $O/r04_f_ntt_persistent_ab.log 2>&1
timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 1 --rounds 8 --reps 10 base=0x0 p1=0x1400 p2=0x2400 p3=0x3400 >> $O/r04_f_ntt_persistent_ab.log 2>&1
grep -v "tune\"" $O/r04_f_ntt_persistent_ab.log; grep "tune\"" $O/r04_f_ntt_persistent_ab.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['ncomp'], d['variant'], d['ifft_ms_median'], d['fft_ms_median'], d.get('paired_delta_vs_first_pct_median'), d.get('paired_delta_vs_first_pct_min_max'))
"
