#!/bin/bash
# Round 4, run O: tile size x pass form sweep for the transform sizes real proving keys have (2^12 .. 2^21), interleaved.
mkdir -p gpurun_out; O=$PWD/gpurun_out
for LOGN in 12 14 15 16 17 18 19 20 21; do
  timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 8 --reps 20 base=0x0 r4=0x1 t10=0x100 t10r4=0x101 t9=0x200 t9r4=0x201 t8=0x300 t8r4=0x301
done > $O/r04_o_ntt_small.log 2>&1
for LOGN in 16 18 20; do
  timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 2 --rounds 6 --reps 20 base=0x0 r4=0x1 t10=0x100 t10r4=0x101 t9=0x200 t9r4=0x201 t8r4=0x301
done >> $O/r04_o_ntt_small.log 2>&1
grep "false" $O/r04_o_ntt_small.log
grep "tune\"" $O/r04_o_ntt_small.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['logn'], d['ncomp'], d['variant'], 'alt', d['alternating_ms_per_transform_median'], d.get('paired_delta_vs_first_pct_median'))
"
