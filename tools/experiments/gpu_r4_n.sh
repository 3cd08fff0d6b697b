#!/bin/bash
# Round 4, run N: ablations of the radix-4 pass (wrong results on purpose), one pass of the 2^22 plan at a time: what is the per-sweep fixed cost made of?
mkdir -p gpurun_out; O=$PWD/gpurun_out
for P in 1 2 3; do
  timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 1 --rounds 6 --reps 10 full=0x${P}000 norounds=0x1${P}000 norounds_noload=0x3${P}000 norounds_nostore=0x5${P}000 norounds_nocanon=0x9${P}000 noio=0x6${P}000 onlylds=0x7${P}000
done > $O/r04_n_ntt_ablate.log 2>&1
grep "tune\"" $O/r04_n_ntt_ablate.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['tune']['ntt_variant'], d['variant'], 'ifft', d['ifft_ms_median'], 'fft', d['fft_ms_median'])
"
