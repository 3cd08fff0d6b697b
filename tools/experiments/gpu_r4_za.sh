#!/bin/bash
# Round 4, run ZA: phase stagger of the NTT blocks that share a CU (ntt_variant bits 24-27 = us per wave slot), interleaved A/B.
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout -s KILL 240 python tools/ntt_ab.py --logn 22 --rounds 8 --reps 10 0 0x2000000 0x4000000 0x8000000 0xC000000 0x300 0x4000300 0x8000300 0x200 0x4000200 > $O/r04_za_stagger_2p22.log 2>&1
timeout -s KILL 120 python tools/ntt_ab.py --logn 24 --rounds 5 --reps 4 0 0x4000000 0x8000000 > $O/r04_za_stagger_2p24.log 2>&1
timeout -s KILL 120 python tools/ntt_ab.py --logn 20 --rounds 8 --reps 20 0 0x2000000 0x4000000 0x8000000 > $O/r04_za_stagger_2p20.log 2>&1
grep -h paired $O/r04_za_stagger_2p2*.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['logn'], r['variant'], r['ifft_ms_median'], r['fft_ms_median'], r['alternating_ms_per_transform_median'], r.get('paired_delta_vs_first_pct_median'))
"
grep -h "ifft_ms_median" $O/r04_za_stagger_2p2*.log | grep -v paired | cut -c1-300
grep -h "equals_first\|round_trip" $O/r04_za_stagger_2p2*.log | grep -c true
grep -h "equals_first\|round_trip" $O/r04_za_stagger_2p2*.log | grep -c false
