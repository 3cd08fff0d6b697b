#!/bin/bash
# Round 4, run E: where does a 2^22 transform spend its time? One pass of the plan at a time (tune ntt_variant bits 12-13; wrong results on
# purpose): pass 1 = stages 0-10 on contiguous tiles, pass 2 = 6 stages (32-column runs), pass 3 = 5 stages (64-column runs).
mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 1 --rounds 8 --reps 10 base=0x0 p1=0x1000 p2=0x2000 p3=0x3000 p1t10=0x1100 p2t10=0x2100 p3t10=0x3100 > $O/r04_e_ntt_per_pass.log 2>&1
timeout 300 python tools/ntt_ab.py --logn 24 --ncomp 1 --rounds 6 --reps 6 base=0x0 p1=0x1000 p2=0x2000 p3=0x3000 >> $O/r04_e_ntt_per_pass.log 2>&1
grep -v equals $O/r04_e_ntt_per_pass.log | cut -c1-260
