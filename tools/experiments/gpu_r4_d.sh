#!/bin/bash
# Round 4, run D: NTT tile size A/B, interleaved (tools/ntt_ab.py): 2^11-entry tiles (two per CU) vs 2^10-entry tiles (four per CU), radix-4 pass.
mkdir -p gpurun_out; O=$PWD/gpurun_out
for LOGN in 22 20 24; do
  timeout 300 python tools/ntt_ab.py --logn $LOGN --ncomp 1 --rounds 12 --reps 10 base=0x0 t10=0x100 r2=0x2 r2t10=0x102
done > $O/r04_d_ntt_tile_ab.log 2>&1
timeout 300 python tools/ntt_ab.py --logn 22 --ncomp 2 --rounds 8 --reps 6 base=0x0 t10=0x100 >> $O/r04_d_ntt_tile_ab.log 2>&1
cat $O/r04_d_ntt_tile_ab.log
