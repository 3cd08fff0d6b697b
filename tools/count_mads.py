#!/usr/bin/env python3
"""64-bit multiply-adds (v_mad_i64_i32) on the common path of one mixed addition of k_msm_accum, per group -> profiles/mads_per_madd.json
(read by tools/make_roofline_inputs.py; bench.py's `roofline.alu` is mixed additions x this count / the measured mad issue peak).

Derived from the formulas in csrc/curve_lazy.hpp::lazy_madd and csrc/field29.hpp (NL limbs; a product is NL^2 mads, a square
NL(NL+1)/2, a Montgomery reduction NL^2): EFD madd-2008-s = u2, s2, ppp, q, zz', zzz' (6 products), pp, r^2 (2 squares), y3 =
r (q - x3) - y1 ppp (two products fused under ONE reduction) -> 9 reductions. Over Fp2 a product is 4 base products + 2
reductions, a square is (a0^2 - a1^2: 2 squares) + (2 a0 a1: 1 product) + 2 reductions, and the fused y3 is 8 products + 2
reductions: four 2NL-term sums fit a 63-bit column on BLS12-381 (14 x 28 bits); on BN254 (9 x 29 bits) the accumulator takes a
carry sweep (FpS::compress_wide, no mads) between the first and the second pair of products.
x3 = r^2 - ppp - 2q takes its subtrahend into the reduction of r^2 (FpS::reduce_sub): NL more mads per base-field reduction.
With --isa FILE.s (hipcc -S --offload-device-only of csrc/msm_inst_<group>.hip) the static count of v_mad_i64_i32 in the kernel
body is printed next to it (it additionally contains the exact zero test behind the low-limb filter: 2 NL^2 per inlined copy)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fp_madd(nl):
    prod, sq, red = nl * nl, nl * (nl + 1) // 2, nl * nl
    return 8 * prod + 2 * sq + 9 * red + nl     # 6 products + fused pair, 2 squares, 9 reductions, x3's subtrahend (reduce_sub)


def fp2_madd(nl, four_fit):
    prod, sq, red = nl * nl, nl * (nl + 1) // 2, nl * nl
    mul2 = 4 * prod + 2 * red
    sqr2 = 2 * sq + prod + 2 * red
    y3 = 8 * prod + 2 * red                      # four_fit: one column holds all four sums; otherwise a carry sweep between the pairs
    return 6 * mul2 + 2 * sqr2 + y3 + 2 * nl    # + x3's subtrahend entering both components' reductions (reduce_sub)


out = {"Bn254G1": fp_madd(9), "GrumpkinG1": fp_madd(9), "Bls381G1": fp_madd(14), "Bn254G2": fp2_madd(9, False), "Bls381G2": fp2_madd(14, True)}
if len(sys.argv) > 2 and sys.argv[1] == "--isa":
    s = open(sys.argv[2]).read()
    for m in re.finditer(r"^(_ZN3csh11k_msm_accum\w+):.*?\n(.*?)s_endpgm", s, re.S | re.M):
        print("ISA static 64-bit mads (v_mad_i64_i32 + v_mad_u64_u32) in", m.group(1)[:48], "=", m.group(2).count("v_mad_i64_i32") + m.group(2).count("v_mad_u64_u32"))
json.dump(out, open(os.path.join(ROOT, "profiles", "mads_per_madd.json"), "w"), indent=1)
print(json.dumps(out))
