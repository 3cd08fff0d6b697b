#!/usr/bin/env python3
"""profiles/<tag>_* (rocprofv3 PMC summaries, known-bytes calibration, instruction-rate probe) -> profiles/roofline_inputs.json,
the file bench.py reads its roofline constants from (nothing about the roofline is a literal in bench.py).

    python tools/make_roofline_inputs.py r02_a

Read-side calibration: FETCH_SIZE tallies one 64-byte unit per memory-side request whether the request was 64 or 128 bytes
wide, so the factor raw -> bytes depends on the access pattern. It is taken from the known-bytes launches of
tools/gpu_calib.py profiled the same way (k_gather_calib<REC, SEQ>): hashed 64-byte records 1.00, 96-byte 1.00, 128-byte
2.00, 192-byte 1.50; coalesced streaming 2.00 (the guide's figure). WRITE_SIZE is exact on the calibration launches.
k_msm_accum gathers records of the group's affine point size at sorted-but-scattered indices -> the hashed-gather factor of
that record size; the share-vector kernels stream -> 2.00."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_a"


def rows(name):
    path = os.path.join(P, f"{tag}_{name}")
    if not os.path.exists(path):
        return []
    return list(csv.DictReader(l for l in open(path) if not l.startswith("#")))


out = {"generated_by": "tools/make_roofline_inputs.py " + tag, "hbm_peak_GBps": 8000.0, "kernels": {}, "fetch_factor": {}}
# 1. calibration factors
known = {}
kb = os.path.join(P, f"{tag}_calib_gather_known_bytes.jsonl")
if os.path.exists(kb):
    for l in open(kb):
        d = json.loads(l)
        known[d["kernel"]] = d
for r in rows("calib_gather_pmc_hbm_bytes.csv"):
    k = r["kernel"].replace("csh::", "")
    if k in known and float(r["fetch_bytes_raw"]) > 0:
        rec, seq = known[k]["rec_bytes"], known[k]["sequential"]
        out["fetch_factor"][f"{'stream' if seq else 'gather'}_{rec}"] = round(known[k]["bytes_read"] / float(r["fetch_bytes_raw"]), 3)
        out["fetch_factor"].setdefault("write", round(known[k]["bytes_written"] / max(1.0, float(r["write_bytes_raw"])), 3))
ff = out["fetch_factor"]
# 2. MSM accumulate kernels
groups = {"0_0": ("Bn254G1", 64), "0_1": ("Bn254G2", 128), "1_0": ("Bls381G1", 96), "1_1": ("Bls381G2", 192)}
for job in ("0_0_20", "0_0_24", "0_1_20", "1_0_20", "1_1_20"):
    name, rec = groups[job[:3]]
    for r in rows(f"msm_{job}_pmc_hbm_bytes.csv"):
        if r["kernel"].startswith(f"csh::k_msm_accum<csh::{name}Cfg>"):
            f = ff.get(f"gather_{rec}", 1.0)
            fb, wb = float(r["fetch_bytes_raw"]), float(r["write_bytes_raw"])
            out["kernels"][f"k_msm_accum<{name}> 2^{job[4:]}"] = {"traffic_bytes": int(fb * f + wb), "fetch_bytes_raw": int(fb), "fetch_factor": f, "write_bytes": int(wb),
                                                               "file": f"profiles/{tag}_msm_{job}_pmc_hbm_bytes.csv"}
# 3. share-vector / NTT kernels (coalesced streaming)
seen_ntt = []
for r in rows("vecops_ntt_pmc_hbm_bytes.csv"):
    fb, wb = float(r["fetch_bytes_raw"]), float(r["write_bytes_raw"])
    if "k_rep3_local_mul" in r["kernel"]:
        f = ff.get("stream_64", 2.0)
        out["kernels"]["k_rep3_local_mul 2^24"] = {"traffic_bytes": int(fb * f + wb), "fetch_bytes_raw": int(fb), "fetch_factor": f, "write_bytes": int(wb),
                                                   "file": f"profiles/{tag}_vecops_ntt_pmc_hbm_bytes.csv"}
    if "k_ntt_pass_lazy" in r["kernel"] or "k_ntt_pass_r4" in r["kernel"]:
        seen_ntt.append((fb, wb, int(r["launches"])))
if seen_ntt:
    # one transform = 3 passes; the loop runs inverse and forward transforms alternately: average per transform over both kinds
    tot_f = sum(f * n for f, _, n in seen_ntt)
    tot_w = sum(w * n for _, w, n in seen_ntt)
    passes = sum(n for _, _, n in seen_ntt)
    per_tr_f, per_tr_w = tot_f / passes * 3, tot_w / passes * 3
    f = ff.get("stream_64", 2.0)
    out["kernels"]["k_ntt_pass_lazy 2^22"] = {"traffic_bytes": int(per_tr_f * f + per_tr_w), "fetch_bytes_raw": int(per_tr_f), "fetch_factor": f, "write_bytes": int(per_tr_w),
                                              "note": "3 passes of one 2^22 transform; read factor = the streaming calibration (tile rows are >= 256 contiguous bytes)",
                                              "file": f"profiles/{tag}_vecops_ntt_pmc_hbm_bytes.csv"}
# 4. instruction-rate probe of the same tree
probe = os.path.join(P, f"{tag}_probe.log")
if os.path.exists(probe):
    first = json.loads(open(probe).readline())
    out["mad_peak_T"] = round(max(first["v_mad_i64_i32"], first["v_mad_u64_u32"]) / 1e3, 2)
    out["mad_peak_source"] = f"profiles/{tag}_probe.log (csh_microbench: forced v_mad_i64_i32 / v_mad_u64_u32 chains, lane-ops/s)"
    # the better of the two 9 x 29-bit multipliers on this tree: the unsigned row-wise one (k_modmul29) and the signed product-
    # scanning one with its multiply-add order pinned (k_modmul29s: what the NTT butterflies run since round 3)
    cands = {k: first[k] for k in ("modmul_bn254_fq_29x9_lazy", "modmul_bn254_fq_29x9_signed_scan") if k in first}
    best = max(cands, key=cands.get)
    out["modmul_peak_G"] = cands[best]
    out["modmul_peak_source"] = f"profiles/{tag}_probe.log ({best}: dependent 9x29-bit lazy Montgomery products, all CUs; candidates {cands})"
# 5. ISA counts of the accumulate loop bodies (64-bit multiply-adds per mixed addition; tools/count_mads.py)
cm = os.path.join(P, "mads_per_madd.json")
out["mads_per_madd"] = json.load(open(cm)) if os.path.exists(cm) else {"Bn254G1": 1467}
json.dump(out, open(os.path.join(P, "roofline_inputs.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
