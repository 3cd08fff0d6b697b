#!/usr/bin/env python3
"""Instruction histogram of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_hist.py FILE.s KERNEL_SUBSTRING [--blocks N] [--json OUT]

Prints the mnemonic histogram of the whole kernel body and of its N largest basic blocks (by instruction count; the
hot loop of the bucket kernels is a handful of straight-line blocks of 500-2000 instructions each), grouped into the
classes the cycle model of DESIGN.md prices: 64-bit multiply-adds, other VALU, SALU, memory, LDS, DPP moves, waits."""
import collections
import json
import re
import sys


def classify(m):
    if m.startswith("v_mad_i64_i32") or m.startswith("v_mad_u64_u32"):
        return "mad64"
    if m.startswith("v_"):
        return "valu"
    if m.startswith("s_waitcnt") or m.startswith("s_nop"):
        return "wait"
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "mem"
    return "other"


def kernel_body(text, sub):
    start = None
    lines = text.split("\n")
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\w+:", ln) and sub in ln:
            start = i
            break
    if start is None:
        raise SystemExit("kernel containing %r not found" % sub)
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith("\t.section") or re.match(r"^_Z\w+:", ln) or ln.startswith(".Lfunc_end"):
            break
        body.append(ln)
    return lines[start].split(":")[0], body


def blocks(body):
    out, cur, name = [], [], "entry"
    for ln in body:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\w+:", s):
                if cur:
                    out.append((name, cur))
                name, cur = s.split(":")[0], []
            continue
        mnem = s.split()[0]
        if s.endswith(":"):
            continue
        cur.append((mnem, s))
        if mnem.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            out.append((name, cur))
            name, cur = name + "+", []
    if cur:
        out.append((name, cur))
    return out


def hist(insts):
    h = collections.Counter(m for m, _ in insts)
    c = collections.Counter()
    for m, n in h.items():
        c[classify(m)] += n
    return h, c


def scratch_report(path, sub):
    """--scratch: where a kernel touches scratch (spills / stack traffic of out-of-line calls): per basic block with scratch
    instructions, its size and multiply-add count. Hot-loop blocks carry hundreds to thousands of multiply-adds; a block with
    scratch traffic and none is a call site / cold path."""
    text = open(path).read()
    name, body = kernel_body(text, sub)
    bl = blocks(body)
    hot = sorted(bl, key=lambda x: -sum(1 for m, _ in x[1] if classify(m) == "mad64"))[:4]
    print("kernel", name[:110])
    print("  %d blocks; largest multiply-add blocks: %s" % (len(bl), ", ".join("%s (%d instr, %d mad64, %d scratch)" % (
        b, len(i), sum(1 for m, _ in i if classify(m) == "mad64"), sum(1 for m, _ in i if m.startswith("scratch_"))) for b, i in hot)))
    any_sc = False
    for bname, insts in bl:
        sc = [s for m, s in insts if m.startswith("scratch_")]
        if sc:
            any_sc = True
            print("  scratch in block %-14s %5d instr, %4d mad64, %3d scratch ops (%d loads, %d stores)" % (
                bname, len(insts), sum(1 for m, _ in insts if classify(m) == "mad64"), len(sc), sum("load" in x for x in sc), sum("store" in x for x in sc)))
    if not any_sc:
        print("  no scratch instructions")


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    if "--scratch" in sys.argv:
        return scratch_report(sys.argv[1], sys.argv[2])
    text = open(sys.argv[1]).read()
    nblocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 6
    name, body = kernel_body(text, sys.argv[2])
    bl = blocks(body)
    allinst = [i for _, b in bl for i in b]
    h, c = hist(allinst)
    result = {"kernel": name, "total": len(allinst), "classes": dict(c), "mnemonics": dict(h.most_common()), "blocks": []}
    print("kernel", name[:100])
    print("  total %d instructions in %d blocks: %s" % (len(allinst), len(bl), dict(c)))
    for m, n in h.most_common(28):
        print("    %-28s %6d" % (m, n))
    for bname, insts in sorted(bl, key=lambda x: -len(x[1]))[:nblocks]:
        bh, bc = hist(insts)
        dpp = sum(1 for _, s in insts if "quad_perm" in s or "row_" in s or "dpp" in s.split()[0])
        print("  block %-14s %5d instructions: %s dpp=%d" % (bname, len(insts), dict(bc), dpp))
        print("     " + ", ".join("%s %d" % (m, n) for m, n in bh.most_common(14)))
        result["blocks"].append({"block": bname, "n": len(insts), "classes": dict(bc), "dpp": dpp, "mnemonics": dict(bh.most_common())})
    if "--json" in sys.argv:
        json.dump(result, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
