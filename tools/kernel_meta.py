#!/usr/bin/env python3
"""Code-object metadata of the kernels in the in-tree build: VGPRs, spills, scratch bytes, LDS per kernel.

    python tools/kernel_meta.py [substring ...] [--csv OUT.csv]

Reads co-snarks_amd/build/*.o (the objects libcosnarks_hip.so is linked from): dumps each object's .hip_fatbin, unbundles the
gfx950 code object and parses the amdhsa.kernels notes with llvm-readelf. Kernel names are demangled down to the part after
`csh::`. With substrings, only kernels whose name contains one of them are printed."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "k.co")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return []
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], capture_output=True, text=True)
    if r.returncode != 0:
        return []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        def f(key, blk=blk):
            m = re.search(r"\.%s:\s*(\S+)" % key, blk)
            return m.group(1) if m else ""
        name = f("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void ", "", dem)
        dem = re.sub(r"\(.*$", "", dem).replace("csh::", "")
        out.append({"kernel": dem, "vgpr": int(f("vgpr_count") or 0), "vgpr_spill": int(f("vgpr_spill_count") or 0), "sgpr": int(f("sgpr_count") or 0),
                    "sgpr_spill": int(f("sgpr_spill_count") or 0), "scratch_bytes": int(f("private_segment_fixed_size") or 0),
                    "lds_bytes": int(f("group_segment_fixed_size") or 0), "max_flat_workgroup_size": int(f("max_flat_workgroup_size") or 0),
                    "object": os.path.basename(obj)})
    for p in (fat, co):
        if os.path.exists(p):
            os.remove(p)
    return out


def main():
    args = [a for a in sys.argv[1:]]
    csv = None
    if "--csv" in args:
        i = args.index("--csv")
        csv = args[i + 1]
        del args[i:i + 2]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, "co-snarks_amd", "build", "*.o"))):
            if os.path.basename(obj).startswith("host_"):
                continue
            rows += kernels_of(obj, tmp)
    if args:
        rows = [r for r in rows if any(a in r["kernel"] for a in args)]
    rows.sort(key=lambda r: (r["object"], r["kernel"]))
    cols = ["kernel", "vgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes", "lds_bytes", "max_flat_workgroup_size", "object"]
    lines = [",".join(cols)] + [",".join('"%s"' % r[c] if c == "kernel" else str(r[c]) for c in cols) for r in rows]
    if csv:
        open(csv, "w").write("# code-object metadata (llvm-readelf --notes of the gfx950 code objects in co-snarks_amd/build/*.o), tools/kernel_meta.py\n" + "\n".join(lines) + "\n")
    for r in rows:
        print("%-88s vgpr %3d spill %3d sgpr_spill %3d scratch %5d B  lds %6d B" % (r["kernel"][:88], r["vgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_bytes"], r["lds_bytes"]))


if __name__ == "__main__":
    main()
