"""In-tree build of the gfx950 libraries (hipcc cross-compiles without a GPU).

    python co-snarks_amd/build.py            # incremental
    python co-snarks_amd/build.py --force

Outputs (git-ignored, but they travel to the GPU box with the snapshot):
    co-snarks_amd/lib/libcosnarks_hip.so        the C ABI of include/cosnarks_hip.h
    co-snarks_amd/lib/libcosnarks_groth16.so    host-side mirror of the reference's Groth16 interface
"""
from __future__ import annotations

import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP vectorizer finds nothing in straight-line 64-bit multiply-add chains and was a third of the compile time
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize"]

HIP_SOURCES = ["capi.hip", "vec_ops.hip", "ntt.hip", "msm.hip", "msm_inst_bn254g1.hip", "msm_inst_bn254g2.hip", "msm_inst_bls381g1.hip", "msm_inst_bls381g2.hip", "msm_inst_grumpking1.hip", "msm_inst_bls377g1.hip", "msm_inst_bls377g2.hip", "msm_accum_bn254g1.hip", "msm_accum_bn254g2.hip", "msm_accum_bls381g1.hip", "msm_accum_bls381g2.hip", "msm_accum_grumpking1.hip", "msm_accum_bls377g1.hip", "msm_accum_bls377g2.hip", "msm_sort.hip", "msm_sort_wide.hip", "msm_split.hip", "groth16_h.hip", "microbench.hip", "selftest.hip", "util.hip", "sparse.hip"]
COMPILE_TIMEOUT_S = 1500


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers(*dirs):
    out = []
    for d in dirs:
        if os.path.isdir(d):
            out += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hpp", ".inc", ".h"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "cosnarks_hip.h"))
    return out


def _run(cmd, timeout=COMPILE_TIMEOUT_S):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _compile(src_dir, name, force, hdrs):
    src = os.path.join(src_dir, name)
    obj = os.path.join(OBJDIR, os.path.splitext(name)[0] + ".o")
    if force or _newer(obj, [src] + hdrs):
        _run([HIPCC] + CFLAGS + ["-c", src, "-o", obj])
    return obj


def build_all(force: bool = False, verbose: bool = True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = _headers(CSRC, HOST)
    srcs = [s for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(CSRC, s, force, hdrs), srcs))
    lib = os.path.join(LIBDIR, "libcosnarks_hip.so")
    if force or _newer(lib, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    if verbose:
        print("built", lib)
    outs = [lib]
    # host-side mirror of the reference interface (C++ above the C ABI)
    host_srcs = [f for f in (os.listdir(HOST) if os.path.isdir(HOST) else []) if f.endswith(".cpp")]
    if host_srcs:
        hobjs = []
        for f in host_srcs:
            src = os.path.join(HOST, f)
            obj = os.path.join(OBJDIR, "host_" + os.path.splitext(f)[0] + ".o")
            if force or _newer(obj, [src] + hdrs):
                _run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-x", "hip", "--offload-arch=" + ARCH, "-Wno-unused-value", "-c", src, "-o", obj])
            hobjs.append(obj)
        hlib = os.path.join(LIBDIR, "libcosnarks_groth16.so")
        if force or _newer(hlib, hobjs + [lib]):
            _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", hlib] + hobjs +
                 ["-L" + LIBDIR, "-lcosnarks_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"])
        if verbose:
            print("built", hlib)
        outs.append(hlib)
    return outs


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
