#!/usr/bin/env python3
"""GPU probe: NTT / share-vector kernel throughput (HIP-event timed, data resident in HBM)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

L = hip.lib()


def timed(fn, reps=5):
    fn()
    B.sync()
    e0, e1 = B.Event(), B.Event()
    best = 1e9
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        best = min(best, e0.elapsed_ms(e1))
    return best


def snarkjs_root(logn):
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    z = pow(5, (r - 1) >> 28, r)
    g = pow(z, 1 << (28 - logn), r)
    R = (1 << 256) % r
    v = g * R % r
    return np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


rs = np.random.RandomState(1)
for logn in [int(x) for x in os.environ.get("NTT_LOGN", "16,20,22,24").split(",")]:
    n = 1 << logn
    for ncomp in (1, 2):
        host = rs.randint(0, 1 << 63, size=(n * ncomp, 4), dtype=np.uint64)
        host[:, 3] >>= np.uint64(3)
        buf = hip.DeviceBuffer.from_host(host)
        dom = hip.Domain(hip.BN254, logn, snarkjs_root(logn))
        t_i = timed(lambda: dom.ifft_in_to_out_dev(buf, ncomp))
        t_f = timed(lambda: dom.fft_out_to_in_dev(buf, ncomp))
        byts = 64.0 * n * ncomp
        print(json.dumps({"op": "ntt", "logn": logn, "ncomp": ncomp, "ifft_ms": round(t_i, 4), "fft_ms": round(t_f, 4),
                          "Melem_s": round(n * ncomp / t_f / 1e3, 1), "alg_GBs": round(byts / t_f / 1e6, 1)}), flush=True)
        buf.free()
        dom.free()

n = 1 << int(os.environ.get("VEC_LOGN", "24"))
a = hip.DeviceBuffer.from_host(rs.randint(0, 1 << 62, size=(2 * n, 4), dtype=np.uint64))
b = hip.DeviceBuffer.from_host(rs.randint(0, 1 << 62, size=(2 * n, 4), dtype=np.uint64))
m = hip.DeviceBuffer.from_host(rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64))
o = hip.DeviceBuffer(n * 64)
ops = {
    "rep3_local_mul_vec(192B)": (lambda: B._check(L.csh_rep3_local_mul_vec_dev(0, a.ptr, b.ptr, m.ptr, o.ptr, C.c_size_t(n), None)), 192),
    "vec_mul(96B)": (lambda: B._check(L.csh_vec_mul_dev(0, a.ptr, b.ptr, o.ptr, C.c_size_t(n), None)), 96),
    "vec_sub(96B)": (lambda: B._check(L.csh_vec_sub_dev(0, a.ptr, b.ptr, o.ptr, C.c_size_t(n), 1, None)), 96),
    "vec_mul_table_rep3(160B)": (lambda: B._check(L.csh_vec_mul_table_dev(0, a.ptr, m.ptr, C.c_size_t(n), 2, None)), 160),
}
for name, (fn, bpe) in ops.items():
    t = timed(fn)
    print(json.dumps({"op": name, "n": n, "ms": round(t, 4), "Gelem_s": round(n / t / 1e6, 2), "alg_GBs": round(n * bpe / t / 1e6, 1)}), flush=True)
