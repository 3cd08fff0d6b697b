#!/usr/bin/env python3
"""One thread driving every GPU of the node: `csh_msm_split` with its three exchanges (hipMemcpyPeer to a root device, one
device-to-host copy per GPU, grouped RCCL all-gather over communicators of csh_comm_init_all), ms per MSM each. Run by
bench.py (rank 0, N > 1) in a subprocess with a timeout, or by hand:

    python tools/bench_single_process_split.py --devices 8 [--curve 0 --group 0 --log-n 24]

Prints one JSON line. No torch: device memory through the C ABI."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cosnarks_amd as hip
from cosnarks_amd import bindings as B

SEED = 0x00C0FFEE5EED


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, required=True)
    ap.add_argument("--curve", type=int, default=0)
    ap.add_argument("--group", type=int, default=0)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--fold", action="store_true", help="every range on device 0 (the N > 1 code path on a one-GPU box; the grouped RCCL "
                                                        "exchange cannot put two communicator ranks on one device and is reported as skipped)")
    a = ap.parse_args()
    L = hip.lib()
    k = a.devices
    dev_of = (lambda d: 0) if a.fold else (lambda d: d)
    total = 1 << a.log_n
    per = total // k
    pb = hip.point_bytes(a.curve, a.group)
    bases, scal = [], []
    rs = np.random.RandomState(7)
    for d in range(k):
        B._check(L.csh_init(dev_of(d)))
        buf = hip.DeviceBuffer(per * pb)
        B._check(L.csh_util_generate_bases_dev(a.curve, a.group, C.c_uint64(SEED + d * per), C.c_size_t(per), buf.ptr, None))
        B.sync()
        h = C.c_void_p()
        B._check(L.csh_bases_upload_dev(a.curve, a.group, buf.ptr, C.c_size_t(per), C.c_size_t(0), None, C.byref(h)))
        buf.free()
        limbs = rs.randint(0, 1 << 61, size=(per, 4), dtype=np.uint64)
        scal.append(hip.DeviceBuffer.from_host(limbs))
        bases.append(h)
    B._check(L.csh_init(0))
    hs = (C.c_void_p * k)(*[h.value for h in bases])
    offs = (C.c_size_t * k)(*([0] * k))
    cnts = (C.c_size_t * k)(*([per] * k))
    ptrs = (C.c_void_p * k)(*[s.ptr.value for s in scal])
    res, outs, comms = {"devices": k, "points": total, "folded_on_device_0": bool(a.fold)}, {}, None
    for name, mode in (("hipMemcpyPeer", B.SPLIT_PEER), ("host_copies", B.SPLIT_HOST), ("rccl_grouped", B.SPLIT_RCCL)):
        cm = None
        if a.fold and mode == B.SPLIT_RCCL:
            res[name + "_skipped"] = "folded: one device cannot hold two ranks of a communicator"
            continue
        try:
            if mode == B.SPLIT_RCCL:
                comms = B.Comm.init_all(list(range(k)))
                cm = (C.c_void_p * k)(*[c.h.value for c in comms])
            o = np.zeros(3 * pb // 16, dtype=np.uint64)
            run = lambda: B._check(L.csh_msm_split(hs, offs, cnts, ptrs, C.c_size_t(k), 1, mode, cm, o.ctypes.data_as(C.c_void_p)))
            run()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                run()
            res[name + "_ms"] = round((time.perf_counter() - t0) / a.reps * 1e3, 3)
            outs[name] = o.copy()
        except Exception as e:  # noqa: BLE001
            res[name + "_error"] = repr(e)
    if outs:
        first = next(iter(outs.values()))
        res["exchanges_agree"] = bool(all((v == first).all() for v in outs.values()))
        res["points_per_s_best"] = round(total / (min(v for kk, v in res.items() if kk.endswith("_ms")) * 1e-3))
    if comms:
        for c in comms:
            c.destroy()
    for d, h in enumerate(bases):
        B._check(L.csh_init(dev_of(d)))
        L.csh_bases_free(h)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
