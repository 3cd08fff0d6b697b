#!/bin/bash
# round 5, call d: full GPU suite on the current tree, the c sweep at 2^24 with stage times (why windows wider than 16 bits do not pay), bench
O=gpurun_out/r05_d; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
python - > $O/c_sweep_2p24.log 2>&1 <<'PY'
import ctypes as C, json, time, statistics
import numpy as np
import cosnarks_amd as hip
from cosnarks_amd import bindings as B
L = hip.lib()
for logn in (22, 24):
    n = 1 << logn
    buf = hip.DeviceBuffer(n * 64)
    B._check(L.csh_util_generate_bases_dev(0, 0, C.c_uint64(1), C.c_size_t(n), buf.ptr, None)); B.sync()
    h = C.c_void_p()
    B._check(L.csh_bases_upload_dev(0, 0, buf.ptr, C.c_size_t(n), C.c_size_t(0), None, C.byref(h))); buf.free()
    limbs = np.random.RandomState(1).randint(0, 1 << 63, size=(n, 4), dtype=np.uint64); limbs[:, 3] >>= np.uint64(3)
    sc = hip.DeviceBuffer.from_host(limbs)
    out = np.zeros(12, dtype=np.uint64)
    call = lambda: B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), sc.ptr, 1, out.ctypes.data_as(C.c_void_p), None))
    for _ in range(5): call()
    for c in (12, 13, 14, 15, 16):
        B.tune_set("msm_c", c); call()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
        B.tune_set("msm_timing", 1); call(); st = B.msm_last_timing(); pr = B.msm_last_params(); B.tune_set("msm_timing", 0)
        print(json.dumps({"log_n": logn, "c": c, "params_c_W_L_S": pr, "ms_median": round(statistics.median(ts), 3),
                          "stage_ms": {k: round(v, 3) for k, v in zip(("digits+hist", "scan", "scatter", "accum", "merge+reduce+fold", "total"), st)}}), flush=True)
    B.tune_set("msm_c", 0)
    L.csh_bases_free(h); sc.free()
PY
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.log 2> $O/bench_20_5.err; echo "bench exit $?" >> $O/bench_20_5.err
tail -25 $O/pytest_gpu.log; cat $O/c_sweep_2p24.log
