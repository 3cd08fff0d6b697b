#!/usr/bin/env python3
"""Thread scaling of the oracle/c CPU MSM port on this host (context for bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import cbridge as cb

logn = int(os.environ.get("LOGN", "20"))
n = 1 << logn
pts = cb.generate_bases(0, 0, 0xBA5E, n)
rs = np.random.RandomState(1)
sc = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.uint64)
sc[:, 3] >>= np.uint64(3)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for t in [1, 8, 16, 32, 64, 128, 256]:
    if t > (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    cb.msm(0, 0, pts, sc, True, threads=t)
    dt = time.perf_counter() - t0
    print(f"threads {t:4d}: {dt*1e3:9.1f} ms  {n/dt/1e6:7.2f} Mpts/s", flush=True)
