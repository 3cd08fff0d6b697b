#!/usr/bin/env python3
"""Where does a communicator whose peer never arrives block? rank 0 of 2, nobody else calls. argv[1] = comm_abort (0 / 1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cosnarks_amd as hip
from cosnarks_amd import bindings as B
B.tune_set("comm_timeout_ms", int(os.environ.get("PROBE_TIMEOUT_MS", "2000")))
print("unique id ...", flush=True)
uid = B.comm_unique_id()
print("init_rank(2 ranks, rank 0) ...", flush=True)
t0 = time.perf_counter()
try:
    c = hip.Comm.init_rank(uid, 2, 0)
    print("came up?!", flush=True)
except Exception as e:
    print("error after %.2f s: %s" % (time.perf_counter() - t0, e), flush=True)
# argv[2] = 1: build a one-rank RCCL communicator right after the abandoned one (RCCL's own initialisation: 15-35 s per communicator on
# these boxes, which is why the GPU suite passes 0 and only checks that the library itself stays usable: a communicator that needs no RCCL)
t1 = time.perf_counter()
if len(sys.argv) > 2 and sys.argv[2] == "1":
    print("one-rank RCCL communicator afterwards ...", flush=True)
    c = hip.Comm.init_rank(B.comm_unique_id(), 1, 0)
else:
    print("one-rank communicator (no RCCL) afterwards ...", flush=True)
    c = hip.Comm.init_rank(None, 1, 0)
print("ok", c.info(), "after %.2f s" % (time.perf_counter() - t1), flush=True)
c.destroy()
print("done", flush=True)
os._exit(0)
