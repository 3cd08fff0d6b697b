import sys; sys.path.insert(0,'.')
import numpy as np, ctypes as C
import cosnarks_amd as hip
from oracle import curves as cv
from tests import helpers as H
for curve,group in [("bn254",0),("bn254",1),("bls12_381",0),("bls12_381",1)]:
    G=cv.CURVES[curve][group]
    pts=H.rand_points(G,64,H.rng(1))
    ap=cv.pack_points(G,pts).reshape(-1)
    bad=C.c_int(-1)
    rc=hip.lib().csh_selftest_lazy_chain_dev(H.CURVE_IDS[curve],group,ap.ctypes.data_as(C.c_void_p),C.c_size_t(64),C.c_size_t(200),C.c_size_t(65536),C.c_size_t(2048),C.byref(bad))
    print(curve,group,'rc',rc,'host-vs-device mismatches among 2048 sampled threads (200 madds each):',bad.value, flush=True)
