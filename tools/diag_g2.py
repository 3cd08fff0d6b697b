import sys, os; sys.path.insert(0,".")
import numpy as np, ctypes as C
import cosnarks_amd as hip
from cosnarks_amd import bindings as B
from oracle import curves as cv
from tests import helpers as H
from tests.check_closed_form import closed_form_point
L=hip.lib(); G=cv.BN254_G2
logn=15; n=1<<logn; seed=5
rs=np.random.RandomState(logn); limbs=rs.randint(0,1<<63,size=(n,4),dtype=np.uint64); limbs[:,3]>>=np.uint64(3)
want=closed_form_point("bn254",1,seed,n,limbs,True)
res=[]
for up in range(2):
    buf=hip.DeviceBuffer(n*128); B._check(L.csh_util_generate_bases_dev(0,1,C.c_uint64(seed),C.c_size_t(n),buf.ptr,None)); B.sync()
    host_pts=buf.to_host()
    h=C.c_void_p(); B._check(L.csh_bases_upload_dev(0,1,buf.ptr,C.c_size_t(n),C.c_size_t(0),None,C.byref(h)))
    for rep in range(3):
        out=np.zeros(24,dtype=np.uint64)
        B._check(L.csh_msm(h,C.c_size_t(0),C.c_size_t(n),limbs.ctypes.data_as(C.c_void_p),1,out.ctypes.data_as(C.c_void_p)))
        res.append((bool(G.eq(H.jac_to_affine(G,out),want)), hex(int(out[0]))[-6:]))
    res.append(hex(int(host_pts.sum() & np.uint64(0xffffff))))
print(os.environ.get("CSH_ACC_VARIANT"), os.environ.get("CSH_ACC_BLK"), os.environ.get("CSH_MSM_C"), res)
