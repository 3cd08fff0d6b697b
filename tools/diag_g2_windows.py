import sys, os; sys.path.insert(0,".")
import numpy as np, ctypes as C
import cosnarks_amd as hip
from cosnarks_amd import bindings as B
from oracle import curves as cv, cbridge, fields as fl
from tests import helpers as H
L=hip.lib(); G=cv.BN254_G2; F=fl.BN254_FR
logn=int(os.environ.get("LOGN","15")); n=1<<logn; seed=5; c=int(os.environ["CSH_MSM_C"])
rs=np.random.RandomState(logn); limbs=rs.randint(0,1<<63,size=(n,4),dtype=np.uint64); limbs[:,3]>>=np.uint64(3)
buf=hip.DeviceBuffer(n*128); B._check(L.csh_util_generate_bases_dev(0,1,C.c_uint64(seed),C.c_size_t(n),buf.ptr,None)); B.sync()
pts=buf.to_host().reshape(n,16)
h=C.c_void_p(); B._check(L.csh_bases_upload_dev(0,1,buf.ptr,C.c_size_t(n),C.c_size_t(0),None,C.byref(h)))
sc=hip.DeviceBuffer.from_host(limbs)
pb=hip.msm_partial_bytes(0,1); part=hip.DeviceBuffer(pb)
B._check(L.csh_msm_partial_dev(h,C.c_size_t(0),C.c_size_t(n),sc.ptr,1,part.ptr,None))
raw=part.to_host(np.uint8)
hdr=raw[:16].view("<u4"); W=int(hdr[2]); assert int(hdr[1])==c
wins=raw[32:32+256*W].view(np.uint64).reshape(W,32)
# digits of canonical scalars
canon=[v*F.Rinv%F.p for v in (int.from_bytes(limbs[i].tobytes(),"little") for i in range(n))]
digs=np.zeros((n,W),dtype=np.int64)
for i,s in enumerate(canon):
    d=(C.c_int32*200)(); Wc=C.c_int(0)
    scl=H.pack(F,[s],mont=False)
    L.csh_selftest_digits(0,scl.ctypes.data_as(C.c_void_p),c,d,C.byref(Wc))
    digs[i,:]=list(d)[:W]
bad=[]
for w in range(W):
    mag=np.abs(digs[:,w]).astype(np.uint64)
    scal=np.zeros((n,4),dtype=np.uint64); scal[:,0]=mag
    P=pts.copy()
    neg=digs[:,w]<0
    # negate y (c0,c1) for negative digits: y -> q - y  (Montgomery form negation = modular negation)
    q=fl.BN254_Q
    for i in np.nonzero(neg)[0]:
        for off in (8,12):
            y=int.from_bytes(P[i,off:off+4].tobytes(),"little")
            if y: P[i,off:off+4]=np.frombuffer(((q-y)%q).to_bytes(32,"little"),dtype="<u8")
    want=cv.unpack_points(G, cbridge.msm(0,1,P,scal,montgomery=False))[0]
    o2=np.zeros(16,dtype=np.uint64)
    L.csh_selftest_curve_op(0,1,4,wins[w].ctypes.data_as(C.c_void_p),None,0,o2.ctypes.data_as(C.c_void_p))
    got=cv.unpack_points(G,o2)[0]
    if not G.eq(got,want): bad.append(w)
print("c",c,"logn",logn,"W",W,"bad windows",bad)
