import sys, os; sys.path.insert(0,".")
import numpy as np, ctypes as C
import cosnarks_amd as hip
from cosnarks_amd import bindings as B
from oracle import curves as cv, cbridge, fields as fl
from tests import helpers as H
CUR=int(os.environ.get("CUR","0")); GRP=int(os.environ.get("GRP","1"))
L=hip.lib(); G=cv.CURVES["bn254" if CUR==0 else "bls12_381"][GRP]; F=fl.BN254_FR if CUR==0 else fl.BLS381_FR
PW=hip.point_bytes(CUR,GRP)//8
logn=int(os.environ.get("LOGN","15")); n=int(os.environ.get("N", 1<<logn)); seed=5; c=None
rs=np.random.RandomState(logn); limbs=rs.randint(0,1<<63,size=(n,4),dtype=np.uint64); limbs[:,3]>>=np.uint64(3)
if os.environ.get("RANDPTS"):
    pl=H.rand_points(G,n,H.rng(1000+n+GRP))
    pts=cv.pack_points(G,pl).reshape(n,PW)
    buf=hip.DeviceBuffer.from_host(pts)
else:
    buf=hip.DeviceBuffer(n*PW*8); B._check(L.csh_util_generate_bases_dev(CUR,GRP,C.c_uint64(seed),C.c_size_t(n),buf.ptr,None)); B.sync()
    pts=buf.to_host().reshape(n,PW)
h=C.c_void_p(); B._check(L.csh_bases_upload_dev(CUR,GRP,buf.ptr,C.c_size_t(n),C.c_size_t(0),None,C.byref(h)))
sc=hip.DeviceBuffer.from_host(limbs)
pb=hip.msm_partial_bytes(CUR,GRP); part=hip.DeviceBuffer(pb)
B._check(L.csh_msm_partial_dev(h,C.c_size_t(0),C.c_size_t(n),sc.ptr,1,part.ptr,None))
raw=part.to_host(np.uint8)
hdr=raw[:16].view("<u4"); W=int(hdr[2]); c=int(hdr[1])
wins=raw[32:32+PW*16*W].view(np.uint64).reshape(W,2*PW)
# digits of canonical scalars
canon=[v*F.Rinv%F.p for v in (int.from_bytes(limbs[i].tobytes(),"little") for i in range(n))]
digs=np.zeros((n,W),dtype=np.int64)
for i,s in enumerate(canon):
    d=(C.c_int32*200)(); Wc=C.c_int(0)
    scl=H.pack(F,[s],mont=False)
    L.csh_selftest_digits(CUR,scl.ctypes.data_as(C.c_void_p),c,d,C.byref(Wc))
    digs[i,:]=list(d)[:W]
bad=[]
for w in range(W):
    mag=np.abs(digs[:,w]).astype(np.uint64)
    scal=np.zeros((n,4),dtype=np.uint64); scal[:,0]=mag
    P=pts.copy()
    neg=digs[:,w]<0
    # negate y (c0,c1) for negative digits: y -> q - y  (Montgomery form negation = modular negation)
    q=G.F.p; nl=(4 if CUR==0 else 6); ncf=PW//(2*nl)
    for i in np.nonzero(neg)[0]:
        for cf in range(ncf):
            off=ncf*nl+cf*nl
            y=int.from_bytes(P[i,off:off+nl].tobytes(),"little")
            if y: P[i,off:off+nl]=np.frombuffer(((q-y)%q).to_bytes(8*nl,"little"),dtype="<u8")
    want=cv.unpack_points(G, cbridge.msm(CUR,GRP,P,scal,montgomery=False))[0]
    o2=np.zeros(PW,dtype=np.uint64)
    L.csh_selftest_curve_op(CUR,GRP,4,wins[w].ctypes.data_as(C.c_void_p),None,0,o2.ctypes.data_as(C.c_void_p))
    got=cv.unpack_points(G,o2)[0]
    if not G.eq(got,want): bad.append(w)
print("c",c,"logn",logn,"W",W,"bad windows",bad)
