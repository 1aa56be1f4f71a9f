import sys; sys.path.insert(0,'/root/repo')
import ctypes as C, numpy as np
import fps_amd
from fps_amd import _lib
lib=_lib.load()
for m,n in [(384,700),(400,800),(512,1030),(513,1030),(640,1300)]:
    rng=np.random.default_rng(m*1000+n)
    A=rng.uniform(-1,1,(m,n))/np.sqrt(n)
    d=C.c_void_p(); assert lib.fpsq_dense_create(C.byref(d),n,m,0)==0
    A=np.ascontiguousarray(A); lib.fpsq_dense_set_jacobian(d,A.ctypes.data)
    info=C.c_int32(); rc=lib.fpsq_dense_factorize(d,0.0,C.byref(info))
    L=np.empty((m,m)); lib.fpsq_dense_get_factor(d,L.ctypes.data)
    Lr=np.linalg.cholesky(A@A.T)
    Lt=np.tril(L)
    err=np.abs(Lt-Lr)
    blk=[[float(err[i*128:(i+1)*128,j*128:(j+1)*128].max()) if j<=i else 0 for j in range((m+127)//128)] for i in range((m+127)//128)]
    print(m,n,'rc',rc,'info',info.value,'maxerr',err.max()); 
    for r in blk: print('   ',['%.1e'%v for v in r])
    lib.fpsq_dense_destroy(d)
