"""Winograd F(4x4,3x3) against F(2x2,3x3) in emulated fp32 (transforms, products and the k-pair accumulation of the MFMA loop in
float32; weights transformed in double as the engine packs them), relative to a float64 direct convolution, on post-ReLU-like
activations and He-normal filters.  CPU only; DESIGN.md section 10 quotes its output."""
import numpy as np
F32=np.float32
def direct(x,w):
    # x (K,H,W) w (M,K,3,3) -> (M,H,W) pad 1, float64
    K,H,W=x.shape; M=w.shape[0]
    xp=np.zeros((K,H+2,W+2),x.dtype); xp[:,1:-1,1:-1]=x
    out=np.zeros((M,H,W),x.dtype)
    for ky in range(3):
        for kx in range(3):
            out+=np.einsum('mk,khw->mhw',w[:,:,ky,kx],xp[:,ky:ky+H,kx:kx+W])
    return out
def wino(x,w,m,pts_scale=None):
    # F(m x m, 3x3) in fp32: transforms in fp32, elementwise GEMM over K in fp32
    if m==2:
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float64)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float64)
    t=m+2
    K,H,W=x.shape; M=w.shape[0]
    U=np.einsum('ij,mkjl,nl->mkin',G,w.astype(np.float64),G).astype(F32)   # weights transformed in double, rounded
    xp=np.zeros((K,H+2,W+2),F32); xp[:,1:-1,1:-1]=x
    out=np.zeros((M,H,W),F32)
    BT32=BT.astype(F32); AT32=AT.astype(F32)
    for ty in range(0,H,m):
        for tx in range(0,W,m):
            d=xp[:,ty:ty+t,tx:tx+t]
            V=np.einsum('ij,kjl->kil',BT32,d).astype(F32)
            V=np.einsum('kil,nl->kin',V,BT32).astype(F32)
            # accumulate over K sequentially-ish in fp32: use float32 matmul (pairwise) - emulate with cumulative float32 sum in chunks of 2 (mfma k=2)
            Mm=np.zeros((M,t,t),F32)
            for k0 in range(0,K,2):
                Mm+= (U[:,k0]*V[k0][None]+U[:,k0+1]*V[k0+1][None]).astype(F32)
            Y=np.einsum('ij,mjl->mil',AT32,Mm).astype(F32)
            Y=np.einsum('mil,nl->min',Y,AT32).astype(F32)
            out[:,ty:ty+m,tx:tx+m]=Y
    return out
rng=np.random.RandomState(0)
for K in (64,256,512):
    M=32; H=W=12
    x=np.maximum(rng.randn(K,H,W),0).astype(F32)*F32(1.0)   # post-ReLU-like activations
    w=(rng.randn(M,K,3,3)*np.sqrt(2/(9*K))).astype(F32)
    ref=direct(x.astype(np.float64),w.astype(np.float64))
    d32=direct(x,w)
    for m in (2,4):
        y=wino(x,w,m)
        print('K',K,'F(%d,3)'%m,'rel-L2 %.2e'%(np.linalg.norm(y-ref)/np.linalg.norm(ref)), ' direct fp32 %.2e'%(np.linalg.norm(d32-ref)/np.linalg.norm(ref)))
