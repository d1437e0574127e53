#!/usr/bin/env python3
"""Fit and error measurement of the one-chain exact GELU of jyutvoice_amd/csrc/jv_device.h (gelu_erf):
    gelu(v) = relu(v) - u 2^Q(u),  u = min(|v|, UMAX),  2^Q(u) = erfc(u / sqrt 2) / 2
Q = -1 + c1 u + ... + c_deg u^deg by iteratively re-weighted least squares (weight = u 2^Q ln 2, the sensitivity of the result
to an error in Q).  Prints, per degree, the fp32-emulated error against fp64 (absolute for v < 0, relative for v > 0.05,
absolute on [0, 0.05]) and the coefficients as fp32 literals.   usage: python tools/fit_gelu.py [UMAX]   (CPU only)"""
import numpy as np, math, sys
from scipy.special import erfc, erf
UMAX=float(sys.argv[1]) if len(sys.argv)>1 else 6.5
def target(u): return np.log2(0.5*erfc(u/np.sqrt(2.0)))
def fit(deg, umax=UMAX, iters=60):
    u=np.linspace(0,umax,40001)
    T=target(u)
    w=u*(0.5*erfc(u/np.sqrt(2)))*np.log(2)+1e-13
    A=np.stack([u**k for k in range(1,deg+1)],1)
    wt=w.copy()
    for it in range(iters):
        c,*_=np.linalg.lstsq(A*wt[:,None],(T+1)*wt,rcond=None)
        e=np.abs((A@c-(T+1))*w)
        wt=wt*(1+3*e/e.max())
    return np.concatenate([[-1.0],c])
f=np.float32
def eval32(c,v):
    v=v.astype(f); u=np.abs(v)
    uc=np.minimum(u,f(UMAX))
    q=np.full_like(u,f(c[-1]))
    for k in range(len(c)-2,-1,-1):
        q=(q.astype(np.float64)*uc.astype(np.float64)+np.float64(f(c[k]))).astype(f)
    E=np.exp2(q.astype(np.float64)).astype(f)
    t=(v+u).astype(f)
    w=(uc*E).astype(f)
    return (0.5*t.astype(np.float64)-w.astype(np.float64)).astype(f)
v=np.concatenate([np.linspace(-12,12,4800001), np.array([-1e4,-100,-30,30,100,1e4])])
ref=0.5*v*(1+erf(v/np.sqrt(2)))
def regions(name,g):
    err=np.abs(g-ref)
    neg=v<0; pos=v>0.05; mid=(v>=0)&(v<=0.05)
    print(f"{name:10s} neg abs {err[neg].max():.3e} (at {v[neg][err[neg].argmax()]:.3f})  pos rel {(err[pos]/ref[pos]).max():.3e}  mid abs {err[mid].max():.3e}")
for deg in (9,10,11,12):
    c=fit(deg); regions(f"deg{deg}",eval32(c,v).astype(np.float64))
    print("   ",[float(f(x)) for x in c])
