#!/usr/bin/env python3
"""Could the bracket scan take the SIGN of the Rayleigh secular function from an FP32 evaluation?  (CPU, numpy only.)
The same compound-matrix recursion (a vectorised float64 / float32 restatement of csrc/swd_common.h's, normalised per layer as
normc does) is evaluated in both precisions at every grid point of long scans (dc = 0.005 km/s from 0.8 x the slowest S velocity
up to just past the first sign change) of random 10-layer models, 10 periods each; printed: how far the FP32 value of the
normalised function is from the FP64 one, and what share of the scan's points lies further from zero than a threshold -- the
points whose sign an FP32 pass with an error bound of that size would settle.
    python tools/cpu_f32_sign.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayhunter_amd.synth import synth_models
def secular(c, omega, h, vp, vs, rho, dt):
    """vectorised over c (array); returns e (5,N) normalised per layer (divide by max abs), dtype dt"""
    T=dt
    c=c.astype(T); om=T(omega)
    k=om/c; k2=k*k
    def hs(a,b,r):
        xka=om/T(a); xkb=om/T(b)
        ra=np.sqrt((k+xka)*np.abs(k-xka)); rb=np.sqrt((k+xkb)*np.abs(k-xkb))
        t=T(b)/om; gammk=T(2)*t*t; gam=gammk*k2; gamm1=gam-T(1); r=T(r)
        return np.array([r*r*(gamm1*gamm1-gam*gammk*ra*rb), -r*ra, r*(gamm1-gammk*ra*rb), r*rb, k2-ra*rb])
    e=hs(vp[-1],vs[-1],rho[-1])
    e=e/np.abs(e).max(axis=0)
    for m in range(len(h)-2,-1,-1):
        a,b,r,d=T(vp[m]),T(vs[m]),T(rho[m]),T(h[m])
        xka=om/a; xkb=om/b
        ra=np.sqrt((k+xka)*np.abs(k-xka)); rb=np.sqrt((k+xkb)*np.abs(k-xkb))
        t=b/om; gammk=T(2)*t*t; gam=gammk*k2
        p=ra*d; q=rb*d
        pp=k<xka; qp=k<xkb
        facp=np.where(p<16,np.exp(-T(2)*np.minimum(p,T(16))),T(0)); facq=np.where(q<16,np.exp(-T(2)*np.minimum(q,T(16))),T(0))
        sinp=np.where(pp,np.sin(p),(T(1)-facp)*T(0.5)); cosp=np.where(pp,np.cos(p),(T(1)+facp)*T(0.5))
        sinq=np.where(qp,np.sin(q),(T(1)-facq)*T(0.5)); cosq=np.where(qp,np.cos(q),(T(1)+facq)*T(0.5))
        w=sinp/ra; x=np.where(pp,-ra*sinp,ra*sinp); y=sinq/rb; z=np.where(qp,-rb*sinq,rb*sinq)
        exa=np.where(pp,T(0),p)+np.where(qp,T(0),q)
        a0=np.where(exa<60,np.exp(-np.minimum(exa,T(60))),T(0))
        cpcq=cosp*cosq; cpy=cosp*y; cpz=cosp*z; cqw=cosq*w; cqx=cosq*x; xy=x*y; xz=x*z; wy=w*y; wz=w*z
        gamm1=gam-T(1); twgm1=gam+gamm1; gmgmk=gam*gammk; gmgm1=gam*gamm1; gm1sq=gamm1*gamm1; rho2=r*r; a0pq=a0-cpcq
        ca=[[None]*5 for _ in range(5)]
        ca[0][0]=cpcq-T(2)*gmgm1*a0pq-gmgmk*xz-k2*gm1sq*wy
        ca[0][1]=(k2*cpy-cqx)/r
        ca[0][2]=-(twgm1*a0pq+gammk*xz+k2*gamm1*wy)/r
        ca[0][3]=(cpz-k2*cqw)/r
        ca[0][4]=-(T(2)*k2*a0pq+xz+k2*k2*wy)/rho2
        ca[1][0]=(gmgmk*cpz-gm1sq*cqw)*r; ca[1][1]=cpcq; ca[1][2]=gammk*cpz-gamm1*cqw; ca[1][3]=-wz; ca[1][4]=ca[0][3]
        ca[3][0]=(gm1sq*cpy-gmgmk*cqx)*r; ca[3][1]=-xy; ca[3][2]=gamm1*cpy-gammk*cqx; ca[3][3]=ca[1][1]; ca[3][4]=ca[0][1]
        ca[4][0]=-(T(2)*gmgmk*gm1sq*a0pq+gmgmk*gmgmk*xz+gm1sq*gm1sq*wy)*rho2; ca[4][1]=ca[3][0]
        ca[4][2]=-(gammk*gamm1*twgm1*a0pq+gam*gammk*gammk*xz+gamm1*gm1sq*wy)*r; ca[4][3]=ca[1][0]; ca[4][4]=ca[0][0]
        tt=-T(2)*k2
        ca[2][0]=tt*ca[4][2]; ca[2][1]=tt*ca[3][2]; ca[2][2]=a0+T(2)*(cpcq-ca[0][0]); ca[2][3]=tt*ca[1][2]; ca[2][4]=tt*ca[0][2]
        en=[sum(e[i]*ca[i][j] for i in range(5)) for j in range(5)]
        e=np.array(en); e=e/np.abs(e).max(axis=0)
    return e
rs=np.random.RandomState(3)
nm=200
nlay,H,VP,VS,RHO=synth_models(rs,nm,10,lvz_frac=0.2)
per=np.linspace(2,60,30)
rel=[];mag=[]
for b in range(nm):
    n=nlay[b]; h,vp,vs,rho=[np.float64(np.float32(x[:n,b])) for x in (H,VP,VS,RHO)]
    for T in per[::3]:
        om=2*np.pi/T
        c=np.arange(0.8*vs.min(), vs[-1]*0.999, 0.005)
        e64=secular(c,om,h,vp,vs,rho,np.float64)
        e32=secular(c,om,h,vp,vs,rho,np.float32)
        f64=e64[0]; f32=e32[0].astype(np.float64)
        # the scan region: up to just past the first sign change
        sc=np.flatnonzero(np.sign(f64[1:])!=np.sign(f64[:-1]))
        end=sc[0]+2 if len(sc) else len(c)
        rel.append(np.abs(f32-f64)[:end]); mag.append(np.abs(f64)[:end])
rel=np.concatenate(rel); mag=np.concatenate(mag)
print('scan points',rel.size)
for q in (50,90,99,99.9,100): print('  |f32-f64| percentile %5.1f: %.2e'%(q,np.percentile(rel,q)))
for thr in (1e-2,1e-3,1e-4): print('  fraction of scan points with |f| > %g: %.4f ; of those, worst |f32-f64|: %.2e '%(thr,(mag>thr).mean(), rel[mag>thr].max()))
