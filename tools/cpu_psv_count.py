#!/usr/bin/env python3
"""A mode count for the P-SV (Rayleigh) secular function -- what DESIGN 7d leaves open, explored on the CPU (numpy only, no GPU,
no oracle; the compound-matrix recursion below is a float64 restatement of csrc/swd_common.h's for this experiment).

Love targets skip scan steps under an exact Sturm count.  For Rayleigh the counterpart would be the number N(c) of
roots of the secular function below c.  This script establishes, numerically, what that count IS in terms of the five minors
e1..e5 the recursion already carries (bottom-up; e1 at the surface is the secular function, e5 is det U of the two solutions'
displacements):
    N(c) = Z(c) + neg(c)
    Z(c)   = number of zeros of e5 along depth (every layer cut into thin slices here, so that sign changes show them all)
    neg(c) = number of negative eigenvalues of the surface impedance, from the signs at the surface:
             e1 e5 > 0 -> 1;   else e2 e5 < 0 -> 0,  e2 e5 > 0 -> 2
(Z(c) is the mode count of the same stack with a RIGID surface; its zeros enter at the surface and move down as c grows.)
    python tools/cpu_psv_count.py check      the identity on random ragged models with low-velocity zones, brute-force root count
    python tools/cpu_psv_count.py interior   how many zeros of e5 the signs at the layer INTERFACES miss (two or more inside one
                                             layer), and in which layers: what an exact count without slicing has to supply"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayhunter_amd.synth import synth_models


def layer_ca(k2, k, omega, a, b, rho, d):
    """5x5 compound matrix of one layer (surfdisp96.f dnka), unscaled variant allowed: use the reference's scaling a0"""
    xka=omega/a; xkb=omega/b
    ra=np.sqrt((k+xka)*abs(k-xka)); rb=np.sqrt((k+xkb)*abs(k-xkb))
    t=b/omega; gammk=2*t*t; gam=gammk*k2
    p=ra*d; q=rb*d
    pex=sex=0.0
    if k<xka:
        sinp,cosp=np.sin(p),np.cos(p); w=sinp/ra; x=-ra*sinp
    else:
        pex=p; fac=np.exp(-2*p) if p<16 else 0.0
        cosp=(1+fac)*0.5; sinp=(1-fac)*0.5; w=sinp/ra; x=ra*sinp
    if k<xkb:
        sinq,cosq=np.sin(q),np.cos(q); y=sinq/rb; z=-rb*sinq
    else:
        sex=q; fac=np.exp(-2*q) if q<16 else 0.0
        cosq=(1+fac)*0.5; sinq=(1-fac)*0.5; y=sinq/rb; z=rb*sinq
    exa=pex+sex
    a0=np.exp(-exa) if exa<60 else 0.0
    cpcq=cosp*cosq; cpy=cosp*y; cpz=cosp*z; cqw=cosq*w; cqx=cosq*x; xy=x*y; xz=x*z; wy=w*y; wz=w*z
    gamm1=gam-1; twgm1=gam+gamm1; gmgmk=gam*gammk; gmgm1=gam*gamm1; gm1sq=gamm1*gamm1; rho2=rho*rho
    a0pq=a0-cpcq
    ca=np.zeros((5,5))
    ca[0,0]=cpcq-2*gmgm1*a0pq-gmgmk*xz-k2*gm1sq*wy
    ca[0,1]=(k2*cpy-cqx)/rho
    ca[0,2]=-(twgm1*a0pq+gammk*xz+k2*gamm1*wy)/rho
    ca[0,3]=(cpz-k2*cqw)/rho
    ca[0,4]=-(2*k2*a0pq+xz+k2*k2*wy)/rho2
    ca[1,0]=(gmgmk*cpz-gm1sq*cqw)*rho
    ca[1,1]=cpcq
    ca[1,2]=gammk*cpz-gamm1*cqw
    ca[1,3]=-wz
    ca[1,4]=ca[0,3]
    ca[3,0]=(gm1sq*cpy-gmgmk*cqx)*rho
    ca[3,1]=-xy
    ca[3,2]=gamm1*cpy-gammk*cqx
    ca[3,3]=ca[1,1]
    ca[3,4]=ca[0,1]
    ca[4,0]=-(2*gmgmk*gm1sq*a0pq+gmgmk*gmgmk*xz+gm1sq*gm1sq*wy)*rho2
    ca[4,1]=ca[3,0]
    ca[4,2]=-(gammk*gamm1*twgm1*a0pq+gam*gammk*gammk*xz+gamm1*gm1sq*wy)*rho
    ca[4,3]=ca[1,0]
    ca[4,4]=ca[0,0]
    tt=-2*k2
    ca[2,0]=tt*ca[4,2]; ca[2,1]=tt*ca[3,2]; ca[2,2]=a0+2*(cpcq-ca[0,0]); ca[2,3]=tt*ca[1,2]; ca[2,4]=tt*ca[0,2]
    return ca
def halfspace(k2,k,omega,a,b,rho):
    xka=omega/a; xkb=omega/b
    ra=np.sqrt((k+xka)*abs(k-xka)); rb=np.sqrt((k+xkb)*abs(k-xkb))
    t=b/omega; gammk=2*t*t; gam=gammk*k2; gamm1=gam-1
    return np.array([rho*rho*(gamm1*gamm1-gam*gammk*ra*rb), -rho*ra, rho*(gamm1-gammk*ra*rb), rho*rb, k2-ra*rb])
def trace(c, omega, h, vp, vs, rho, nsub=1):
    """compound vector after every (sub)layer interface, bottom-up; returns list of vectors (normalised)"""
    k=omega/c; k2=k*k
    n=len(h)
    e=halfspace(k2,k,omega,vp[-1],vs[-1],rho[-1])
    out=[e/np.abs(e).max()]
    for m in range(n-2,-1,-1):
        for s in range(nsub):
            ca=layer_ca(k2,k,omega,vp[m],vs[m],rho[m],h[m]/nsub)
            e=e@ca
            e=e/np.abs(e).max()
            out.append(e.copy())
    return out


def models(seed, nm):
    rs = np.random.RandomState(seed)
    nlay, H, VP, VS, RHO = synth_models(rs, nm, 10, lvz_frac=0.5, ragged=True)
    for b in range(nm):
        n = nlay[b]
        yield [np.float64(np.float32(x[:n, b])) for x in (H, VP, VS, RHO)]


def check(nm=12, ncs=1500, nsub=60):
    viol = tot = 0
    negs = {0: 0, 1: 0, 2: 0}
    for h, vp, vs, rho in models(5, nm):
        for T in (1.5, 4.0, 12.0, 40.0):
            om = 2 * np.pi / T
            prevf, N = None, 0
            for c in np.linspace(0.75 * vs.min(), vs[-1] * 0.9995, ncs):
                tr = trace(c, om, h, vp, vs, rho, nsub=nsub)
                top = tr[-1]
                if prevf is not None and np.sign(top[0]) != np.sign(prevf):
                    N += 1                       # brute force: sign changes of the secular function on a fine grid
                prevf = top[0]
                e5 = np.array([t[4] for t in tr])
                Z = int((np.sign(e5[1:]) != np.sign(e5[:-1])).sum())
                neg = 1 if top[0] * top[4] > 0 else (0 if top[1] * top[4] < 0 else 2)
                negs[neg] += 1
                tot += 1
                viol += N != Z + neg
    print("points", tot, "violations of N = Z + neg:", viol, " neg histogram", negs)


def interior(nm=10, ncs=400, nsub=60):
    tot, rows = 0, []
    for h, vp, vs, rho in models(7, nm):
        n = len(h)
        for T in (1.5, 4.0, 12.0, 40.0):
            om = 2 * np.pi / T
            for c in np.linspace(0.75 * vs.min(), vs[-1] * 0.9995, ncs):
                s = np.sign([t[4] for t in trace(c, om, h, vp, vs, rho, nsub=nsub)])
                for j, m in enumerate(range(n - 2, -1, -1)):
                    seg = s[j * nsub:(j + 1) * nsub + 1]
                    z = int((seg[1:] != seg[:-1]).sum())
                    p = om * h[m] * np.sqrt(abs(1 / vp[m] ** 2 - 1 / c ** 2))
                    q = om * h[m] * np.sqrt(abs(1 / vs[m] ** 2 - 1 / c ** 2))
                    tot += 1
                    if z != int(seg[0] != seg[-1]):
                        rows.append(((c > vp[m]) * 2 + (c > vs[m]) * 1, z, p, q))
    st = np.array(rows)
    print("layer evaluations", tot, " with zeros of e5 the interface signs miss:", len(st))
    for reg, name in ((0, "P, S evanescent"), (1, "S propagating"), (3, "P, S propagating")):
        sel = st[st[:, 0] == reg] if len(st) else st
        if len(sel):
            print("  %-18s %6d  zeros inside %s  smallest q %.2f  smallest phase (q, or p + q) %.2f" % (
                name, len(sel), dict(zip(*[a.astype(int).tolist() for a in np.unique(sel[:, 1], return_counts=True)])),
                sel[:, 3].min(), (sel[:, 2] * (reg == 3) + sel[:, 3]).min()))
        else:
            print("  %-18s      0" % name)


if __name__ == "__main__":
    {"check": check, "interior": interior}[sys.argv[1] if len(sys.argv) > 1 else "check"]()
