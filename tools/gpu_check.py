#!/usr/bin/env python3
"""Quick on-GPU parity + timing probe (development tool; the real checks live in tests/)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayhunter_amd import engine as E
from oracle import oracle as O

def synth_models(rs, B, L, lvz_frac=0.1, ragged=False):
    nlay = np.full(B, L, dtype=np.int32) if not ragged else rs.randint(2, L + 1, B).astype(np.int32)
    h = np.zeros((L, B)); vp = np.zeros((L, B)); vs = np.zeros((L, B)); rho = np.zeros((L, B))
    for b in range(B):
        n = nlay[b]
        v = np.sort(rs.uniform(2.0, 4.8, n))
        if rs.uniform() < lvz_frac and n > 3:
            i = rs.randint(1, n - 1); v[i] = 0.9 * v[i - 1]
        hh = rs.uniform(1.5, 8.0, n); hh[-1] = 0
        k = rs.uniform(1.6, 1.9)
        vs[:n, b] = v; vp[:n, b] = v * k; h[:n, b] = hh; rho[:n, b] = 0.32 * v * k + 0.77
    return nlay, h, vp, vs, rho

def main():
    out = {}
    eng = E.Engine(0)
    eng.set_swd_search("reference")   # (bit-level comparison with the oracle's reference sequence)
    rs = np.random.RandomState(11)
    # device libm vs host libm
    x = rs.uniform(0.01, 40, 20000)
    for op, name, f in ((0, 'sqrt', np.sqrt), (1, 'sin', np.sin), (2, 'cos', np.cos), (3, 'exp', lambda v: np.exp(-v)), (5, 'recip', lambda v: 1 / v)):
        xin = -x if name == 'exp' else x
        g = eng.probe_math(op, xin); r = f(x) if name != 'exp' else np.exp(xin)
        ulp = np.abs(g - r) / np.spacing(np.abs(r))
        out['ulp_' + name] = [float(ulp.max()), float((ulp > 0).mean())]
    print(json.dumps(out))
    per = np.linspace(2, 60, 30)
    B = 1024
    nlay, h, vp, vs, rho = synth_models(rs, B, 12, ragged=True)
    eng.set_instrumentation(True, True)
    for iwave, igr, name in ((2, 0, 'rdispph'), (2, 1, 'rdispgr'), (1, 0, 'ldispph'), (1, 1, 'ldispgr')):
        t = time.time(); vel, err = eng.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr); tg = time.time() - t
        tot, fam = eng.last_timing(); ne = eng.last_neval()
        t = time.time(); ov, oe, one = O.swd_batch(nlay, h.T.copy(), vp.T.copy(), vs.T.copy(), rho.T.copy(), per, iwave, igr); tc = time.time() - t
        ok = (oe == 0)
        rel = np.abs(vel[ok] - ov[ok]) / np.abs(ov[ok])
        print(name, 'err agree', bool(np.array_equal(err, oe)), 'nerr', int(oe.sum()), 'maxrel', float(rel.max()), 'exact frac', float((rel == 0).mean()),
              'n>1e-5', int((rel > 1e-5).sum()), 'neval gpu/cpu', ne, one, 'kernel ms', round(fam['swd'], 3), 'wall', round(tg, 3), 'cpu s', round(tc, 3))
    # RF
    nl2, h2, vp2, vs2, rho2 = synth_models(rs, 64, 12, ragged=True)
    for nsamp, fs, nk in ((512, 5.0, 201), (2048, 20.0, 1024)):
        for wv in (0, 1):
            rf = eng.rf_batch(nl2, h2, vp2, vs2, rho2, 6.4, 2.5, nsamp, fs, 5.0, wv, nk)
            tot, fam = eng.last_timing()
            orf = O.rf_batch(nl2, h2.T.copy(), vp2.T.copy(), vs2.T.copy(), rho2.T.copy(), 6.4, 2.5, nsamp, fs, 5.0, wv, nk)
            d = np.abs(rf - orf).max(axis=1) / np.abs(orf).max(axis=1)
            print('rf nsamp', nsamp, 'wave', wv, 'max abs diff / peak', float(d.max()), 'abs', float(np.abs(rf - orf).max()), 'kernel ms', round(fam['rf'], 3))
    # timing at the benchmark size
    B = 4096
    nlay, h, vp, vs, rho = synth_models(rs, B, 10)
    for iwave, igr, name in ((2, 0, 'rdispph'), (1, 0, 'ldispph'), (2, 1, 'rdispgr'), (1, 1, 'ldispgr')):
        for rep in range(2):
            vel, err = eng.swd_batch(nlay, h, vp, vs, rho, per, iwave, igr)
        tot, fam = eng.last_timing(); ne = eng.last_neval()
        print('B=4096', name, 'kernel ms', round(fam['swd'], 3), 'neval', ne, 'errs', int(err.sum()))
    rf = eng.rf_batch(nlay, h, vp, vs, rho, 6.4, 2.5, 2048, 20.0, 5.0, 0, 1024)
    rf = eng.rf_batch(nlay, h, vp, vs, rho, 6.4, 2.5, 2048, 20.0, 5.0, 0, 1024)
    tot, fam = eng.last_timing()
    print('B=4096 prf kernel ms', round(fam['rf'], 3), 'total', round(tot, 3))

if __name__ == '__main__':
    main()
