"""Synthetic workloads of SURVEY.md 8(d): seeded batches of layered models, observed data and
noise hyper-parameters for bench.py and the parity tests (no network, no datasets)."""
import numpy as np

SEED = 20260927


def synth_models(rs, B, L, lvz_frac=0.1, ragged=False, hmin=1.5, hmax=8.0):
    """B models with L layers incl. half-space (ragged: 2..L layers each), layer-major
    float64 [L, B]: vs = sorted U(2.0, 4.8) km/s (a fraction gets one low-velocity layer),
    h = U(hmin, hmax) km, vp/vs = U(1.6, 1.9), rho = 0.32 vp + 0.77."""
    nlay = (rs.randint(2, L + 1, B) if ragged else np.full(B, L)).astype(np.int32)
    h = np.zeros((L, B)); vp = np.zeros((L, B)); vs = np.zeros((L, B)); rho = np.zeros((L, B))
    for b in range(B):
        n = int(nlay[b])
        v = np.sort(rs.uniform(2.0, 4.8, n))
        if rs.uniform() < lvz_frac and n > 3:
            i = rs.randint(1, n - 1)
            v[i] = 0.9 * v[i - 1]
        hh = rs.uniform(hmin, hmax, n)
        hh[-1] = 0.0
        k = rs.uniform(1.6, 1.9)
        vs[:n, b] = v
        vp[:n, b] = v * k
        h[:n, b] = hh
        rho[:n, b] = 0.32 * v * k + 0.77
    return nlay, h, vp, vs, rho


def prior_models(rs, B, L, vs=(2.0, 5.0), z=(0.0, 60.0), vpvs=(1.4, 2.1), thickmin=0.1, nmin=2):
    """B models drawn from a sampler's prior (src/SingleChain.py:71-157, src/Models.py:39-52): nmin..L layers incl. the
    half-space, Voronoi nuclei with vs = U(vs) in ANY order at depths U(z), interfaces at the midpoints of consecutive
    nuclei, vp/vs = U(vpvs), rho = 0.32 vp + 0.77; redrawn until every finite layer is at least thickmin thick.  What a
    chain's proposals look like (low-velocity zones everywhere, thin layers, a half-space that need not be the fastest
    layer) -- unlike synth_models' sorted velocities.  Layer-major float64 [L, B]."""
    nlay = rs.randint(nmin, L + 1, B).astype(np.int32)
    h = np.zeros((L, B)); vp = np.zeros((L, B)); vs_ = np.zeros((L, B)); rho = np.zeros((L, B))
    for b in range(B):
        n = int(nlay[b])
        while True:
            zn = np.sort(rs.uniform(z[0], z[1], n))
            hh = np.diff(np.concatenate(([0.0], 0.5 * (zn[1:] + zn[:-1]))))
            if n == 1 or hh.min() >= thickmin:
                break
        v = rs.uniform(vs[0], vs[1], n)
        k = rs.uniform(vpvs[0], vpvs[1])
        vs_[:n, b] = v
        vp[:n, b] = v * k
        h[:n - 1, b] = hh
        rho[:n, b] = 0.32 * v * k + 0.77
    return nlay, h, vp, vs_, rho


def true_model(L=10):
    """The fixed 'true' model observed data are generated from."""
    rs = np.random.RandomState(SEED + 1)
    return synth_models(rs, 1, L, lvz_frac=0.0)


SWD_PERIODS = np.linspace(2.0, 60.0, 30)      # C2/C3: 30 periods
RF_TIME = -5.0 + 0.05 * np.arange(1024)       # C3: 20 Hz, tshift 5 s -> nsamp 2048
