"""numpy restatement of the device chain step's random numbers (test infrastructure).

Philox4x32-10 (Salmon et al., SC'11; Random123) with counter (chain, iteration, purpose, 0) and key =
the two halves of the seed, and the mapping from its words to the six draws of one iteration, as in
bayhunter_amd/csrc/chain_kernel.hip `get_draws`.  Checked against the Random123 known-answer vectors
in tests/test_host_logic.py."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 array [..., 4]; key: (k0, k1) python ints -> uint32 array [..., 4]"""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def _u01(hi, lo):
    v = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    return (v >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def draws(seed, C, iiter, offset=0):
    """-> [6, C]: u_move, u_index, u_z, u_accept, u_noise, normal for iteration `iiter` of the chains with
    global indices offset .. offset + C - 1."""
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    ctr = np.zeros((4, C, 4), dtype=np.uint32)
    ctr[:, :, 0] = (np.arange(C, dtype=np.uint64) + np.uint64(offset)).astype(np.uint32)
    ctr[:, :, 1] = np.uint32(iiter & 0xFFFFFFFF)
    ctr[:, :, 2] = np.arange(4, dtype=np.uint32)[:, None]
    r, q, t, n = philox4x32_10(ctr, key)
    out = np.zeros((6, C))
    out[0], out[1] = _u01(r[:, 0], r[:, 1]), _u01(r[:, 2], r[:, 3])
    out[2], out[3] = _u01(q[:, 0], q[:, 1]), _u01(q[:, 2], q[:, 3])
    out[4] = _u01(t[:, 0], t[:, 1])
    a = 1.0 - _u01(n[:, 0], n[:, 1])
    b = _u01(n[:, 2], n[:, 3])
    out[5] = np.sqrt(-2.0 * np.log(a)) * np.cos(2.0 * np.pi * b)
    return out


class InjectedRandomState(object):
    """Stands in for a chain's numpy RandomState inside bayhunter_amd.chains.ChainBatch: every call
    returns the injected draw that the device kernel uses for the same purpose."""

    def __init__(self):
        self.d = None

    def set(self, d6):
        self.d = d6

    def choice(self, seq):
        u = self.d[4] if isinstance(seq, np.ndarray) else self.d[0]   # noise index | modification
        return seq[min(int(u * len(seq)), len(seq) - 1)]

    def randint(self, low=0, high=None):
        low, high = int(low), int(high)
        return low + min(int(self.d[1] * (high - low)), high - low - 1)

    def uniform(self, low=0.0, high=1.0):
        if low == 0 and high == 1:
            return self.d[3]                                        # acceptance draw
        return low + self.d[2] * (high - low)                       # birth depth

    def normal(self, loc=0.0, scale=1.0):
        return loc + self.d[5] * scale
