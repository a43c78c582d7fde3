"""Multi-GPU plumbing: one process per GPU, models / chains sharded by rank.

The hot path shards by model with no data-path collective: every rank evaluates
its own slice.  The only exchanges are (1) gathering per-chain summaries at checkpoints and
(2) the replica-exchange (parallel-tempering) swap of BASELINE.json config 5 -- an all-gather of
a few floats per replica (RCCL over xGMI on GPUs, gloo in the CPU tests), after which every rank
computes the same swap permutation from a shared seed.  Parallel tempering has no counterpart in
the reference: "parity unpinned"; it is checked through invariants (permutation validity, detailed
balance of the acceptance rule, identical decisions on every rank).
"""
import threading

import numpy as np


def _cuda_device(device=None):
    """torch device for collective buffers under nccl: the GPU named by the caller (a chain driver passes its
    engine's GPU), else the current one.  Never the implicit `'cuda'` of a process that has not selected a device:
    with one rank per GPU that would put every rank's buffers on GPU 0."""
    import torch
    return torch.device("cuda", torch.cuda.current_device() if device is None else int(device))


def shard_slice(n_items, rank, world):
    """Contiguous block partition of `n_items` units over `world` ranks (sizes differ by <= 1)."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def chain_owner(chain, n_chains, world):
    """Rank that owns global chain index `chain` under shard_slice."""
    for r in range(world):
        s = shard_slice(n_chains, r, world)
        if s.start <= chain < s.stop:
            return r
    raise IndexError(chain)


def all_gather_rows(local, dist=None, counts=None):
    """Concatenate per-rank row blocks [n_r, m] (n_r may differ by rank) on every rank.
    counts: the n_r of every rank when the caller knows them (static layouts): the call is then ONE fixed-size
    collective with no host synchronisation; otherwise the counts are gathered first (an extra collective and a
    device-to-host read)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    home = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:   # CPU tests / dry runs: gloo moves host memory
        return all_gather_rows(local.cpu(), dist, counts).to(home)
    if counts is None:
        n = torch.tensor([local.shape[0]], device=local.device)
        cts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(cts, n)
        counts = [int(c.item()) for c in cts]
    counts = [int(c) for c in counts]
    if counts[dist.get_rank()] != local.shape[0]:
        raise ValueError("all_gather_rows: rank %d holds %d rows, counts say %d" % (dist.get_rank(), local.shape[0], counts[dist.get_rank()]))
    nmax = max(counts)
    if local.shape[0] == nmax:
        pad = local.contiguous()
    else:
        pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    out = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    if all(c == nmax for c in counts):
        return out
    return torch.cat([out[r * nmax:r * nmax + c] for r, c in enumerate(counts)], dim=0)


_TLS = threading.local()   # scratch generator of the swap decisions, one per thread (two chain drivers may run in threads)


def _scratch_rs():
    rs = getattr(_TLS, "rs", None)
    if rs is None:
        rs = _TLS.rs = np.random.RandomState(0)
    return rs


def swap_decisions(logL, beta, sweep, seed):
    """Replica-exchange decisions for one sweep.  `logL[r]`, `beta[r]` for all replicas ordered by
    temperature rung; neighbours (r, r+1) with r of parity `sweep % 2` are proposed; accept with
    probability min(1, exp((beta_r - beta_{r+1}) * (logL_{r+1} - logL_r))).  Deterministic in
    (seed, sweep): every rank computes the identical answer from the gathered values.
    Returns the permutation `perm` such that rung r continues with the state of rung perm[r]."""
    logL = np.asarray(logL, dtype=float)
    beta = np.asarray(beta, dtype=float)
    n = logL.size
    perm = np.arange(n)
    rs = _scratch_rs()
    rs.seed((int(seed) * 1000003 + int(sweep)) % (2 ** 32))       # (re-seeding is 30x cheaper than a new RandomState)
    u = rs.uniform(size=n)
    for r in range(sweep % 2, n - 1, 2):
        log_alpha = (beta[r] - beta[r + 1]) * (logL[r + 1] - logL[r])
        if np.log(u[r]) < log_alpha:
            perm[r], perm[r + 1] = perm[r + 1], perm[r]
    return perm


def tempering_swap(local_logL, local_beta, sweep, seed, dist=None, device=None):
    """All-gather (logL, beta) of the local replicas (rank-major rung order), decide the swaps.
    Returns (perm over all replicas, slice of the global arrays owned by this rank).
    device: this rank's GPU (nccl groups; default the current device)."""
    import torch
    local = torch.stack((torch.as_tensor(local_logL, dtype=torch.float64),
                         torch.as_tensor(local_beta, dtype=torch.float64)), dim=1)
    if dist is not None and dist.is_initialized() and dist.get_backend() == "nccl":
        local = local.to(_cuda_device(device))
    allv = all_gather_rows(local, dist).cpu().numpy()
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    perm = swap_decisions(allv[:, 0], allv[:, 1], sweep, seed)
    return perm, shard_slice(allv.shape[0], rank, world) if world > 1 else slice(0, allv.shape[0])


def ladder_swap_betas(logL, beta, ladder, sweep, seed):
    """Replica exchange by swapping TEMPERATURES, so that no chain state ever moves between chains
    (or GPUs): `logL[i]`, `beta[i]`, `ladder[i]` for ALL chains of the job (any order, e.g. rank-major
    after an all-gather).  Inside every ladder the chains are ordered by temperature (beta descending,
    ties by chain index) and neighbours (r, r+1) with r of parity `sweep % 2` exchange their betas
    with probability min(1, exp((beta_r - beta_{r+1}) * (logL_{r+1} - logL_r))) -- the rule of
    `swap_decisions`.  Deterministic in (seed, sweep): every rank computes the identical answer.
    Returns (new beta array, number of accepted swaps)."""
    logL = np.asarray(logL, dtype=float)
    beta = np.asarray(beta, dtype=float)
    ladder = np.asarray(ladder)
    out = beta.copy()
    nacc = 0
    for lid in np.unique(ladder):
        idx = np.flatnonzero(ladder == lid)
        order = idx[np.lexsort((idx, -beta[idx]))]          # rung 0 = coldest
        perm = swap_decisions(logL[order], beta[order], sweep, (int(seed) * 7919 + int(lid)) % (2 ** 31))
        # rung r continues with the state of rung perm[r]  <=>  the chain at rung perm[r] takes beta of rung r
        out[order[perm]] = beta[order]
        nacc += int(np.count_nonzero(perm != np.arange(perm.size)) // 2)
    return out, nacc


def tempering_exchange(local_logL, local_beta, local_ladder, sweep, seed, dist=None):
    """The swap step of a sharded job: all-gather (logL, beta, ladder) of every rank's chains (RCCL over
    xGMI when the tensors are on GPUs; a few floats per chain), decide with `ladder_swap_betas`, return
    this rank's new betas (same dtype/device as `local_beta`) and the global number of accepted swaps."""
    import torch
    lb = torch.as_tensor(local_beta)
    local = torch.stack((torch.as_tensor(local_logL, dtype=torch.float64, device=lb.device),
                         lb.to(torch.float64), torch.as_tensor(local_ladder, device=lb.device).to(torch.float64)), dim=1)
    allv = all_gather_rows(local, dist).cpu().numpy()
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    newb, nacc = ladder_swap_betas(allv[:, 0], allv[:, 1], allv[:, 2].astype(np.int64), sweep, seed)
    if world > 1:   # rank-major concatenation: this rank's block
        counts = all_gather_rows(torch.tensor([[local.shape[0]]], device=lb.device, dtype=torch.int64), dist).cpu().numpy().ravel()
        start = int(counts[:rank].sum())
        mine = slice(start, start + local.shape[0])
    else:
        mine = slice(0, local.shape[0])
    return torch.as_tensor(newb[mine], dtype=lb.dtype, device=lb.device), nacc


class DeviceExchange(object):
    """`tempering_exchange` without leaving the GPU: the decisions of `ladder_swap_betas` as torch operations on the
    stream the chains run on, so that a swap sweep is enqueued like an iteration -- no host synchronisation, no copy
    of (logL, beta) to the host and back.  What the decisions need besides the gathered (logL, beta) is static or
    host-computable: the ladder of every chain of the job, and the uniforms of a sweep (they depend on (seed, ladder,
    sweep) only; their logarithms are uploaded from pinned memory).  Sharded jobs all-gather (logL, beta) with RCCL
    (device tensors).  Same decisions as the NumPy form, bit for bit (tests/test_gpu_sharded.py)."""

    def __init__(self, ladder_all, seed, mine, device, rank_counts=None):
        """rank_counts: chains per rank of the job (static): the gather of a sweep is then one fixed-size
        collective and the sweep is enqueue-only on every rank count, not just on one."""
        import torch
        self.torch, self.dev, self.seed, self.mine = torch, device, int(seed), mine
        self.rank_counts = None if rank_counts is None else [int(c) for c in rank_counts]
        lad = np.asarray(ladder_all, dtype=np.int64)
        self.N = lad.size
        self.lids, self.counts = np.unique(lad, return_counts=True)          # sorted ladder ids, chains per ladder
        # layout after sorting by (ladder, -beta, index): position p = rung r(p) of ladder lid(p), static
        starts = np.concatenate(([0], np.cumsum(self.counts)[:-1]))
        rung = np.arange(self.N) - np.repeat(starts, self.counts)
        last = rung == np.repeat(self.counts, self.counts) - 1
        self.left = [torch.as_tensor(((rung % 2 == par) & ~last)[:-1] if self.N > 1 else np.zeros(0, bool), device=device)
                     for par in (0, 1)]
        self.ladder = torch.as_tensor(lad, device=device)
        self.nacc = torch.zeros((), dtype=torch.int64, device=device)

    def _log_uniforms(self, sweep):
        """log(u) in the sorted layout: ladder by ladder the numbers `swap_decisions` draws."""
        out = np.empty(self.N)
        p = 0
        for lid, n in zip(self.lids, self.counts):
            s2 = (self.seed * 7919 + int(lid)) % (2 ** 31)
            rs = _scratch_rs()
            rs.seed((s2 * 1000003 + int(sweep)) % (2 ** 32))
            out[p:p + n] = rs.uniform(size=n)
            p += n
        with np.errstate(divide="ignore"):
            return np.log(out)

    def sweep(self, like, beta, sweep, dist=None):
        """Enqueue one sweep on the CURRENT torch stream: `like`, `beta` = this rank's chains (device tensors);
        `beta` is updated in place."""
        torch = self.torch
        # a fresh pinned buffer per sweep: the host runs ahead of the stream, and torch's host allocator reuses a
        # pinned block only after the copy that reads it has completed
        logu = torch.from_numpy(self._log_uniforms(sweep))
        if torch.device(self.dev).type == "cuda":
            logu = logu.pin_memory().to(self.dev, non_blocking=True)
        local = torch.stack((like.to(torch.float64), beta.to(torch.float64)), dim=1)
        allv = all_gather_rows(local, dist, self.rank_counts)
        L, Bt = allv[:, 0], allv[:, 1]
        o1 = torch.sort(-Bt, stable=True).indices                      # beta descending, ties by chain index
        order = o1[torch.sort(self.ladder[o1], stable=True).indices]   # grouped by ladder, coldest first
        bs, ls = Bt[order], L[order]
        if self.N > 1:
            acc = self.left[int(sweep) % 2] & (logu[:-1] < (bs[:-1] - bs[1:]) * (ls[1:] - ls[:-1]))
            nb = bs.clone()
            nb[:-1] = torch.where(acc, bs[1:], nb[:-1])
            nb[1:] = torch.where(acc, bs[:-1], nb[1:])
            self.nacc += acc.sum()
        else:
            nb = bs
        out = torch.empty_like(Bt)
        out[order] = nb
        beta.copy_(out[self.mine].to(beta.dtype))


# ---- sharded chains: global numbering, result gather, cold-chain assembly -------------------------
def rank_chain_counts(n_local, dist=None, device=None):
    """Chains held by every rank of the job (one small all-gather; rank order)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(n_local)]
    dev = _cuda_device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
    world = dist.get_world_size()
    return all_gather_rows(torch.tensor([[int(n_local)]], dtype=torch.int64, device=dev), dist, [1] * world).cpu().numpy().ravel().tolist()


def chain_layout(n_local, dist=None, device=None):
    """(offset of this rank's first chain, total chains of the job): ranks own contiguous blocks of the
    global chain list in rank order (block sizes may differ).  device: this rank's GPU (nccl groups: where the
    collective's buffer lives; default the current device)."""
    counts = rank_chain_counts(n_local, dist, device)
    if len(counts) == 1:
        return 0, int(n_local)
    return int(sum(counts[:dist.get_rank()])), int(sum(counts))


def chain_seeds(seed, offset, n_local):
    """Per-chain seeds of the initial-state generator, a function of the job seed and the GLOBAL chain index
    (so that a sharded job starts every chain where the unsharded job would)."""
    rs = np.random.RandomState((int(seed) ^ (int(seed) >> 32)) & 0xFFFFFFFF)
    return rs.randint(0, 2 ** 31 - 1, size=int(offset) + int(n_local))[int(offset):]


def gather_chain_axis(a, axis, dist=None, device=None):
    """numpy array with a chain axis (this rank's chains) -> the same array for ALL chains of the job, on every
    rank (all-gather; RCCL when the process group is nccl, gloo otherwise).  device: this rank's GPU (nccl)."""
    import torch
    a = np.asarray(a)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return a
    moved = np.ascontiguousarray(np.moveaxis(a, axis, 0))
    t = torch.from_numpy(moved.reshape(moved.shape[0], -1))
    if dist.get_backend() == "nccl":
        t = t.to(_cuda_device(device))
    out = all_gather_rows(t, dist).cpu().numpy()
    return np.moveaxis(out.reshape((out.shape[0],) + moved.shape[1:]), 0, axis)


def cold_samples(samples, ladder):
    """Posterior samples of a tempered run: for every ladder and every snapshot the state of the chain that
    holds beta = 1 (the largest beta of the ladder) at that time.  `samples`: dict of arrays [nsnap, C, ...]
    incl. "beta" [nsnap, C]; `ladder` [C] ladder id of every chain.  Returns (ladder ids, dict of arrays
    [nsnap, n_ladders, ...])."""
    ladder = np.asarray(ladder)
    ids = np.unique(ladder)
    beta = np.asarray(samples["beta"])
    ns = beta.shape[0]
    pick = np.zeros((ns, ids.size), dtype=np.int64)
    for k, lid in enumerate(ids):
        idx = np.flatnonzero(ladder == lid)
        pick[:, k] = idx[np.argmax(beta[:, idx], axis=1)]
    rows = np.arange(ns)[:, None]
    return ids, {k: np.asarray(v)[rows, pick] for k, v in samples.items()}


def write_chain_files(datapath, tag, samples, ids):
    """c%03d_<tag>{models,likes,misfits,noise,vpvs}.npy, one set per column of `samples` ([nsnap, n, ...]),
    numbered by `ids` -- the reference's per-chain result files (src/SingleChain.py:646-690)."""
    import os
    os.makedirs(datapath, exist_ok=True)
    for j, cid in enumerate(ids):
        for k in ("models", "likes", "misfits", "noise", "vpvs"):
            np.save(os.path.join(datapath, "c%.3d_%s%s" % (int(cid), tag, k)), np.asarray(samples[k])[:, j])
