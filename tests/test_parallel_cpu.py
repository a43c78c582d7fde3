"""Multi-process plumbing on CPU: world_size-2 gloo (no GPU).  Covers the N>1 path of bench.py
(rank partition, barrier/max reduction) and the parallel-tempering exchange."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayhunter_amd import parallel


def test_shard_slice_partitions_everything():
    for n in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            seen = np.zeros(n, dtype=int)
            sizes = []
            for r in range(world):
                s = parallel.shard_slice(n, r, world)
                seen[s] += 1
                sizes.append(s.stop - s.start)
            assert np.all(seen == 1) and max(sizes) - min(sizes) <= 1
    assert parallel.chain_owner(63, 64, 8) == 7 and parallel.chain_owner(8, 64, 8) == 1


def test_swap_decisions_properties():
    rs = np.random.RandomState(0)
    n = 16
    beta = 1.0 / np.geomspace(1, 30, n)
    for sweep in range(50):
        logL = rs.normal(-100, 20, n)
        perm = parallel.swap_decisions(logL, beta, sweep, seed=7)
        assert sorted(perm) == list(range(n))                       # a permutation
        assert np.array_equal(perm, parallel.swap_decisions(logL, beta, sweep, seed=7))  # deterministic
        moved = np.flatnonzero(perm != np.arange(n))
        for r in moved:                                             # only neighbour pairs of this parity
            assert abs(perm[r] - r) == 1 and min(r, perm[r]) % 2 == sweep % 2
    # a hotter rung holding a better state always hands it down (alpha >= 1)
    logL = np.array([-100.0, -10.0]); perm = parallel.swap_decisions(logL, [1.0, 0.5], 0, 3)
    assert list(perm) == [1, 0]
    # detailed balance of the rule: accept ratio forward/backward = exp(delta)
    d = (1.0 - 0.5) * (-30.0 - -28.0)
    acc_f = np.mean([parallel.swap_decisions([-28.0, -30.0], [1.0, 0.5], 0, s)[0] == 1 for s in range(4000)])
    assert abs(acc_f - np.exp(d)) < 0.03


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _pt_job():
    rs = np.random.RandomState(4)
    L = rs.normal(-40.0, 6.0, 12)
    Bt = np.repeat([1.0, 0.6], 6)            # rank 0 holds the cold rung of every ladder, rank 1 the warm one
    lad = np.tile(np.arange(6), 2)
    return L, Bt, lad


def test_ladder_swap_betas_properties():
    rs = np.random.RandomState(0)
    nl, nr = 5, 8
    ladder = np.repeat(np.arange(nl), nr)
    beta0 = np.tile(1.0 / np.geomspace(1, 50, nr), nl)
    order = rs.permutation(nl * nr)                       # any chain order
    ladder, beta = ladder[order], beta0[order]
    acc = 0
    for sweep in range(40):
        logL = rs.normal(-100, 15, nl * nr)
        newb, nacc = parallel.ladder_swap_betas(logL, beta, ladder, sweep, seed=3)
        again, _ = parallel.ladder_swap_betas(logL, beta, ladder, sweep, seed=3)
        assert np.array_equal(newb, again)               # deterministic in (seed, sweep)
        for lid in range(nl):                             # every ladder keeps its set of temperatures
            assert np.array_equal(np.sort(newb[ladder == lid]), np.sort(beta[ladder == lid]))
        changed = np.flatnonzero(newb != beta)
        assert changed.size == 2 * nacc
        acc += nacc
        beta = newb
    assert acc > 20
    # a swap that raises the cold chain's likelihood is always accepted; the reverse with probability exp(-delta)
    nb, n1 = parallel.ladder_swap_betas([-100.0, -10.0], [1.0, 0.5], [0, 0], 0, 1)
    assert n1 == 1 and nb.tolist() == [0.5, 1.0]
    f = np.mean([parallel.ladder_swap_betas([-28.0, -30.0], [1.0, 0.5], [0, 0], 0, s)[1] for s in range(4000)])
    assert abs(f - np.exp(-1.0)) < 0.03


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the bench's timing reduction: max over ranks after a barrier
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # sharded "evaluation": every rank fills its own slice, gathered rows reproduce the whole
        n = 11
        sl = parallel.shard_slice(n, rank, world)
        local = torch.arange(sl.start, sl.stop, dtype=torch.float64).reshape(-1, 1) * 2.0
        rows = parallel.all_gather_rows(local, dist)
        # replica exchange: 3 + 2 replicas, same permutation on both ranks
        nl = 3 if rank == 0 else 2
        allL = np.array([-50.0, -40.0, -45.0, -20.0, -60.0]); allB = 1.0 / np.geomspace(1, 8, 5)
        off = 0 if rank == 0 else 3
        perm, mine = parallel.tempering_swap(allL[off:off + nl], allB[off:off + nl], sweep=1, seed=5, dist=dist)
        # temperature exchange of a job sharded "one temperature per rank": 2 ranks x 6 chains, 6 ladders of 2 rungs
        L, Bt, lad = _pt_job()
        my = slice(0, 6) if rank == 0 else slice(6, 12)
        newb, nacc = parallel.tempering_exchange(torch.tensor(L[my]), torch.tensor(Bt[my]), lad[my], sweep=0, seed=9, dist=dist)
        # the same exchange as device work (here CPU tensors over gloo; RCCL when the tensors are on GPUs), three sweeps
        ex = parallel.DeviceExchange(lad, seed=9, mine=my, device="cpu")
        bt = torch.tensor(Bt[my])
        for sw in range(3):
            ex.sweep(torch.tensor(L[my]), bt, sw, dist)
        out.put((rank, float(t.item()), rows.squeeze(1).tolist(), perm.tolist(), (mine.start, mine.stop), newb.tolist(), nacc,
                 bt.tolist(), int(ex.nacc)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, rows0, perm0, mine0, nb0, na0, db0, dn0), (r1, t1, rows1, perm1, mine1, nb1, na1, db1, dn1) = res
    L, Bt, lad = _pt_job()
    expect_b, expect_n = parallel.ladder_swap_betas(L, Bt, lad, 0, 9)
    b3, n3 = Bt, 0
    for sw in range(3):
        b3, k = parallel.ladder_swap_betas(L, b3, lad, sw, 9)
        n3 += k
    assert db0 == b3[:6].tolist() and db1 == b3[6:].tolist() and dn0 == dn1 == n3
    assert na0 == na1 == expect_n and expect_n > 0
    assert nb0 == expect_b[:6].tolist() and nb1 == expect_b[6:].tolist()
    assert t0 == t1 == 2.0
    assert rows0 == rows1 == [2.0 * i for i in range(11)]
    assert perm0 == perm1 and sorted(perm0) == [0, 1, 2, 3, 4]
    expect = parallel.swap_decisions([-50.0, -40.0, -45.0, -20.0, -60.0], 1.0 / np.geomspace(1, 8, 5), 1, 5)
    assert perm0 == expect.tolist()
    assert mine0 == (0, 3) and mine1 == (3, 5)


# ---- sharded chains: global numbering, job-wide seed, gather, cold-chain assembly, result files ----------
class ToyChains(object):
    """A stand-in for DeviceChains without a GPU: one scalar state per chain, random-walk Metropolis on
    logL(x) = -20 x^2 with tempered acceptance; every draw is a function of (job seed, GLOBAL chain index,
    iteration), the sharding protocol is the product's (bayhunter_amd.parallel): chain_layout / chain_seeds /
    tempering_exchange / gather_chain_axis / cold_samples / write_chain_files."""

    def __init__(self, nlocal, seed, ladders, rungs, swap_every, dist_=None):
        self.C, self.seed, self.dist, self.swap_every = nlocal, seed, dist_, swap_every
        self.off, self.tot = parallel.chain_layout(nlocal, dist_)
        gid = self.off + np.arange(nlocal)
        self.gid = gid
        self.ladder = gid % ladders                                   # ladders span the ranks
        self.beta = None if rungs == 0 else (1.0 / np.geomspace(1.0, 6.0, rungs))[gid // ladders]
        self.x = parallel.chain_seeds(seed, self.off, nlocal) / 2.0 ** 31 - 0.5
        self.sweep, self.it, self.snap = 0, 0, []

    @staticmethod
    def logL(x):
        return -20.0 * x * x

    def step(self):
        for c in range(self.C):
            rs = np.random.RandomState([self.seed, int(self.gid[c]), self.it])
            xn = self.x[c] + 0.3 * rs.normal()
            b = 1.0 if self.beta is None else self.beta[c]
            if np.log(rs.uniform()) < b * (self.logL(xn) - self.logL(self.x[c])):
                self.x[c] = xn
        self.it += 1
        if self.beta is not None and self.it % self.swap_every == 0:
            nb, _ = parallel.tempering_exchange(torch.tensor(self.logL(self.x)), torch.tensor(self.beta), self.ladder,
                                                self.sweep, self.seed, self.dist)
            self.beta = nb.numpy().copy()
            self.sweep += 1
        if self.it % 2 == 0:
            self.snap.append((self.x.copy(), None if self.beta is None else self.beta.copy()))

    def save(self, path):
        ns = len(self.snap)
        x = np.array([s[0] for s in self.snap])                       # [ns, C]
        s = dict(models=np.stack((x, np.broadcast_to(self.gid, x.shape).astype(float)), axis=2).astype(np.float32),
                 likes=self.logL(x).astype(np.float32), misfits=np.zeros((ns, self.C, 2), np.float32),
                 noise=np.zeros((ns, self.C, 2), np.float32), vpvs=np.full((ns, self.C), 1.7, np.float32))
        if self.beta is not None:
            s["beta"] = np.array([q[1] for q in self.snap])
        s = {k: parallel.gather_chain_axis(v, 1, self.dist) for k, v in s.items()}
        ids = np.arange(s["models"].shape[1])
        if self.beta is not None:
            ids, s = parallel.cold_samples(s, parallel.gather_chain_axis(self.ladder, 0, self.dist))
        rank = self.dist.get_rank() if self.dist is not None else 0
        if rank == 0:
            parallel.write_chain_files(path, "p2", s, ids)
        return s


def _toy_worker(rank, world, port, path, rungs):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nl = 7 if rank == 0 else 5                                    # uneven blocks: 12 chains in all
        tc = ToyChains(nl, seed=77, ladders=4, rungs=rungs, swap_every=3, dist_=dist)
        assert (tc.off, tc.tot) == ((0, 12) if rank == 0 else (7, 12))
        for _ in range(40):
            tc.step()
            if rungs:                                                 # every ladder keeps its set of temperatures
                allb = parallel.gather_chain_axis(tc.beta, 0, dist)
                alll = parallel.gather_chain_axis(tc.ladder, 0, dist)
                for lid in range(4):
                    assert np.allclose(np.sort(allb[alll == lid]), np.sort(1.0 / np.geomspace(1.0, 6.0, rungs)))
        tc.save(path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rungs", [0, 3])
def test_sharded_chains_write_what_the_single_rank_job_writes(tmp_path, rungs):
    """2 ranks (7 + 5 chains) against 1 rank x 12 chains with the same job seed: same chains, same exchange
    decisions, same files with global numbers; tempered: one file set per ladder holding the beta = 1 samples,
    assembled across ranks (3 rungs, small logL gaps: swaps are not near-deterministic)."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    two = str(tmp_path / "two")
    procs = [ctx.Process(target=_toy_worker, args=(r, 2, port, two, rungs)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    one = str(tmp_path / "one")
    tc = ToyChains(12, seed=77, ladders=4, rungs=rungs, swap_every=3)
    for _ in range(40):
        tc.step()
    s = tc.save(one)
    names = sorted(os.listdir(one))
    assert names == sorted(os.listdir(two))
    n_sets = 4 if rungs else 12
    assert len(names) == 5 * n_sets and names[0] == "c000_p2likes.npy" and names[-1] == "c%03d_p2vpvs.npy" % (n_sets - 1)
    for nme in names:
        assert np.array_equal(np.load(os.path.join(one, nme)), np.load(os.path.join(two, nme))), nme
    if rungs:
        assert tc.sweep == 13 and (s["beta"] == 1.0).all()             # only cold samples are written
        moved = np.load(os.path.join(one, "c000_p2models.npy"))[:, 1]   # global id of the chain that was cold
        assert len(set(moved.tolist())) > 1                            # the cold chain really moved between chains
    else:
        ids = [int(np.load(os.path.join(two, "c%03d_p2models.npy" % c))[0, 1]) for c in range(12)]
        assert ids == list(range(12))                                   # 12 distinct chains, numbered globally


def test_device_exchange_takes_the_decisions_of_the_numpy_form():
    """parallel.DeviceExchange (the swap sweep as torch operations; on a GPU it runs on the engine's stream) against
    parallel.ladder_swap_betas, here on CPU tensors: unequal ladders scattered over the chains, tied temperatures,
    many sweeps of both parities, -1e15 sentinels among the likelihoods."""
    import torch
    from bayhunter_amd import parallel as P
    rs = np.random.RandomState(9)
    sizes = [1, 2, 3, 8, 5, 8, 7, 4]
    ladder = rs.permutation(np.repeat(np.arange(len(sizes)) * 3 + 1, sizes))       # ids need not be 0..n-1
    N = ladder.size
    beta = np.ones(N)
    for lid, n in zip(np.arange(len(sizes)) * 3 + 1, sizes):
        b = 1.0 / np.geomspace(1.0, 25.0, n)
        if n >= 5:
            b[3] = b[2]
        beta[ladder == lid] = rs.permutation(b)
    ex = P.DeviceExchange(ladder, seed=4711, mine=slice(0, N), device="cpu")
    bt = torch.from_numpy(beta.copy())
    bn = beta.copy()
    nacc = 0
    for sweep in range(60):
        logL = rs.normal(300.0, 8.0, N)
        logL[rs.rand(N) < 0.05] = -1e15
        bn, k = P.ladder_swap_betas(logL, bn, ladder, sweep, 4711)
        nacc += k
        ex.sweep(torch.from_numpy(logL), bt, sweep)
        assert np.array_equal(bt.numpy(), bn), sweep
    assert int(ex.nacc) == nacc and nacc > 50
    for lid in np.unique(ladder):                                                    # temperatures stay in their ladder
        assert np.array_equal(np.sort(bn[ladder == lid]), np.sort(beta[ladder == lid]))
