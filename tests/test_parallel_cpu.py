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
        out.put((rank, float(t.item()), rows.squeeze(1).tolist(), perm.tolist(), (mine.start, mine.stop), newb.tolist(), nacc))
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
    (r0, t0, rows0, perm0, mine0, nb0, na0), (r1, t1, rows1, perm1, mine1, nb1, na1) = res
    L, Bt, lad = _pt_job()
    expect_b, expect_n = parallel.ladder_swap_betas(L, Bt, lad, 0, 9)
    assert na0 == na1 == expect_n and expect_n > 0
    assert nb0 == expect_b[:6].tolist() and nb1 == expect_b[6:].tolist()
    assert t0 == t1 == 2.0
    assert rows0 == rows1 == [2.0 * i for i in range(11)]
    assert perm0 == perm1 and sorted(perm0) == [0, 1, 2, 3, 4]
    expect = parallel.swap_decisions([-50.0, -40.0, -45.0, -20.0, -60.0], 1.0 / np.geomspace(1, 8, 5), 1, 5)
    assert perm0 == expect.tolist()
    assert mine0 == (0, 3) and mine1 == (3, 5)
