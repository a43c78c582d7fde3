"""Sharded chains on the GPU: the RCCL code path, and N ranks == 1 rank.

The GPU boxes of the build have ONE GPU, so (a) the `nccl` (= RCCL) branches of bayhunter_amd.parallel and of
DeviceChains run with world_size = 1, and (b) the two-rank job runs with both ranks on GPU 0 over gloo.  What (b)
pins is the sharding contract: one job-wide seed, global chain numbers in the Philox counter and in the initial
states, exchange decisions from the gathered (logL, beta, ladder), rank 0 writing one file set per ladder with the
beta = 1 samples assembled across ranks -- two ranks x 12 chains write exactly what one rank x 24 chains writes."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _two_rank_job(outdir, backend):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "sharded_rank.py"), outdir, backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "accepted swaps" in r.stdout


def test_two_ranks_over_rccl_one_gpu_each(tmp_path):
    """The sharded job with the REAL collective path: two ranks, one GPU each, nccl (= RCCL over xGMI): chain layout
    gather, the exchange sweeps on the device (all_gather_into_tensor of (logL, beta)), the end-of-run gather -- and the
    files equal those of one rank running all 24 chains.  Needs two GPUs (skipped on the one-GPU boxes of the build; the
    ranks do NOT call torch.cuda.set_device: every buffer must follow DeviceChains(device=))."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    _two_rank_job(two, "nccl")
    sys.path.insert(0, HERE)
    from sharded_rank import job
    job(24, 0, 24, None, one)
    names = sorted(n for n in os.listdir(os.path.join(one, "data")) if n.endswith(".npy"))
    assert sorted(os.listdir(os.path.join(one, "data"))) == sorted(os.listdir(os.path.join(two, "data"))) and len(names) == 60
    for n in names:
        a, b = np.load(os.path.join(one, "data", n)), np.load(os.path.join(two, "data", n))
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), n


def test_two_ranks_write_what_one_rank_writes(tmp_path):
    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    _two_rank_job(two, "gloo")
    sys.path.insert(0, HERE)
    from sharded_rank import job
    dc = job(24, 0, 24, None, one)
    assert dc.sweep == 40 and dc.nswaps > 10
    assert sorted(os.listdir(os.path.join(one, "data"))) == sorted(os.listdir(os.path.join(two, "data")))
    assert "test_config.pkl" in os.listdir(os.path.join(two, "data"))           # what the reference's PlotFromStorage opens
    names = sorted(n for n in os.listdir(os.path.join(one, "data")) if n.endswith(".npy"))
    assert len(names) == 2 * 5 * 6 and names[0] == "c000_p1likes.npy"          # 6 ladders, both phases
    for n in names:
        a, b = np.load(os.path.join(one, "data", n)), np.load(os.path.join(two, "data", n))
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), n
    s = dc.samples("p2", cold_only=True)
    assert (s["beta"] == 1.0).all() and s["models"].shape[:2] == (40, 6)
    full = dc.samples("p2")
    assert full["beta"].shape == (40, 24) and (np.sum(full["beta"] == 1.0, axis=1) == 6).all()
    # the cold (beta = 1) chain of a ladder is not the chain that started cold (chains 0..5): temperatures moved
    first = dc.samples("p1")["beta"][0]
    assert (first[:6] == 1.0).all() and (full["beta"][-1][:6] != 1.0).any()


def test_rccl_code_path_world_size_1():
    """nccl backend (RCCL on ROCm) with one rank: all_gather_rows / tempering_exchange on CUDA tensors and a
    tempered DeviceChains run with dist= set -- every collective the sharded job issues, on the real backend."""
    import torch
    import torch.distributed as dist
    from bayhunter_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = torch.arange(12, dtype=torch.float64, device="cuda").reshape(4, 3)
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)                                                       # a real RCCL launch
        assert float(t.item()) == 1.0
        assert torch.equal(parallel.all_gather_rows(x, dist), x)
        assert parallel.chain_layout(5, dist) == (0, 5)
        logL = torch.tensor([-30.0, -10.0, -20.0, -25.0], dtype=torch.float64, device="cuda")
        beta = torch.tensor([1.0, 0.5, 1.0, 0.5], dtype=torch.float64, device="cuda")
        nb, nacc = parallel.tempering_exchange(logL, beta, np.array([0, 0, 1, 1]), 0, 5, dist)
        ref, nref = parallel.ladder_swap_betas(logL.cpu().numpy(), beta.cpu().numpy(), np.array([0, 0, 1, 1]), 0, 5)
        assert nb.is_cuda and nb.cpu().tolist() == ref.tolist() and nacc == nref and nacc >= 1
        g = parallel.gather_chain_axis(np.arange(6.0).reshape(2, 3), 1, dist)
        assert g.tolist() == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
        sys.path.insert(0, HERE)
        from conftest import golden
        from test_gpu_chains import SETUPS, make_targets
        from bayhunter_amd.device_chains import DeviceChains
        su = SETUPS["exp"]
        C = 16
        init = dict(su["init"], iter_burnin=150, iter_main=50, maxmodels=10)
        betas = np.tile([1.0, 0.4], C // 2)
        dc = DeviceChains(make_targets(golden("chain_golden.npz")), C, init, su["priors"], seed=5, betas=betas,
                          ladder=np.arange(C) // 2, swap_every=20, dist=dist).run()
        assert dc.sweep == 10 and dc.chain_offset == 0 and dc.C_global == C
        st = dc.state_host()
        assert np.array_equal(np.sort(st["beta"].reshape(-1, 2), axis=1), np.tile([0.4, 1.0], (C // 2, 1)))
        assert dc.samples("p2", cold_only=True)["models"].shape[1] == C // 2
    finally:
        dist.destroy_process_group()


def test_bench_line_of_a_two_rank_tempered_job_parses():
    """`bench.py --gpus 2 --workload c5` as the driver launches it (torch.distributed.run, two ranks), in its
    dry-run form (BH_BENCH_DRYRUN=1: both ranks on GPU 0 over gloo -- the boxes of the build have one GPU): rank 0 prints ONE
    parseable line under the size limit, the ladder laid across the ranks exchanges temperatures."""
    import json
    repo = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BH_BENCH_DRYRUN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "2", "--workload", "c5", "--steps", "300",
           "--warmup", "100", "--out", os.path.join(repo, "gpurun_out", "bench_full_dryrun.json")]
    os.makedirs(os.path.join(repo, "gpurun_out"), exist_ok=True)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8000
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "chain-iterations/s" and "DRY RUN" in d["data"]
    assert d["config"]["accepted_swaps"] > 0
    assert d["collective_check"]["world_size"] == 2 and d["collective_check"]["ranks_seen_by_all_reduce"] == 2
